#!/bin/bash
# one data point of the box-to-box spread: the 3-plane copy probe and the headline workload on whatever box this call got
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/${TAG:-r04}; mkdir -p $O
ID=$(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)
C=$(tools/ubench/copy3 2>/dev/null | grep "sustained tile T=256 U=2 nt=1" | head -1 | awk '{print $6}')
python3 bench.py --no-extras --steps 1000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gpu', '${ID:-unknown}', 'copy3_us', '$C', 'layer_us_per_step', round(d['device_ms_per_step']*1e3,1), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/box_spread.log
