#!/bin/bash
# round 3, experiment 22: backward with the forward's knot workspace handed back (CURL_F_WS_READY: no prep launch)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 -m pytest tests/test_gpu_backward.py -x -q -m gpu > $O/exp22_tests.log 2>&1; tail -1 $O/exp22_tests.log
grep -q " passed" $O/exp22_tests.log || exit 1
grep -q " failed" $O/exp22_tests.log && exit 1
for w in layer_bwd_crop layer_bwd; do
python3 bench.py --workload $w --no-extras --steps 2000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'us/step', round(d['device_ms_per_step']*1e3,1))" >> $O/exp22_ws_ready.log
done
cat $O/exp22_ws_ready.log
