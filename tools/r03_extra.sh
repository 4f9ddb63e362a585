#!/bin/bash
# round 3: board power per kernel, parity stress over knot spreads, the headline over batch sizes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 tools/power_sample.py 3 > $O/power_sample.log 2>&1 || { tail -5 $O/power_sample.log; exit 1; }
grep -v amdgpu $O/power_sample.log | tail -12
python3 tools/parity_stress.py > $O/parity_stress.log 2>&1 || { tail -5 $O/parity_stress.log; exit 1; }
tail -3 $O/parity_stress.log
for b in 1 2 4 8 16 64 128; do
  python3 bench.py --batch $b --no-extras --steps 300 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d['config']['batch_per_gpu'], 'us/step', round(d['device_ms_per_step']*1e3,1), 'Mpix/s', round(d['value']), 'frac', round(d['roofline']['frac'],3))" >> $O/batch_sweep.log
done
cat $O/batch_sweep.log
