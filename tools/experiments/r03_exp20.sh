#!/bin/bash
# round 3, experiment 20: byte egress of the fused uint8 paths without the saturation pair (values known in [0,1]);
# prev = the previous commit's library through CURL_HIP_LIB
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "u8 or byte or uint8 or infer or config1 or white or bs256_one" > $O/exp20_tests.log 2>&1; tail -1 $O/exp20_tests.log
grep -q " passed" $O/exp20_tests.log || exit 1
grep -q " failed" $O/exp20_tests.log && exit 1
for w in layer_u8 trispace_u8; do
for lib in curl_amd/lib/variants/libcurlhip_prev.so curl_amd/lib/libcurlhip.so curl_amd/lib/variants/libcurlhip_prev.so curl_amd/lib/libcurlhip.so; do
CURL_HIP_LIB=$R/$lib python3 bench.py --workload $w --no-extras --steps 1000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$lib'.split('/')[-1], 'us/step', round(d['device_ms_per_step']*1e3,1))" >> $O/exp20_u8_egress.log
done; done
cat $O/exp20_u8_egress.log
