#!/bin/bash
# round 3, experiment 24: SGPR plane base + 32-bit lane byte offset (saddr form of global_load / global_store; default)
# against pointer + 64-bit lane offset (addr64): 11 -> 3 address instructions per wave
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu > $O/exp24_tests.log 2>&1; tail -1 $O/exp24_tests.log
grep -q " passed" $O/exp24_tests.log || exit 1
grep -q " failed" $O/exp24_tests.log && exit 1
for w in layer lab_stage hsv_stage adjust_rgb rgb2lab; do
echo "== $w: A = addr64, B = default (saddr + 32-bit offset)" >> $O/exp24_saddr.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_addr64.so $L $w 2>&1 | grep -v amdgpu >> $O/exp24_saddr.log || exit 1
done
cat $O/exp24_saddr.log
