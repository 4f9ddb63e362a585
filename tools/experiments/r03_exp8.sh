#!/bin/bash
# round 3, experiment 8: the backward tapes' rare linear branches skipped by waves that need none (default) against the
# previous commit (prev); random pixels and pixels without dark values
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_backward.py -x -q -m gpu > $O/exp8_tests.log 2>&1; tail -1 $O/exp8_tests.log
grep -q " passed" $O/exp8_tests.log || exit 1
echo "== 32 x 1500x1000, random pixels" > $O/exp8_bwd_lazy.log
LAUNCHES=100 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_prev.so $L layer_bwd 2>&1 | grep -v amdgpu >> $O/exp8_bwd_lazy.log || exit 1
echo "== 32 x 1500x1000, pixels in [0.2, 1]" >> $O/exp8_bwd_lazy.log
IMG_LO=0.2 LAUNCHES=100 ROUNDS=9 python3 tools/ab.py $V/libcurlhip_prev.so $L layer_bwd 2>&1 | grep -v amdgpu >> $O/exp8_bwd_lazy.log || exit 1
echo "== 32 x 256x256 (training crop batch), random pixels" >> $O/exp8_bwd_lazy.log
B=32 H=256 W=256 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_prev.so $L layer_bwd 2>&1 | grep -v amdgpu >> $O/exp8_bwd_lazy.log || exit 1
cat $O/exp8_bwd_lazy.log
