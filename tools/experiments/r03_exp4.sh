#!/bin/bash
# round 3, experiment 4: the taped layer backward (curl_math_bwd.h rewritten: no recomputation on the way back, clamp gates
# as lane predicates, pixels of a lane one after the other) against the round-2 backward (r03head = the previous commit)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_backward.py -x -q -m gpu > $O/exp4_bwd_tests.log 2>&1; tail -3 $O/exp4_bwd_tests.log
grep -q " passed" $O/exp4_bwd_tests.log || exit 1
for b in 32 8; do
  B=$b python3 tools/ab.py $V/libcurlhip_r03head.so $L layer_bwd >> $O/exp4_layer_bwd_ab.log 2>&1 || exit 1
  B=$b python3 tools/ab.py $L $V/libcurlhip_bwd_w4.so layer_bwd >> $O/exp4_layer_bwd_ab.log 2>&1 || exit 1
  B=$b python3 tools/ab.py $L $V/libcurlhip_bwd_interleave.so layer_bwd >> $O/exp4_layer_bwd_ab.log 2>&1 || exit 1
done
B=32 H=256 W=256 python3 tools/ab.py $V/libcurlhip_r03head.so $L layer_bwd >> $O/exp4_layer_bwd_ab.log 2>&1 || exit 1
grep -v amdgpu.ids $O/exp4_layer_bwd_ab.log
