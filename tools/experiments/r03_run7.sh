#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -x -q -m gpu -s -k "exact_order_validation or backward or pwl" > $O/run7_tests.log 2>&1; grep -E "frame|passed|failed|Error" $O/run7_tests.log | tail -8
grep -q " passed" $O/run7_tests.log || exit 1
L=curl_amd/lib/libcurlhip.so
FLAGS_B=0x1 ROUNDS=5 python3 tools/ab.py $L $L layer > $O/run7_exact_mode_time.log 2>&1; grep -v amdgpu $O/run7_exact_mode_time.log | head -4
B=8 python3 tools/ab.py curl_amd/lib/variants/libcurlhip_r03head.so $L layer_bwd > $O/run7_bwd8.log 2>&1; grep -v amdgpu $O/run7_bwd8.log
