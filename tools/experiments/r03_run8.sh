#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -x -q -m gpu -k "uneven or backward or golden" > $O/run8_tests.log 2>&1; tail -2 $O/run8_tests.log
grep -q " passed" $O/run8_tests.log || exit 1
L=curl_amd/lib/libcurlhip.so
for b in 8 32; do B=$b python3 tools/ab.py curl_amd/lib/variants/libcurlhip_r03head.so $L layer_bwd > $O/run8_bwd$b.log 2>&1; grep -v amdgpu $O/run8_bwd$b.log; done
B=32 H=256 W=256 python3 tools/ab.py curl_amd/lib/variants/libcurlhip_r03head.so $L layer_bwd > $O/run8_bwd_crop.log 2>&1; grep -v amdgpu $O/run8_bwd_crop.log
