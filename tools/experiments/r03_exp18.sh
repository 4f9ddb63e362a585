#!/bin/bash
# round 3, experiment 18: CURLLoss terms forward with rgb2lab's selects predicated / skipped per wave (loss_lazy) vs default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
CURL_HIP_LIB=$R/$V/libcurlhip_loss_lazy.so python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py -x -q -m gpu -k "loss" > $O/exp18_tests.log 2>&1; tail -1 $O/exp18_tests.log
echo "== loss_fwd: A = default, B = loss_lazy" > $O/exp18_loss_lazy.log
LAUNCHES=200 ROUNDS=15 python3 tools/ab.py $L $V/libcurlhip_loss_lazy.so loss_fwd 2>&1 | grep -v amdgpu >> $O/exp18_loss_lazy.log || exit 1
cat $O/exp18_loss_lazy.log
