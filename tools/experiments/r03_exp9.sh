#!/bin/bash
# round 3, experiment 9: the lane's four mask bytes kept as one dword, diagnostics' set-up out of the product path
# (prologue 44 -> 13 VALU instructions per wave) against the previous commit
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -x -q -m gpu > $O/exp9_tests.log 2>&1; tail -1 $O/exp9_tests.log
grep -q " passed" $O/exp9_tests.log || exit 1
for w in layer lab_stage hsv_stage; do
echo "== $w" >> $O/exp9_mask_dword.log
LAUNCHES=400 ROUNDS=21 python3 tools/ab.py $V/libcurlhip_prev.so $L $w 2>&1 | grep -v amdgpu >> $O/exp9_mask_dword.log || exit 1
done
cat $O/exp9_mask_dword.log
