#!/bin/bash
# round 3, experiment 16: the hue's three sextant terms predicated to the lanes whose channel attains the maximum
# (pred_hue) against the default (every lane computes all three and selects)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
CURL_HIP_LIB=$R/$V/libcurlhip_pred_hue.so python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -x -q -m gpu -k "golden or fullsize or hsv or properties or tie" > $O/exp16_tests.log 2>&1; tail -1 $O/exp16_tests.log
grep -q " passed" $O/exp16_tests.log || exit 1
echo "== layer: A = default, B = predicated hue terms" > $O/exp16_pred_hue.log
LAUNCHES=400 ROUNDS=21 python3 tools/ab.py $L $V/libcurlhip_pred_hue.so layer 2>&1 | grep -v amdgpu >> $O/exp16_pred_hue.log || exit 1
echo "== hsv_stage (not UNIT: unaffected by construction)" >> $O/exp16_pred_hue.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=9 python3 tools/ab.py $L $V/libcurlhip_pred_hue.so hsv_stage 2>&1 | grep -v amdgpu >> $O/exp16_pred_hue.log || exit 1
cat $O/exp16_pred_hue.log
