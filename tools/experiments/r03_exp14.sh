#!/bin/bash
# round 3, experiment 14: XCD-contiguous tile mapping (CURL_F_TUNE_XCD = 2: each XCD walks one contiguous eighth of an
# image's tiles) against the plain mapping; the copy probe said -4 % for one float4 group per lane, +0.8 % for two
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or shapes or misaligned or bs32" > $O/exp14_tests.log 2>&1; tail -1 $O/exp14_tests.log
for w in lab_stage layer hsv_stage rgb2lab adjust_rgb; do
echo "== $w: B = XCD-contiguous" >> $O/exp14_xcd_mapping.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 FLAGS_B=0x4000 python3 tools/ab.py $L $L $w 2>&1 | grep -v amdgpu >> $O/exp14_xcd_mapping.log || exit 1
done
echo "== lab_stage, two groups per lane: A plain, B XCD-contiguous" >> $O/exp14_xcd_mapping.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 FLAGS_A=0x200 FLAGS_B=0x4200 python3 tools/ab.py $L $L lab_stage 2>&1 | grep -v amdgpu >> $O/exp14_xcd_mapping.log || exit 1
cat $O/exp14_xcd_mapping.log
