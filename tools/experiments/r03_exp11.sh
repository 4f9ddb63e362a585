#!/bin/bash
# round 3, experiment 11: the mask's load first (default) against prev (round-2 prologue) and against mask-load-last
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
for w in hsv_stage lab_stage layer; do
echo "== $w: A = prev (round-2 prologue), B = default (mask dword, its load first, diagnostics out of line)" >> $O/exp11.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_prev.so $L $w 2>&1 | grep -v amdgpu >> $O/exp11.log || exit 1
echo "== $w: A = mask load last, B = default" >> $O/exp11.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_mask_last.so $L $w 2>&1 | grep -v amdgpu >> $O/exp11.log || exit 1
done
cat $O/exp11.log
