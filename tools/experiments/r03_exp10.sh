#!/bin/bash
# round 3, experiment 10: which half of experiment 9 slowed the memory-bound stages?  A = mask dword only (diagnostics
# branch as the compiler places it), B = default (diagnostics out of line); and prev (neither) against A
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
for w in hsv_stage lab_stage layer; do
echo "== $w: A = prev (round-2 prologue), B = mask dword only" >> $O/exp10.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_prev.so $V/libcurlhip_nomem_inline.so $w 2>&1 | grep -v amdgpu >> $O/exp10.log || exit 1
echo "== $w: A = mask dword only, B = default (+ diagnostics branch out of line)" >> $O/exp10.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_nomem_inline.so $L $w 2>&1 | grep -v amdgpu >> $O/exp10.log || exit 1
done
cat $O/exp10.log
