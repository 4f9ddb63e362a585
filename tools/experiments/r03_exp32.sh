#!/bin/bash
# round 3, experiment 32 (the mask-first path as a kernel variant of its own): CURL_F_MASK_FIRST as a run-time flag.  (1) the GPU suite; (2) the flag's code in the kernel costs the
# default path nothing (before_mask_first build = the commit before); (3) forward with the flag on all-ones and disk masks;
# (4) the backward: dead-wave shortcut (always on) and the flag, all-ones and disk masks
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; P=curl_amd/lib/variants/libcurlhip_before_mask_first.so
python3 -m pytest tests -x -q -m gpu > $O/exp32_tests.log 2>&1; tail -1 $O/exp32_tests.log
grep -q " passed" $O/exp32_tests.log || exit 1
grep -q " failed" $O/exp32_tests.log && exit 1
rm -f $O/exp32_mask_first_variant.log
run() { echo "== $1 ($2): A = $4, B = $5" >> $O/exp32_mask_first_variant.log
env $3 FULL_ONLY=1 LAUNCHES=400 ROUNDS=11 python3 tools/ab.py $6 $7 $1 2>&1 | grep -v amdgpu >> $O/exp32_mask_first_variant.log || exit 1; }
run layer "all-ones mask" "X=0" "the commit before" "this build, no flag" $P $L
run lab_stage "all-ones mask" "X=0" "the commit before" "this build, no flag" $P $L
run layer "all-ones mask" "FLAGS_B=0x400000" "no flag" "CURL_F_MASK_FIRST" $L $L
run layer "disk, 70 %" "MASK=disk FLAGS_B=0x400000" "no flag" "CURL_F_MASK_FIRST" $L $L
run lab_stage "disk, 70 %" "MASK=disk FLAGS_B=0x400000" "no flag" "CURL_F_MASK_FIRST" $L $L
run layer "disk, 40 %" "MASK=disk DISK_R2=0.5 FLAGS_B=0x400000" "no flag" "CURL_F_MASK_FIRST" $L $L
run hsv_stage "all-ones mask" "X=0" "the commit before" "this build (mask-first by default)" $P $L
run hsv_stage "disk, 70 %" "MASK=disk" "the commit before" "this build (mask-first by default)" $P $L
run layer_bwd "all-ones mask" "X=0" "the commit before" "this build, no flag" $P $L
run layer_bwd "disk, 70 %" "MASK=disk" "the commit before" "this build, no flag (dead-wave shortcut)" $P $L
run layer_bwd "all-ones mask" "FLAGS_B=0x400000" "no flag" "CURL_F_MASK_FIRST" $L $L
run layer_bwd "disk, 70 %" "MASK=disk FLAGS_B=0x400000" "no flag" "CURL_F_MASK_FIRST" $L $L
run layer_bwd "disk, 40 %" "MASK=disk DISK_R2=0.5 FLAGS_B=0x400000" "no flag" "CURL_F_MASK_FIRST" $L $L
cat $O/exp32_mask_first_variant.log
