#!/bin/bash
# round 3, experiment 29: wait for the mask, skip the plane loads of fully masked-out wavefronts (mask_first build) against
# the default (all four loads issued at once): all-ones mask (the headline: pure cost), disk masks of ~70 % and ~40 % coverage
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
rm -f $O/exp29_mask_first.log
run() { echo "== $1 ($2): A = default, B = mask_first" >> $O/exp29_mask_first.log
env $3 FULL_ONLY=1 LAUNCHES=400 ROUNDS=11 python3 tools/ab.py $L $V/libcurlhip_mask_first.so $1 2>&1 | grep -v amdgpu >> $O/exp29_mask_first.log || exit 1; }
run layer "all-ones mask" "X=0"
run layer "disk, 70 %" "MASK=disk"
run layer "disk, 40 %" "MASK=disk DISK_R2=0.5"
run lab_stage "all-ones mask" "X=0"
run lab_stage "disk, 70 %" "MASK=disk"
run hsv_stage "all-ones mask" "X=0"
run hsv_stage "disk, 70 %" "MASK=disk"
cat $O/exp29_mask_first.log
