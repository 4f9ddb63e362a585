#!/bin/bash
# round 3, experiment 6: experiments 1 and 2(a) again with the tight protocol (400-launch windows, 31 paired rounds)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
echo "== B = the same library with CURL_F_DIAG_SKIP_PREP (no knot-prep launch)" > $O/exp6_tight.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=31 FLAGS_B=0x20000 python3 tools/ab.py $L $L layer 2>&1 | grep -v amdgpu >> $O/exp6_tight.log || exit 1
echo "== A = round-2 HSV code (two ramps per channel, masked s), B = default (trapezoid, unmasked s)" >> $O/exp6_tight.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=31 python3 tools/ab.py $V/libcurlhip_r2_hsv.so $L layer 2>&1 | grep -v amdgpu >> $O/exp6_tight.log || exit 1
cat $O/exp6_tight.log
