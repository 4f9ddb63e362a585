#!/bin/bash
# round 3, experiment 2: (a) hsv2rgb as one trapezoid per channel + unmasked s (default) against the round-2 HSV code;
# (b) layer backward, pixels serialised + non-temporal accesses against the round-2 kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
ROUNDS=15 python3 tools/ab.py $V/libcurlhip_r2_hsv.so $L layer > $O/exp2_hsv_trapezoid_ab.log 2>&1 || exit 1
ROUNDS=9 python3 tools/ab.py $V/libcurlhip_r2_hsv.so $L hsv_stage >> $O/exp2_hsv_trapezoid_ab.log 2>&1 || exit 1
cat $O/exp2_hsv_trapezoid_ab.log
python3 tools/ab.py $V/libcurlhip_bwd_r2.so $L layer_bwd > $O/exp2_layer_bwd_ab.log 2>&1 || exit 1
B=8 python3 tools/ab.py $V/libcurlhip_bwd_r2.so $L layer_bwd >> $O/exp2_layer_bwd_ab.log 2>&1 || exit 1
cat $O/exp2_layer_bwd_ab.log
