#!/bin/bash
# round 3, experiment 3: experiment 2(a) again with 61 rounds and the paired per-round statistic (the unpaired medians of
# 15 rounds contradicted each other: a power-capped kernel drifts by +-2 % within a session)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
ROUNDS=61 python3 tools/ab.py $V/libcurlhip_r2_hsv.so $L layer > $O/exp3_hsv_trapezoid_ab_paired.log 2>&1 || exit 1
FLAGS_B=0x20000 ROUNDS=61 python3 tools/ab.py $L $L layer > $O/exp3_skip_prep_ab_paired.log 2>&1 || exit 1
cat $O/exp3_hsv_trapezoid_ab_paired.log $O/exp3_skip_prep_ab_paired.log
