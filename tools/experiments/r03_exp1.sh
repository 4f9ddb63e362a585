#!/bin/bash
# round 3, experiment 1 (one box, one session): what does the knot-prep launch + its kernel boundary cost per step?
#   A = the product call (prep + main), B = the same library with CURL_F_DIAG_SKIP_PREP (main only, workspace reused)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
FLAGS_B=0x20000 ROUNDS=15 python3 tools/ab.py $L $L layer > $O/exp1_skip_prep_ab.log 2>&1 || exit 1
cat $O/exp1_skip_prep_ab.log
python3 tools/ab.py $L $L layer_bwd > $O/exp1_layer_bwd_base.log 2>&1 || exit 1
B=8 python3 tools/ab.py $L $L layer_bwd >> $O/exp1_layer_bwd_base.log 2>&1 || exit 1
cat $O/exp1_layer_bwd_base.log
