#!/bin/bash
# round 3, experiment 17: experiment 15 again with the SCC clobber declared on the predicated asm (the whole GPU suite first)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests -q -m gpu > $O/exp17_tests.log 2>&1; tail -1 $O/exp17_tests.log
grep -q " passed" $O/exp17_tests.log || exit 1
grep -q " failed" $O/exp17_tests.log && exit 1
echo "== layer, random pixels: A = v_cndmask selects (prev), B = predicated overwrites" > $O/exp17_predicated_select_scc.log
LAUNCHES=400 ROUNDS=21 python3 tools/ab.py $V/libcurlhip_prev.so $L layer 2>&1 | grep -v amdgpu >> $O/exp17_predicated_select_scc.log || exit 1
cat $O/exp17_predicated_select_scc.log
