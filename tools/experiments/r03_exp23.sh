#!/bin/bash
# round 3, experiment 23: issue priorities once more with the round-3 instruction mix (666 per wave): none / level 1 (default) / level 3
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
for v in nopk_t0 nopk_t3; do
echo "== layer: A = default (transcendental runs at priority 1), B = $v" >> $O/exp23_priorities.log
LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $L $V/libcurlhip_$v.so layer 2>&1 | grep -v amdgpu >> $O/exp23_priorities.log || exit 1
done
cat $O/exp23_priorities.log
