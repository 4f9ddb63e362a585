#!/bin/bash
# round 3, experiment 19: the backward tapes' linear branches + constant derivatives as predicated overwrites vs two selects
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py -x -q -m gpu -k "backward or gradient or loss or trispace or poly or autograd" > $O/exp19_tests.log 2>&1; tail -1 $O/exp19_tests.log
grep -q " passed" $O/exp19_tests.log || exit 1
grep -q " failed" $O/exp19_tests.log && exit 1
echo "== 32 x 1500x1000" > $O/exp19_bwd_predicated.log
LAUNCHES=100 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_prev.so $L layer_bwd 2>&1 | grep -v amdgpu >> $O/exp19_bwd_predicated.log || exit 1
echo "== 32 x 256x256" >> $O/exp19_bwd_predicated.log
B=32 H=256 W=256 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_prev.so $L layer_bwd 2>&1 | grep -v amdgpu >> $O/exp19_bwd_predicated.log || exit 1
cat $O/exp19_bwd_predicated.log
