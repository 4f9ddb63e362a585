#!/bin/bash
# round 3, experiment 13: lazy selects working in place (no copies on the skipping side) against the previous commit
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or fullsize or hsv_stage or bs32 or mask" > $O/exp13_tests.log 2>&1; tail -1 $O/exp13_tests.log
grep -q " passed" $O/exp13_tests.log || exit 1
for w in layer hsv_stage; do
echo "== $w" >> $O/exp13_lazy_in_place.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=21 python3 tools/ab.py $V/libcurlhip_prev.so $L $w 2>&1 | grep -v amdgpu >> $O/exp13_lazy_in_place.log || exit 1
done
cat $O/exp13_lazy_in_place.log
