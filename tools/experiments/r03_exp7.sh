#!/bin/bash
# round 3, experiment 7: threshold selects of the Lab converters skipped by waves that need none (default) against always
# executed (nolazy); on the benchmark's uniformly random pixels and on pixels without dark values
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or fullsize or converter or bs32 or twin or shapes or misaligned" > $O/exp7_tests.log 2>&1; tail -2 $O/exp7_tests.log
grep -q " passed" $O/exp7_tests.log || exit 1
for w in layer lab_stage; do
echo "== $w, uniformly random pixels (the benchmark)" >> $O/exp7_lazy_select.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=31 python3 tools/ab.py $V/libcurlhip_nolazy.so $L $w 2>&1 | grep -v amdgpu >> $O/exp7_lazy_select.log || exit 1
echo "== $w, pixels in [0.2, 1] (no dark values: every wave skips all four)" >> $O/exp7_lazy_select.log
IMG_LO=0.2 FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $V/libcurlhip_nolazy.so $L $w 2>&1 | grep -v amdgpu >> $O/exp7_lazy_select.log || exit 1
done
cat $O/exp7_lazy_select.log
