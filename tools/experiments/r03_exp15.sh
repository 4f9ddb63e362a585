#!/bin/bash
# round 3, experiment 15: the linear branches of the Lab converters' selects as PREDICATED overwrites (exec narrowed to the
# lanes that take them: one VALU + two scalar instructions per value instead of two VALU) against the v_cndmask form (prev)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -x -q -m gpu > $O/exp15_tests.log 2>&1; tail -1 $O/exp15_tests.log
grep -q " passed" $O/exp15_tests.log || exit 1
for w in layer lab_stage; do
echo "== $w, random pixels" >> $O/exp15_predicated_select.log
LAUNCHES=400 ROUNDS=21 python3 tools/ab.py $V/libcurlhip_prev.so $L $w 2>&1 | grep -v amdgpu >> $O/exp15_predicated_select.log || exit 1
done
echo "== layer, pixels in [0.2, 1]" >> $O/exp15_predicated_select.log
IMG_LO=0.2 FULL_ONLY=1 LAUNCHES=400 ROUNDS=11 python3 tools/ab.py $V/libcurlhip_prev.so $L layer 2>&1 | grep -v amdgpu >> $O/exp15_predicated_select.log || exit 1
cat $O/exp15_predicated_select.log
