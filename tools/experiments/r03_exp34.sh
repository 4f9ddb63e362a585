#!/bin/bash
# round 3, experiment 34: the knot-prep kernel without run-time indices into its kernel arguments (selects over the three
# segments instead of a.K[s]: those were global loads) against the commit before (before_mask_first build), via the layer step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; P=curl_amd/lib/variants/libcurlhip_before_mask_first.so
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -x -q -m gpu > $O/exp34_tests.log 2>&1; tail -1 $O/exp34_tests.log
grep -q " passed" $O/exp34_tests.log || exit 1
grep -q " failed" $O/exp34_tests.log && exit 1
rm -f $O/exp34_prep_selects.log
for w in layer lab_stage; do
echo "== $w: A = prep with a.K[s], B = prep with selects" >> $O/exp34_prep_selects.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=21 python3 tools/ab.py $P $L $w 2>&1 | grep -v amdgpu >> $O/exp34_prep_selects.log || exit 1
done
cd /tmp && export TMPDIR=/tmp
for lib in $R/$P $R/$L; do
CURL_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/exp34_prof_$(basename $lib .so) -- python3 $R/bench.py --workload layer --steps 1000 --warmup 50 --no-extras > /dev/null 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/exp34_prof_$(basename $lib .so) 2>/dev/null | head -2 >> $R/$O/exp34_prep_selects.log
done
cat $R/$O/exp34_prep_selects.log
