#!/bin/bash
# round 3, experiment 33: the knot-prep kernel samples the mask and the main kernel (variant 2) asks the workspace whether to
# test its mask first.  A = the product (mask-first only by flag), B = the mask_sample build.
# (1) GPU suite; (2) all-ones mask: what the sampling and the run-time branch cost; (3) disk masks: automatic against the flag
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/variants/libcurlhip_mask_sample.so; P=curl_amd/lib/libcurlhip.so
python3 -m pytest tests -x -q -m gpu > $O/exp33_tests.log 2>&1; tail -1 $O/exp33_tests.log
grep -q " passed" $O/exp33_tests.log || exit 1
grep -q " failed" $O/exp33_tests.log && exit 1
rm -f $O/exp33_mask_sample.log
run() { echo "== $1 ($2): A = $4, B = $5" >> $O/exp33_mask_sample.log
env $3 FULL_ONLY=1 LAUNCHES=400 ROUNDS=${RN:-21} python3 tools/ab.py $6 $7 $1 2>&1 | grep -v amdgpu >> $O/exp33_mask_sample.log || exit 1; }
run layer "all-ones mask" "X=0" "no sampling" "sampling (automatic)" $P $L
run lab_stage "all-ones mask" "X=0" "no sampling" "sampling (automatic)" $P $L
run hsv_stage "all-ones mask" "X=0" "no sampling" "sampling (automatic)" $P $L
RN=11
run layer "disk, 70 %" "MASK=disk" "no sampling" "sampling (automatic)" $P $L
run layer "disk, 70 %" "MASK=disk FLAGS_A=0x400000" "CURL_F_MASK_FIRST" "automatic" $L $L
run layer "disk, 40 %" "MASK=disk DISK_R2=0.5" "no sampling" "sampling (automatic)" $P $L
run lab_stage "disk, 70 %" "MASK=disk" "no sampling" "sampling (automatic)" $P $L
run hsv_stage "disk, 70 %" "MASK=disk" "no sampling" "sampling (automatic)" $P $L
cat $O/exp33_mask_sample.log
cd /tmp && export TMPDIR=/tmp
for lib in $R/$P $R/$L; do
CURL_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/exp33_prof_$(basename $lib .so) -- python3 $R/bench.py --workload layer --steps 1000 --warmup 50 --no-extras > /dev/null 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/exp33_prof_$(basename $lib .so) 2>/dev/null | head -2 >> $R/$O/exp33_mask_sample.log
done
tail -4 $R/$O/exp33_mask_sample.log
