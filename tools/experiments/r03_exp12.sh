#!/bin/bash
# round 3, experiment 12: tile shape of the fused HSV stage (63 instructions per pixel: between the copy-like kernels,
# which run two float4 groups per lane, and the Lab stage, which runs one)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
for f in 0x200 0x400 0x800; do
echo "== hsv_stage FLAGS_B=$f (0x200: two float4 groups per lane, 0x400: four, 0x800: 128-thread blocks)" >> $O/exp12.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 FLAGS_B=$f python3 tools/ab.py $L $L hsv_stage 2>&1 | grep -v amdgpu >> $O/exp12.log || exit 1
done
echo "== lab_stage FLAGS_B=0x200" >> $O/exp12.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 FLAGS_B=0x200 python3 tools/ab.py $L $L lab_stage 2>&1 | grep -v amdgpu >> $O/exp12.log || exit 1
cat $O/exp12.log
