#!/bin/bash
# round 3, experiment 27d: the light streaming kernels under an occupancy cap, one and two float4 groups per lane
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
rm -f $O/exp27d_occupancy_light_kernels.log
run() { echo "== $1: A = default, B = $4" >> $O/exp27d_occupancy_light_kernels.log
FULL_ONLY=1 LAUNCHES=200 ROUNDS=7 FLAGS_A=$2 FLAGS_B=$3 python3 tools/ab.py $L $L $1 2>&1 | grep -v amdgpu >> $O/exp27d_occupancy_light_kernels.log || exit 1; }
for w in rgb2lab adjust_rgb hsv_stage; do
for u in 1 2; do
for k in 0 2 3 4; do
run $w 0 $(( (k << 19) | (u << 8) )) "U=$u, $k workgroups per CU (0 = no cap)"
done; done; done
cat $O/exp27d_occupancy_light_kernels.log
