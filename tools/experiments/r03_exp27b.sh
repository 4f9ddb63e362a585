#!/bin/bash
# round 3, experiment 27b: the occupancy cap again where 27 showed a gain -- HSV stage, Lab stage at two groups per lane,
# the layer at two groups per lane (each wave then has twice the loads in flight, so fewer waves might do)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
rm -f $O/exp27b_occupancy_cap.log
run() { echo "== $1: A flags $2, B flags $3" >> $O/exp27b_occupancy_cap.log
FULL_ONLY=1 LAUNCHES=200 ROUNDS=9 FLAGS_A=$2 FLAGS_B=$3 python3 tools/ab.py $L $L $1 2>&1 | grep -v amdgpu >> $O/exp27b_occupancy_cap.log || exit 1; }
for k in 5 6 7; do run hsv_stage 0 $((k << 19)); done
for k in 4 5 6 7; do run lab_stage 0 $(( (k << 19) | 0x200 )); done
run lab_stage 0 0x200
for k in 4 5 6 7; do run layer 0 $(( (k << 19) | 0x200 )); done
run lab_stage 0 $((7 << 19))
run rgb2lab 0 $((7 << 19))
run rgb2lab 0 $((3 << 19))
cat $O/exp27b_occupancy_cap.log
