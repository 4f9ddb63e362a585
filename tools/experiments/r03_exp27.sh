#!/bin/bash
# round 3, experiment 27: resident workgroups per CU held to k = 3..7 (CURL_F_TUNE_OCC: 160 KB / k of unused LDS per workgroup;
# the default lets 8 waves per SIMD in) -- does a streaming kernel at the power cap gain from fewer waves in flight?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
rm -f $O/exp27_occupancy_cap.log
for w in layer lab_stage adjust_rgb; do
for k in 3 4 5 6 7; do
echo "== $w: A = no cap, B = $k workgroups per CU" >> $O/exp27_occupancy_cap.log
FULL_ONLY=1 LAUNCHES=200 ROUNDS=9 FLAGS_B=$((k << 19)) python3 tools/ab.py $L $L $w 2>&1 | grep -v amdgpu >> $O/exp27_occupancy_cap.log || exit 1
done
done
cat $O/exp27_occupancy_cap.log
