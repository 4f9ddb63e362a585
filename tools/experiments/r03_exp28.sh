#!/bin/bash
# round 3, experiment 28: is it the arithmetic or the memory wait that needs all eight waves per SIMD?  The layer with and
# without its memory instructions (CURL_F_DIAG_NO_MEM) at 4..7 resident workgroups per CU against the uncapped launch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
rm -f $O/exp28_layer_occupancy_valu_only.log
for k in 7 6 5 4 3; do
echo "== layer: A = no cap, B = $k workgroups per CU" >> $O/exp28_layer_occupancy_valu_only.log
LAUNCHES=200 ROUNDS=7 FLAGS_B=$((k << 19)) python3 tools/ab.py $L $L layer 2>&1 | grep -v amdgpu >> $O/exp28_layer_occupancy_valu_only.log || exit 1
done
cat $O/exp28_layer_occupancy_valu_only.log
