#!/bin/bash
# round 3, experiment 21: tile shape of the layer once more, after the predicated selects (666 instructions per wave)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
for f in 0x200 0x800; do
echo "== layer FLAGS_B=$f (0x200: two float4 groups per lane; 0x800: 128-thread blocks)" >> $O/exp21_tile_shape_again.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 FLAGS_B=$f python3 tools/ab.py $L $L layer 2>&1 | grep -v amdgpu >> $O/exp21_tile_shape_again.log || exit 1
done
cat $O/exp21_tile_shape_again.log
