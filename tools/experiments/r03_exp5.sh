#!/bin/bash
# round 3, experiment 5: tile shape of the fused layer again, with the round-3 code and the paired statistic over long
# windows (FULL_ONLY: the power-capped kernel only; 400 launches = 88 ms per window, 31 rounds)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so
for f in 0x200 0x800 0x1000 0xA00; do
  echo "== FLAGS_B=$f (0x200: two float4 groups per lane; 0x800: 128-thread blocks; 0x1000: 64-thread; 0xA00: both)" >> $O/exp5_tile_shape.log
  FULL_ONLY=1 LAUNCHES=400 ROUNDS=31 FLAGS_B=$f python3 tools/ab.py $L $L layer 2>&1 | grep -v amdgpu >> $O/exp5_tile_shape.log || exit 1
done
cat $O/exp5_tile_shape.log
