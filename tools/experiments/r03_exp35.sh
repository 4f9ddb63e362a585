#!/bin/bash
# round 3, experiment 35: workgroup order.  Default grid = (tiles per image, images): consecutive workgroups take consecutive
# 4 KB tiles of ONE image; grid_image_major build = (images, tiles): consecutive workgroups take the same tile of 32 images
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants/libcurlhip_grid_image_major.so
rm -f $O/exp35_grid_order.log
for w in layer lab_stage adjust_rgb rgb2lab; do
echo "== $w: A = default, B = grid_image_major" >> $O/exp35_grid_order.log
FULL_ONLY=1 LAUNCHES=300 ROUNDS=11 python3 tools/ab.py $L $V $w 2>&1 | grep -v amdgpu >> $O/exp35_grid_order.log || exit 1
done
cat $O/exp35_grid_order.log
