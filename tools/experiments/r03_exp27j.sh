#!/bin/bash
# round 3, experiment 27j: the sweeps again with reservations that give the residency they name (LDS comes in 1 280-byte
# granules, exp27i): every operator x {1, 2 groups per lane} x {no cap, 2..7}, then the pointwise kernels outside the skeleton
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
V=curl_amd/lib/variants
timeout -k 10 600 python3 tools/occ_sweep.py 2>&1 | grep -v amdgpu > $O/exp27j_occ_sweep_true_residency.log || exit 1
cat $O/exp27j_occ_sweep_true_residency.log
rm -f $O/exp27j_occupancy_aux_kernels.log
for w in to_u8 from_u8 psnr; do
for k in 2 3 4 5 6; do
echo "== $w: A = no cap (aux_res0 build), B = $k workgroups per CU" >> $O/exp27j_occupancy_aux_kernels.log
FULL_ONLY=1 LAUNCHES=200 ROUNDS=5 python3 tools/ab.py $V/libcurlhip_aux_res0.so $V/libcurlhip_aux_res$k.so $w 2>&1 | grep -v amdgpu | grep "per-round" >> $O/exp27j_occupancy_aux_kernels.log || exit 1
done; done
cat $O/exp27j_occupancy_aux_kernels.log
