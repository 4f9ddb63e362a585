#!/bin/bash
# round 3, experiment 27g: the GPU suite on the build with per-operator resident-workgroup defaults, then the pointwise
# kernels outside the streaming skeleton (PSNR, byte edges, CURLLoss terms) at 2 / 4 / 6 workgroups per CU
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests -x -q -m gpu > $O/exp27g_tests.log 2>&1; tail -1 $O/exp27g_tests.log
grep -q " passed" $O/exp27g_tests.log || exit 1
grep -q " failed" $O/exp27g_tests.log && exit 1
rm -f $O/exp27g_occupancy_aux_kernels.log
for w in psnr to_u8 from_u8 loss_fwd loss_bwd; do
for k in 2 4 6; do
echo "== $w: A = no cap (aux_res0 build), B = $k workgroups per CU" >> $O/exp27g_occupancy_aux_kernels.log
FULL_ONLY=1 LAUNCHES=200 ROUNDS=7 python3 tools/ab.py $V/libcurlhip_aux_res0.so $V/libcurlhip_aux_res$k.so $w 2>&1 | grep -v amdgpu >> $O/exp27g_occupancy_aux_kernels.log || exit 1
done; done
cat $O/exp27g_occupancy_aux_kernels.log
