#!/bin/bash
# round 3, experiment 25: the saddr address form (stream.inc at()) in the other streaming kernels -- fused backward,
# CURLLoss terms forward / backward, masked PSNR, the byte edges -- against the 64-bit form (addr64 build)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 -m pytest tests -x -q -m gpu > $O/exp25_tests.log 2>&1; tail -1 $O/exp25_tests.log
grep -q " passed" $O/exp25_tests.log || exit 1
grep -q " failed" $O/exp25_tests.log && exit 1
rm -f $O/exp25_saddr_other_kernels.log
for w in layer_bwd loss_fwd loss_bwd psnr to_u8 from_u8; do
echo "== $w: A = addr64, B = default (saddr + 32-bit offset)" >> $O/exp25_saddr_other_kernels.log
FULL_ONLY=1 LAUNCHES=200 ROUNDS=11 python3 tools/ab.py $V/libcurlhip_addr64.so $L $w 2>&1 | grep -v amdgpu >> $O/exp25_saddr_other_kernels.log || exit 1
done
cat $O/exp25_saddr_other_kernels.log
