#!/bin/bash
# round 3, experiment 26: u^2.4 of the fused stages taken directly, 2^(2.4 log2 u) (pow24_direct build), against the split
# form u*u * 2^(0.4 log2 u) (default): error distributions against the oracle first, then the parity tests ON THE DIRECT
# BUILD (CURL_HIP_LIB), then time.  Outcome: -1.0 % and one out-of-range test pixel at 1.11e-5 -- withdrawn.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
L=curl_amd/lib/libcurlhip.so; V=curl_amd/lib/variants
python3 tools/err_dist.py $L $V/libcurlhip_pow24_direct.so 2>&1 | grep -v amdgpu > $O/exp26_pow24_err.log || { tail -5 $O/exp26_pow24_err.log; exit 1; }
cat $O/exp26_pow24_err.log
CURL_HIP_LIB=$V/libcurlhip_pow24_direct.so python3 -m pytest tests/test_gpu_parity.py -q -m gpu > $O/exp26_tests.log 2>&1; tail -3 $O/exp26_tests.log
rm -f $O/exp26_pow24_direct.log
for w in layer lab_stage; do
echo "== $w: A = default (split form), B = pow24_direct" >> $O/exp26_pow24_direct.log
FULL_ONLY=1 LAUNCHES=400 ROUNDS=15 python3 tools/ab.py $L $V/libcurlhip_pow24_direct.so $w 2>&1 | grep -v amdgpu >> $O/exp26_pow24_direct.log || exit 1
done
cat $O/exp26_pow24_direct.log
