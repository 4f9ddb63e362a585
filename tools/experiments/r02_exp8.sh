#!/bin/bash
# Round-2 experiment 8 (GPU box): polynomial kernel -- raw coefficients staged in LDS, pixel loads before the staging
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp8
mkdir -p $OUT
cd $R
V=curl_amd/lib/variants
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "trispace or poly or infer or row" > $OUT/pytest_gpu_poly.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu_poly.log
tail -3 $OUT/pytest_gpu_poly.log
timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_poly_stage_r1.so $V/libcurlhip_base.so trispace > $OUT/ab_stage_r1_vs_base_trispace.log 2>&1
tail -4 $OUT/ab_stage_r1_vs_base_trispace.log
SETTLE=150 timeout -k 10 300 python3 tools/stamp.py trispace trispace_nomem > $OUT/stamp_trispace.log 2>&1
tail -2 $OUT/stamp_trispace.log
timeout -k 10 300 python3 tools/poly_bench.py > $OUT/poly_bench.log 2>&1
tail -9 $OUT/poly_bench.log
echo "exit $?" > $OUT/done.txt
