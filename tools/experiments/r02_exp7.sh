#!/bin/bash
# Round-2 experiment 7 (GPU box): polynomial chains' first fma by op_sel (no v_mov broadcasts)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp7
mkdir -p $OUT
cd $R
V=curl_amd/lib/variants
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_poly_splat.so $V/libcurlhip_base.so trispace > $OUT/ab_poly_splat_vs_base_trispace.log 2>&1
tail -4 $OUT/ab_poly_splat_vs_base_trispace.log
timeout -k 10 300 python3 tools/poly_bench.py > $OUT/poly_bench.log 2>&1
tail -12 $OUT/poly_bench.log
echo "exit $?" > $OUT/done.txt
