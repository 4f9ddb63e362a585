#!/bin/bash
# Round-2 experiment 5 (GPU box): polynomial model priorities; sustained A/B of the layer (scalar vs packed helpers)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp5
mkdir -p $OUT
cd $R
V=curl_amd/lib/variants
for v in poly1 poly1_t2 poly1_pk; do
  timeout -k 10 300 python3 tools/ab.py $V/libcurlhip_base.so $V/libcurlhip_$v.so trispace > $OUT/ab_base_vs_${v}_trispace.log 2>&1 || exit 1
  tail -4 $OUT/ab_base_vs_${v}_trispace.log
done
ROUNDS=21 timeout -k 10 400 python3 tools/ab.py $V/libcurlhip_base.so $V/libcurlhip_pk_tp1.so layer > $OUT/ab_base_vs_pk_tp1_layer_long.log 2>&1
tail -4 $OUT/ab_base_vs_pk_tp1_layer_long.log
PMC="SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS"
for v in base poly1; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_tri_$v -- python3 $R/tools/run_variant.py $v trispace 100 > $OUT/pmc_tri_$v.log 2> $OUT/pmc_tri_$v.err) || exit 1
  python3 tools/pmc_table.py $OUT/pmc_tri_$v OpTriSpace > $OUT/pmc_tri_${v}_table.txt
done
echo "exit $?" > $OUT/done.txt
