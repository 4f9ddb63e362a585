#!/bin/bash
# Round-2 experiment 3 (GPU box): issue priority of the unpairable runs (s_setprio), packed vs scalar helpers.
# (Build names follow tools/variants.py as it is NOW.  When profiles/r02/issue_priority_ab.log was taken the round-1 code
# was "base", pk_* were "ps_*", nopk_t0 was "nopk" and today's base -- the default -- was "nopk_t1".)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp3
mkdir -p $OUT
cd $R
PMC="SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
VARS="pk_t1 pk_tp1 pk_tp3 pk_p1 nopk_t0 base"
for v in $VARS; do
  timeout -k 10 300 python3 tools/ab.py curl_amd/lib/variants/libcurlhip_r1.so curl_amd/lib/variants/libcurlhip_$v.so layer > $OUT/ab_r1_vs_$v.log 2>&1 || exit 1
  tail -4 $OUT/ab_r1_vs_$v.log
done &&
for v in $VARS; do
  timeout -k 10 300 python3 tools/ab.py curl_amd/lib/variants/libcurlhip_r1.so curl_amd/lib/variants/libcurlhip_$v.so lab_stage > $OUT/ab_lab_r1_vs_$v.log 2>&1 || exit 1
done &&
for v in r1 $VARS; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$v -- python3 $R/tools/run_variant.py $v layer 200 > $OUT/pmc_$v.log 2> $OUT/pmc_$v.err) || exit 1
  python3 tools/pmc_table.py $OUT/pmc_$v OpLayer > $OUT/pmc_${v}_table.txt
done
echo "exit $?" > $OUT/done.txt
