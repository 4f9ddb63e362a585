#!/bin/bash
# Round-2 experiment 2 (GPU box): pairing matrix of the VALU (SQ_ACTIVE_INST_VALU2), s_setprio, VGPR constants.
# (Build names follow tools/variants.py as it is NOW: r1 = the round-1 code, which was "base" when profiles/r02/
# ab_r1_code_experiments.log was taken; r1_fast1 was "prio1", r1_vconst "vconst".)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_exp2
mkdir -p $OUT
cd $R
PMC="SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU"
timeout -k 10 300 ./tools/ubench/issue_pair 8 > $OUT/issue_pair.log 2>&1 &&
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_pair -- $R/tools/ubench/issue_pair 8 > $OUT/pmc_pair.log 2> $OUT/pmc_pair.err) &&
python3 tools/pmc_table.py $OUT/pmc_pair > $OUT/pmc_pair_table.txt &&
for v in r1_fast1 r1_vconst; do
  timeout -k 10 300 python3 tools/ab.py curl_amd/lib/variants/libcurlhip_r1.so curl_amd/lib/variants/libcurlhip_$v.so layer > $OUT/ab_r1_vs_$v.log 2>&1 || exit 1
done &&
for v in r1 r1_fast1 r1_vconst; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$v -- python3 $R/tools/run_variant.py $v layer 200 > $OUT/pmc_$v.log 2> $OUT/pmc_$v.err) || exit 1
  python3 tools/pmc_table.py $OUT/pmc_$v OpLayer > $OUT/pmc_${v}_table.txt
done
echo "exit $?" > $OUT/done.txt
ls $OUT
