#!/bin/bash
# SQ counters of the polynomial backward's kernels at the training crop batch (one counter pass, its own run)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bwd_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPES=crop ROUNDS=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python3 $R/tools/bwd_ab.py base > $OUT/sq.log 2> $OUT/sq.err
python3 $R/tools/pmc_table.py $OUT/sq trispace
