#!/bin/bash
# rocprofv3 passes for one bench workload (run on the GPU box from the repo root):
#   tools/prof.sh <workload> <tag>
# pass A: kernel trace + stats; pass B: SQ counters; pass C: FETCH_SIZE; pass D: WRITE_SIZE
# (counters in their own runs, no sys/hip traces -- see the task notes on this pool)
set -u
W=${1:-layer}; TAG=${2:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${W}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload $W --steps 1000 --warmup 50 --no-extras"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
