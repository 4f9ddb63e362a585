#!/bin/bash
# per-kernel durations of every operator (tools/all_ops_bench.py) under rocprofv3 --kernel-trace --stats
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/all_ops_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/all_ops_bench.py 20 > $OUT/run.log 2> $OUT/run.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
cut -d, -f1-4 $OUT/kernel_stats.csv | head -60
