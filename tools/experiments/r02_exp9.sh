#!/bin/bash
# per-kernel times of the polynomial backward's three passes, column-strip build vs the flat-tile build (32 x 256 x 256)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bwd_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base bwd_plain; do
  SHAPES=crop ROUNDS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $R/tools/bwd_ab.py $v > $OUT/$v.log 2> $OUT/$v.err
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  echo "## $v"; cut -d, -f1-4 "$f" | grep -i "trispace\|Name" 
done
