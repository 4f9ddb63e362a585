#!/bin/bash
# one PMC pass with an arbitrary counter list: tools/prof_pmc.sh <workload> <tag> <counter> [counter...]
W=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${TAG}_${W}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -- python3 $R/bench.py --workload $W --steps 60 --warmup 60 --no-extras > $OUT/run.json 2> $OUT/run.err
