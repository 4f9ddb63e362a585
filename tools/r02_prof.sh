#!/bin/bash
# rocprofv3 passes of the round-2 default build: tools/r02_prof.sh <workload> [<workload> ...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for w in "$@"; do
  bash tools/prof.sh $w r02 > gpurun_out/prof_r02_$w.log 2>&1 || exit 1
  python3 tools/kstats.py gpurun_out/prof_r02_$w/trace | head -5
done
