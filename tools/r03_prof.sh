#!/bin/bash
# rocprofv3 passes of the round-3 build: tools/r03_prof.sh <workload> [<workload> ...]; then the copy ceiling in the same session
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r03
for w in "$@"; do
  bash tools/prof.sh $w r03 > gpurun_out/prof_r03_$w.log 2>&1 || exit 1
  python3 tools/kstats.py gpurun_out/prof_r03_$w/trace | head -4
done
tools/ubench/copy3 > gpurun_out/r03/copy3_same_session.log 2>&1 || exit 1
grep -E "sustained|nt=1" gpurun_out/r03/copy3_same_session.log
python3 tools/make_traffic.py r03
