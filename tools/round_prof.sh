#!/bin/bash
# rocprofv3 passes of one round's build: tools/round_prof.sh <tag> <workload> [<workload> ...]   (tag: r04, ...)
# per workload: kernel trace + stats, SQ counters, FETCH_SIZE, WRITE_SIZE -- each its own run (tools/prof.sh);
# then profiles/traffic_<tag>.json from the counter passes (tools/make_traffic.py)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
for w in "$@"; do
  bash tools/prof.sh $w $TAG > gpurun_out/prof_${TAG}_$w.log 2>&1 || exit 1
  python3 tools/kstats.py gpurun_out/prof_${TAG}_$w/trace | head -6
done
python3 tools/make_traffic.py $TAG
