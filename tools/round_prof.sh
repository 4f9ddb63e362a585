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
# SUMMARISE=1 (a GPU call hands back at most 64 MiB of gpurun_out/, one workload's four passes are 12 MiB): file the summaries
# on the box (tools/file_profiles.py -> profiles/<tag>/, profiles/traffic_<tag>.json), hand THOSE back under
# gpurun_out/filed_<tag>/ and drop the raw passes
if [ "${SUMMARISE:-0}" = 1 ]; then
  python3 tools/file_profiles.py $TAG "$@" || exit 1
  mkdir -p gpurun_out/filed_$TAG && cp -r profiles/$TAG/. gpurun_out/filed_$TAG/ && cp profiles/traffic_$TAG.json gpurun_out/filed_$TAG/
  rm -rf gpurun_out/prof_${TAG}_*
fi
