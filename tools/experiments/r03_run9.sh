#!/bin/bash
# rehearsal of the multi-rank bench on the 1-GPU box: 2 ranks share the GPU over gloo (launcher, sharding, reduce, DDP train step)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python3 bench.py --gpus 2 --steps 50 --warmup 5 > $O/run9_bench_2rank_rehearsal.json 2> $O/run9_bench_2rank.err || { tail -20 $O/run9_bench_2rank.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/run9_bench_2rank_rehearsal.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "ranks_seen", "backend", "gpus_visible", "ms_per_step")}, d.get("rehearsal"))
print(json.dumps(d["end_to_end"].get("train_step"))[:400])
PY
