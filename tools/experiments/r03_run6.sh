#!/bin/bash
# round 3, run 6: smoke (seeded) + the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/run6_smoke.log 2>&1 || { tail -5 $O/run6_smoke.log; exit 1; }
tail -1 $O/run6_smoke.log
python3 bench.py > $O/run6_bench.json 2> $O/run6_bench.err || { tail -5 $O/run6_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/run6_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for w in d["other_workloads"]:
    r = w["roofline"]
    print(f'{w["workload"][:60]:60s} {w["device_ms_per_step"]*1e3:8.1f} us  {r["bound"]} {r["frac"]:.3f}  secondary {r["secondary"]["frac"]:.3f}')
print(json.dumps(d["end_to_end"].get("train_step"))[:700])
PY
