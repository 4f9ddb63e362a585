#!/bin/bash
# round 3, the records of the final build: GPU suite, smoke, the default bench line, the bs256 line, every operator
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 -m pytest tests -q -m gpu > $O/final_pytest_gpu.log 2>&1; tail -2 $O/final_pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1; tail -1 $O/final_smoke.log
python3 bench.py > $O/final_bench_1gpu.json 2> $O/final_bench_1gpu.err || { tail -5 $O/final_bench_1gpu.err; exit 1; }
echo bench done
python3 bench.py --batch 256 --steps 100 --warmup 10 > $O/final_bench_bs256.json 2> $O/final_bench_bs256.err || { tail -5 $O/final_bench_bs256.err; exit 1; }
echo bs256 done
python3 tools/all_ops_bench.py 100 > $O/final_all_ops_bench.log 2>&1 || { tail -5 $O/final_all_ops_bench.log; exit 1; }
python3 - <<'PY'
import json
for f in ("final_bench_1gpu.json", "final_bench_bs256.json"):
    d = json.loads(open("gpurun_out/r03/" + f).read().strip().splitlines()[-1])
    print(f, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "batch", d["config"]["batch_per_gpu"])
    for w in d.get("other_workloads", []):
        r = w["roofline"]
        print(f'  {w["workload"][:56]:56s} {w["device_ms_per_step"]*1e3:8.1f} us {r["bound"]} {r["frac"]:.3f} / {r["secondary"]["frac"]:.3f}')
PY
