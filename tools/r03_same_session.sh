#!/bin/bash
# round 3: the bench line and the rocprofv3 kernel statistics of ONE session on ONE box (boxes of the pool differ by 2.5 %),
# with the 3-plane copy ceiling measured in between
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03; mkdir -p $O
python3 bench.py > $O/same_session_bench_1gpu.json 2> $O/same_session_bench.err || { tail -5 $O/same_session_bench.err; exit 1; }
echo bench done
bash tools/r03_prof.sh layer lab_stage hsv_stage layer_bwd > $O/same_session_prof.log 2>&1 || { tail -5 $O/same_session_prof.log; exit 1; }
grep -E "stream_kernel|layer_bwd_kernel|sustained" $O/same_session_prof.log
python3 bench.py --no-extras --steps 1000 > $O/same_session_bench_after.json 2>> $O/same_session_bench.err
python3 - <<'PY'
import json
for f in ("same_session_bench_1gpu.json", "same_session_bench_after.json"):
    d = json.loads(open("gpurun_out/r03/" + f).read().strip().splitlines()[-1])
    print(f, "ms/step", round(d["ms_per_step"], 4), "dev", round(d["device_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4))
PY
