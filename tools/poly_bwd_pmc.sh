#!/bin/bash
# Counter passes of the polynomial backward alone (tools/poly_bwd_run.py): tools/poly_bwd_pmc.sh <tag>  -> gpurun_out/r04/pmc_poly_<tag>/
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04/pmc_poly_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/tools/poly_bwd_run.py > $OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -- python3 $R/tools/poly_bwd_run.py > $OUT/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $R/tools/poly_bwd_run.py > $OUT/f.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $R/tools/poly_bwd_run.py > $OUT/w.log 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for sub in "abfw":
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if "trispace" in k:
                print(k, {c: round(sum(x[3:]) / len(x[3:])) for c, x in v.items()})
    for f in glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"][:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in d.items():
            if "trispace" in k:
                print(sub, k, "avg us", round(sum(v[5:]) / len(v[5:]), 1))
PY
