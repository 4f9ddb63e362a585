"""Per-kernel table of a rocprofv3 --pmc run: tools/pmc_table.py <dir> [kernel substring]
Averages each counter over the second half of the launches of every kernel; adds VALU2 per instruction and the
issue quads per instruction where the counters are there."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
frag = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    c = {n: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for n, v in acc[k].items()}
    d = dur.get(k, [0.0])
    d = d[len(d) // 2:]
    line = f"{k[:70]:70s} us {sum(d) / len(d):9.1f}"
    for n in sorted(c):
        line += f"  {n.replace('SQ_', '')} {c[n]:.4g}"
    if "SQ_INSTS_VALU" in c and "SQ_ACTIVE_INST_VALU2" in c:
        line += f"  | VALU2/inst {c['SQ_ACTIVE_INST_VALU2'] / c['SQ_INSTS_VALU']:.3f}"
    if "SQ_INSTS_VALU" in c and "SQ_CYCLES" in c:
        line += f"  cyc/inst/SIMD {c['SQ_CYCLES'] / 32 * 1024 / c['SQ_INSTS_VALU']:.2f}"
    print(line)
