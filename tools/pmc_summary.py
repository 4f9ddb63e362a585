"""Summarise rocprofv3 CSVs written by tools/prof.sh: per-kernel averages of duration and counters."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load_counters(d):
    out = defaultdict(lambda: defaultdict(list))
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:  # the newest pass only (gpurun_out/ accumulates over calls)
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def load_trace(d):
    out = defaultdict(list)
    files = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def main(root, frag):
    res = {}
    tr = load_trace(os.path.join(root, "trace"))
    for k, v in tr.items():
        if frag in k:
            v = v[10:] if len(v) > 20 else v  # drop warm-up launches
            res["kernel"] = k
            res["launches"] = len(v)
            res["avg_us"] = sum(v) / len(v)
            res["min_us"], res["max_us"] = min(v), max(v)
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        c = load_counters(os.path.join(root, sub))
        t = load_trace(os.path.join(root, sub))
        for k, cs in c.items():
            if frag in k:
                for name, vals in cs.items():
                    vals = vals[10:] if len(vals) > 20 else vals
                    res[name] = sum(vals) / len(vals)
                v = t.get(k, [])
                v = v[10:] if len(v) > 20 else v
                if v:
                    res[sub + "_avg_us"] = sum(v) / len(v)
    return res


if __name__ == "__main__":
    r = main(sys.argv[1], sys.argv[2])
    print(json.dumps(r, indent=1))
