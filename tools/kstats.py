"""Print a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <dir-or-csv> [substring]"""
import csv
import glob
import os
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = max(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(p)):
    if sub in r["Name"]:
        print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f}  "
              f"min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
