import csv, glob, sys, collections
root, frag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 2:]
    print(f"{k:28s} {sum(v)/len(v):16.1f}")
dur = []
for f in glob.glob(root + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
dur = dur[len(dur) // 2:]
print("avg_us (2nd half)", sum(dur) / len(dur))
