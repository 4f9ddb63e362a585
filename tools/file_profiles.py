"""File the rocprofv3 passes of tools/round_prof.sh (merged back under gpurun_out/prof_<tag>_<workload>/) into profiles/<tag>/:
<workload>_kernel_stats.csv, <workload>_pmc_summary.json, per_launch_us.json, ../traffic_<tag>.json.

    python tools/file_profiles.py r03 layer lab_stage hsv_stage layer_bwd
"""
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from pmc_summary import main as summarise  # noqa: E402

# the workload's dominant kernel (the summary and the per-launch series are its; traffic_<tag>.json sums a call's kernels)
FRAG = {"layer": "OpLayer", "layer_8bit": "OpLayer", "layer_disk": "OpLayer", "layer_disk_mask_first": "OpLayer", "layer_u8": "OpLayer",
        "trispace_u8": "OpTriSpace", "lab_stage": "OpLabStage", "hsv_stage": "OpHsvStage", "rgb_only": "OpAdjust3",
        "trispace": "OpTriSpace", "layer_bwd": "layer_bwd_kernel", "layer_bwd_crop": "layer_bwd_kernel",
        "layer_bwd_knots": "layer_bwd_kernel", "layer_bwd_crop_knots": "layer_bwd_kernel",
        "loss_fwd": "loss_terms_kernel", "loss_bwd": "loss_terms_bwd_kernel", "train_fwd": "layer_loss_kernel",
        "train_fwd_two_calls": "loss_terms_kernel", "trispace_bwd": "trispace_coef_grad"}
tag, workloads = sys.argv[1], sys.argv[2:]
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
per_launch_path = os.path.join(dst, "per_launch_us.json")
per_launch = json.load(open(per_launch_path)) if os.path.exists(per_launch_path) else {}
for w in workloads:
    root = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{w}")
    newest = lambda pat: max(glob.glob(os.path.join(root, "trace", "**", pat), recursive=True), key=os.path.getmtime)  # noqa: E731
    shutil.copy(newest("*_kernel_stats.csv"), os.path.join(dst, f"{w}_kernel_stats.csv"))
    json.dump(summarise(root, FRAG[w]), open(os.path.join(dst, f"{w}_pmc_summary.json"), "w"), indent=1)
    us = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
          for r in csv.DictReader(open(newest("*_kernel_trace.csv"))) if FRAG[w] in r["Kernel_Name"]]
    us = [d for _, d in sorted(us)]
    if not us:
        continue
    tail = us[200:] if len(us) > 400 else us
    per_launch[w] = {"launches": len(us), "avg_all_us": round(statistics.mean(us), 2), "avg_after_200_us": round(statistics.mean(tail), 2),
                     "median_after_200_us": round(statistics.median(tail), 2), "min_us": round(min(us), 2),
                     "first_100_avg_us": round(statistics.mean(us[:100]), 2)}
    print(w, per_launch[w])
json.dump(per_launch, open(per_launch_path, "w"), indent=1)
subprocess.check_call([sys.executable, os.path.join(HERE, "make_traffic.py"), tag], cwd=ROOT, stdout=subprocess.DEVNULL)
