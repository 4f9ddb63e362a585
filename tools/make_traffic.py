"""profiles/traffic_<tag>.json from the PMC passes of tools/prof.sh (FETCH_SIZE / WRITE_SIZE, KiB).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half the bytes of a wide coalesced
streaming read (16 B/lane) -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import main as summarise  # noqa: E402

tag = sys.argv[1]
out = {}
for workload, frag in (("layer", "OpLayer"), ("lab_stage", "OpLabStage"), ("hsv_stage", "OpHsvStage"), ("rgb_only", "OpAdjust3"),
                       ("trispace", "OpTriSpace"), ("layer_bwd", "layer_bwd_kernel")):
    root = os.path.join("gpurun_out", f"prof_{tag}_{workload}")
    if not os.path.isdir(root):
        continue
    r = summarise(root, frag)
    if "FETCH_SIZE" not in r or "WRITE_SIZE" not in r:
        continue
    rd, wr = 2.0 * r["FETCH_SIZE"] * 1024, r["WRITE_SIZE"] * 1024
    out[frag] = {"kernel": r.get("kernel"), "fetch_size_kib_raw": r["FETCH_SIZE"], "write_size_kib": r["WRITE_SIZE"],
                 "read_bytes_corrected_x2": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                 "kernel_avg_us_trace_pass": r.get("avg_us"), "launches": r.get("launches")}
json.dump(out, open(os.path.join("profiles", f"traffic_{tag}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
