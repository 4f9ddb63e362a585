"""profiles/traffic_<tag>.json from the PMC passes of tools/prof.sh (FETCH_SIZE / WRITE_SIZE in KiB, the SQ counters).

    python tools/make_traffic.py r04

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read
(16 B/lane) -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.  (Kernels that read narrower than 16 B per lane
-- the polynomial backward's passes read 4- and 8-byte elements -- are listed with the same correction and a note: for those
the doubled figure is an UPPER bound on the read bytes.)

One entry per bench workload (a call that launches several kernels is summed over them, launches per call counted from the
trace) and, for the single-kernel forward workloads, one per kernel-name fragment (what earlier rounds' files held).
valu_issue_util: VALU issue-slot utilisation of the workload's dominant kernel from the SQ pass of the same session,
(SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) quad-cycles x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)  (DESIGN.md 3c:
ACTIVE_INST_VALU counts a quad per VALU instruction, two per transcendental; VALU2 the quads that issued two at once)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import load_counters, load_trace  # noqa: E402

# workload -> kernel-name fragments of the kernels one call launches (knots_prep_kernel: 640 B per image, left out)
KERNELS = {
    "layer": ["OpLayer"], "layer_8bit": ["OpLayer"], "layer_disk": ["OpLayer"], "layer_disk_mask_first": ["OpLayer"],
    "layer_u8": ["OpLayer"], "trispace_u8": ["OpTriSpace"], "lab_stage": ["OpLabStage"], "hsv_stage": ["OpHsvStage"],
    "rgb_only": ["OpAdjust3"], "trispace": ["OpTriSpace"],
    "layer_bwd": ["layer_bwd_kernel", "knots_bwd_kernel"], "layer_bwd_crop": ["layer_bwd_kernel", "knots_bwd_kernel"],
    "layer_bwd_knots": ["layer_bwd_kernel", "knots_bwd_kernel"], "layer_bwd_crop_knots": ["layer_bwd_kernel", "knots_bwd_kernel"],
    "loss_fwd": ["loss_terms_kernel", "loss_terms_final_kernel"], "loss_bwd": ["loss_terms_bwd_kernel"],
    "train_fwd": ["layer_loss_kernel", "loss_terms_final_kernel"],
    "train_fwd_two_calls": ["OpLayer", "loss_terms_kernel", "loss_terms_final_kernel"],
    "trispace_bwd": ["trispace_bwd_px", "trispace_coef_grad", "trispace_coef_final", "trispace_bwd_fused"],
}
FRAG_KEYS = {"layer": "OpLayer", "lab_stage": "OpLabStage", "hsv_stage": "OpHsvStage", "rgb_only": "OpAdjust3",
             "trispace": "OpTriSpace", "layer_bwd": "layer_bwd_kernel"}


def avg(v):
    v = v[10:] if len(v) > 20 else v  # drop warm-up launches
    return sum(v) / len(v) if v else None


def kernel_entry(root, frag, n_calls):
    tr = load_trace(os.path.join(root, "trace"))
    out = None
    for name, durs in tr.items():
        if frag not in name:
            continue
        e = {"kernel": name, "launches": len(durs), "launches_per_call": round(len(durs) / n_calls) if n_calls else None,
             "kernel_avg_us_trace_pass": avg(durs)}
        for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
            c = load_counters(os.path.join(root, sub)).get(name, {})
            for cname, vals in c.items():
                e[cname] = avg(vals)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["read_bytes_corrected_x2"] = 2.0 * e["FETCH_SIZE"] * 1024
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
            e["hbm_bytes_per_launch"] = e["read_bytes_corrected_x2"] + e["write_bytes"]
        if all(k in e for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "GRBM_GUI_ACTIVE")) and e["GRBM_GUI_ACTIVE"]:
            e["valu_issue_util"] = (e["SQ_ACTIVE_INST_VALU"] - e["SQ_ACTIVE_INST_VALU2"]) * 4.0 / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_INSTS_VALU" in e and e.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
        if out is None or (e["kernel_avg_us_trace_pass"] or 0) * e["launches"] > (out["kernel_avg_us_trace_pass"] or 0) * out["launches"]:
            out = e  # several instantiations match (float4 / scalar): keep the one the time goes to
    return out


def main(tag):
    res = {}
    for workload, frags in KERNELS.items():
        root = os.path.join("gpurun_out", f"prof_{tag}_{workload}")
        if not os.path.isdir(os.path.join(root, "trace")):
            continue
        # calls = launches of the workload's first kernel that appears in the trace (one per call)
        tr = load_trace(os.path.join(root, "trace"))
        n_calls = next((len(d) for f in frags for n, d in tr.items() if f in n), 0)
        ks = [k for k in (kernel_entry(root, f, n_calls) for f in frags) if k]
        if not ks or not all("hbm_bytes_per_launch" in k for k in ks):
            continue
        dom = max(ks, key=lambda k: (k["kernel_avg_us_trace_pass"] or 0) * (k["launches_per_call"] or 1))
        res[workload] = {
            "hbm_bytes_per_launch": sum(k["hbm_bytes_per_launch"] * (k["launches_per_call"] or 1) for k in ks),
            "us_per_call_trace_pass": sum((k["kernel_avg_us_trace_pass"] or 0) * (k["launches_per_call"] or 1) for k in ks),
            "valu_issue_util": dom.get("valu_issue_util"), "dominant_kernel": dom["kernel"],
            "kernels": ks,
        }
        if workload in FRAG_KEYS and len(ks) >= 1:
            res[FRAG_KEYS[workload]] = dict(ks[0])
    os.makedirs("profiles", exist_ok=True)
    out_path = os.path.join("profiles", f"traffic_{tag}.json")
    if os.path.exists(out_path):  # passes taken in several sessions (a GPU call hands back 64 MiB at most): new entries replace old ones
        try:
            res = {**json.load(open(out_path)), **res}
        except Exception:
            pass
    json.dump(res, open(out_path, "w"), indent=1)
    for w, e in res.items():
        if "kernels" in e:
            print(f"{w:16s} {e['hbm_bytes_per_launch'] / 1e9:8.4f} GB per call, {e['us_per_call_trace_pass']:8.1f} us, "
                  f"valu issue util {e['valu_issue_util'] if e['valu_issue_util'] is None else round(e['valu_issue_util'], 3)}")
    return res


if __name__ == "__main__":
    main(sys.argv[1])
