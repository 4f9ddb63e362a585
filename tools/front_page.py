"""DESIGN.md's front page (section 0): one row per kernel -- current form, launch shape, time, bound, the log that justifies it.
The numbers come from a bench record (bench_detail.json as bench.py writes it) and the committed counter passes
(profiles/traffic_r*.json); the words are this file's.  Rewrites the block between the FRONT markers of DESIGN.md.

    python tools/front_page.py profiles/r05/bench_detail_driver_shape.json
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# workload -> (kernel / C-ABI entry, current form, launch shape, the log)
WORDS = {
    "layer": ("`stream_kernel<OpLayer,4,1,U8>` · `curl_layer_fwd_f32`",
              "CURLLayer.forward (model.py:137-176) in one pass: four pixels per lane in phases (converters' transcendental runs lined "
              "up at raised issue priority, lazy threshold selects), collapsed curves `a + b·x` from SGPRs, non-temporal float4 planes",
              "46 880 workgroups × 256 threads, one 1 024-px tile each, 8 waves/SIMD (44 VGPRs); `knots_prep_kernel` in front "
              "(one launch with the collapse inside for ≤ 2 048 workgroups)",
              "§3c, §3d; `profiles/r04/layer_pmc_summary.json`, `ab_*` logs of r03 / r04"),
    "layer_disk": ("same kernel, bool disk mask (70 %)", "fully masked wavefronts take the constant shortcut", "as above", "§3d.14"),
    "layer_disk_mask_first": ("`stream_kernel<OpLayer,…,mask_first>` · `CURL_F_MASK_FIRST`",
                              "asks for the mask bytes first; a fully masked-out wavefront never reads its pixels", "as above",
                              "§3d.14, `profiles/r05/layer_disk_mask_first_pmc_summary.json`"),
    "layer_8bit": ("same kernel on coherent 8-bit content", "what data.py / infer.py feed it (ties, zeros, flat darks)", "as above",
                   "`tools/synth8.py`, `profiles/r04/layer_8bit_pmc_summary.json`"),
    "lab_stage": ("`stream_kernel<OpLabStage>` · `curl_lab_stage_f32`", "RGB→Lab→3 curves→×mask→RGB (the kernel north_star's 70 % names)",
                  "46 880 workgroups, 6 resident per CU (LDS-granule cap)", "§3d.13, `profiles/r04/lab_stage_pmc_summary.json`"),
    "hsv_stage": ("`stream_kernel<OpHsvStage>` · `curl_hsv_stage_f32`", "RGB→HSV→4 curves→×mask→RGB", "4 resident workgroups per CU",
                  "§3d.13, `profiles/r04/hsv_stage_pmc_summary.json`"),
    "rgb_only": ("`stream_kernel<OpAdjust3>` · `curl_adjust_rgb_f32`", "3 collapsed curves, no mask (BASELINE configs[1])",
                 "2 resident workgroups per CU: the memory system's own best shape", "§3d.13, `profiles/r04/rgb_only_pmc_summary.json`"),
    "trispace": ("`stream_kernel<OpTriSpaceRows>` · `curl_trispace_fwd_f32`",
                 "3 × degree-4 polynomial layers (126 coeffs × 3) in RGB / Lab / HSV + converters, row-folded packed Horner, "
                 "coefficients in LDS", "one image row segment per workgroup", "§3a, `profiles/r04/trispace_pmc_summary.json`"),
    "layer_u8": ("`stream_kernel<OpLayer,…,FMT_U8HWC>` · `curl_layer_fwd_u8hwc`", "the layer between interleaved bytes (byte/255 and truncating ×255 "
                 "in registers)", "as the layer; 3 dwords = 4 px per lane", "`profiles/r05/layer_u8_pmc_summary.json`"),
    "trispace_u8": ("`stream_kernel<OpTriSpaceRows,…,FMT_U8HWC>` · `curl_trispace_fwd_u8hwc`", "infer.py:35-47 in one launch", "as trispace",
                    "`profiles/r05/trispace_u8_pmc_summary.json`"),
    "layer_bwd": ("`layer_bwd_kernel<4,U8,true>` + `knots_bwd_kernel` · `curl_layer_bwd_f32`",
                  "taped reverse mode, one pixel after the other (gates as lane predicates in SGPR pairs: 104 of 104 SGPRs), 20 curve sums "
                  "wave → LDS → block partials, float64 fixed-order second pass; 8 × 1500×1000",
                  "11 719 workgroups, 4 waves/SIMD (120 VGPRs)", "§Backward, §3f.2; `profiles/r04/layer_bwd_pmc_summary.json`"),
    "layer_bwd_crop": ("same, 32 × 256×256 (the training crop batch)", "two dependent launches for 2.1 Mpx", "2 048 workgroups = two residency rounds",
                       "`profiles/r04/layer_bwd_crop_pmc_summary.json`"),
    "layer_bwd_knots": ("`layer_bwd_kernel<4,U8,false>` · `grad_img = NULL`", "stops at the Lab curves' sums (no RGB2LAB pullback, nothing stored): what "
                        "training runs", "109 VGPRs", "§3e.8; `profiles/r04/layer_bwd_knots_pmc_summary.json`"),
    "layer_bwd_crop_knots": ("same, 32 × 256×256", "", "", "`profiles/r04/layer_bwd_crop_knots_pmc_summary.json`"),
    "loss_fwd": ("`loss_terms_kernel` + `loss_terms_final_kernel` · `curl_loss_terms_f32`", "RGB L1, cosine, Lab L1, HSV-cone L1 + the two L planes; phases "
                 "over the 8 colours of a lane", "46 880 workgroups", "§3b; `profiles/r04/loss_fwd_pmc_summary.json`"),
    "loss_bwd": ("`loss_terms_bwd_kernel` · `curl_loss_terms_bwd_f32`", "taped converters, `sign` as two scalings + median; round 5: without the identity "
                 "clamps / gates on RGB2HSV's output", "46 880 workgroups", "§3b, §3f.3; `profiles/r05/ab_loss_hsv_identity_clamps.log`"),
    "train_fwd": ("`layer_loss_kernel` + `loss_terms_final_kernel` · `curl_layer_loss_fwd_f32`", "round 5: the train step's forward in one pass "
                  "(main.py:283-285) -- the layer, then CURLLoss' pointwise terms on the prediction in registers; `out`, `reg`, sums and L planes "
                  "are the two-call route's bits", "46 880 workgroups, 4 waves/SIMD (115 VGPRs); ≤ 2 048 workgroups collapse their curves inside",
                  "§3f.7; `profiles/r05/ab_train_fwd.log`"),
    "train_fwd_two_calls": ("`curl_layer_fwd_f32` then `curl_loss_terms_f32`", "the same work as two calls (25 + 33 B/px)", "as `layer` + `loss_fwd`", "§3f.7"),
    "trispace_bwd": ("`trispace_bwd_px` + `trispace_coef_grad` + `trispace_coef_final` · `curl_trispace_bwd_f32`",
                     "per-pixel pullback to 18 planes, then strip-tiled coefficient sums (72 B/px of intermediates; the one-pass form lost)",
                     "three launches; 8 × 1500×1000", "§3e.4; `profiles/r04/poly_bwd_fused_vs_three_kernel_ab.log`"),
}
ORDER = ["layer", "layer_8bit", "layer_disk", "layer_disk_mask_first", "lab_stage", "hsv_stage", "rgb_only", "layer_u8", "trispace",
         "trispace_u8", "layer_bwd", "layer_bwd_knots", "layer_bwd_crop", "layer_bwd_crop_knots", "loss_fwd", "loss_bwd", "train_fwd", "train_fwd_two_calls", "trispace_bwd"]


def table(detail):
    rows = {"layer": detail["headline"], **detail["other_workloads"]}
    out = ["| row (`bench.py --workload`) | kernel · entry | current form | launch shape | µs per call | bound: fraction of its peak · other roofline · "
           "issue slots used · HBM traffic ÷ algorithmic | where it is argued |", "|---|---|---|---|---|---|---|"]
    for n in ORDER:
        if n not in rows:
            continue
        r, w = rows[n], WORDS[n]
        ro = r["roofline"]
        sec = ro.get("secondary", {})
        alg = ro["algorithmic_bytes_per_px"] * ro["px_per_launch"]
        peak = "HBM 8 TB/s" if ro["bound"] == "hbm" else "157.3 TFLOP/s f32 vector"
        other = f"{sec.get('frac', 0):.2f} of {'the vector peak' if ro['bound'] == 'hbm' else 'HBM'}"
        util = "—" if ro.get("valu_issue_util") is None else f"{ro['valu_issue_util']:.2f}"
        tr = "—" if ro.get("traffic") is None else f"{ro['traffic'] / alg:.2f}×"
        out.append(f"| `{n}` | {w[0]} | {w[1]} | {w[2]} | **{r['device_ms_per_step'] * 1e3:.1f}** | **{ro['frac']:.3f}** of {peak} · {other} · "
                   f"{util} · {tr} | {w[3]} |")
    return "\n".join(out)


def main(path):
    detail = json.load(open(path))
    line = detail["line"]
    src = os.path.relpath(os.path.abspath(path), ROOT)
    head = (f"Numbers: `{src}` (`python bench.py --steps {line['steps']} --warmup {line['warmup']}`, one MI355X, HIP events on the "
            f"launch stream, per call incl. every launch of the call) and the committed counter passes `profiles/traffic_r0*.json`; "
            f"batch 32 × 1500×1000 unless the row says otherwise.  Headline: **{line['value']:.0f} Mpix/s**, "
            f"{line['ms_per_step']:.4f} ms per step (wall), kernel-side {line['roofline']['frac']:.3f} of the HBM peak; the contract read "
            f"literally (no clock-settle launches): {line['literal_protocol_ms_per_step']:.4f} ms = {line['literal_protocol_frac']:.3f}.  "
            f"CPU beside it: {line['cpu_baseline']['value']:.2f} Mpix/s on {line['cpu_baseline']['cores']} threads "
            f"({line['cpu_baseline'].get('cpu_model', '')}).  Regenerate: `python tools/front_page.py {src}`.")
    block = "<!-- FRONT:BEGIN (tools/front_page.py) -->\n" + head + "\n\n" + table(detail) + "\n<!-- FRONT:END -->"
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    if "<!-- FRONT:BEGIN" in s:
        s = re.sub(r"<!-- FRONT:BEGIN.*?<!-- FRONT:END -->", lambda m: block, s, flags=re.S)
    else:
        raise SystemExit("DESIGN.md has no FRONT markers")
    open(p, "w").write(s)
    print(block)


if __name__ == "__main__":
    main(sys.argv[1])
