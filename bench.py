#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on the MI355X: Mpix/s through the fused curve-apply at
1500x1000, batch 32 per GPU, plus PSNR delta vs the reference arithmetic (the oracle).

    python bench.py --gpus N --steps K --warmup W

N=1 runs in this process.  N>1 under torch.distributed.run (RANK/WORLD_SIZE set: how the driver starts it) is one
rank per GPU; N>1 started plainly launches `python -m torch.distributed.run --nproc-per-node N bench.py ...`
ITSELF as a child process -- before anything touches the GPU -- relays its output and exits with its code
(the launcher shape of main.py:98-124: one process per GPU, init_process_group("nccl")).

A step = one pass of the hot path over one batch: CURLLayer.forward (model.py:137-176) as the fused
HIP kernel (RGB->Lab->curves->RGB->curves->HSV->curves->RGB + residual), inputs resident in HBM.
Batches shard by image across ranks with NO data-path collective (weak scaling, 32 images per GPU).
Rank 0 prints ONE compact JSON line on stdout (make_line: the contract's keys, `roofline`, `cpu_baseline`, four accuracy
scalars and the literal-protocol figure; <= 4 KB, strict JSON -- tests/test_host_logic.py holds it to that).  Everything
else (the other workloads' rows, the end-to-end and train-step records, the full accuracy block, board power) goes to
bench_detail.json (beside this file and under gpurun_out/) and to stderr, never stdout.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_IMG, W_IMG = 1000, 1500
CLOCK_SETTLE_LAUNCHES = 150
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# the path's own 3-planes-in / 3-planes-out copy at its best launch shape (2 waves per SIMD: 174.0-174.9 us on two cards;
# 6.17 TB/s at full occupancy): tools/ubench/occ.hip, profiles/r03/exp27c_occupancy_copy_probe.log, exp27i_residency.log
COPY_CEILING_GBPS = 6585.0


def disk_mask(B, H, W, device):
    yy = torch.arange(H, device=device, dtype=torch.float32).view(H, 1)
    xx = torch.arange(W, device=device, dtype=torch.float32).view(1, W)
    d = ((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2
    return (d < 0.9).view(1, 1, H, W).expand(B, 1, H, W).contiguous()  # ~70 % coverage, bool


def make_inputs(B, device, seed, n_sets=2):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sets = []
    for _ in range(n_sets):  # rotate inputs so no step re-reads what the last one left in the 256 MiB MALL
        img = torch.rand(B, 3, H_IMG, W_IMG, generator=g).to(device)
        L = (torch.randn(B, 48, generator=g) * 0.1).to(device)
        R = (torch.randn(B, 48, generator=g) * 0.1).to(device)
        Hk = (torch.randn(B, 64, generator=g) * 0.1).to(device)
        poly = (torch.randn(B, 3, 3, 126, generator=g) * 0.2).to(device)
        u8 = (img * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()  # the file edge's interleaved bytes
        sets.append((img, L, R, Hk, poly, u8))
    return sets


VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz

WORKLOADS = {
    # name: description, algorithmic bytes per pixel [SURVEY.md 8(d)], kernel name fragment, mask, what bounds it
    # ("hbm": the streaming kernels; "valu": the kernels DESIGN.md 3/3a show to be arithmetic-bound), and for the
    # latter FLOP per pixel counted from the kernel's hot block (tools/flops_from_isa.py: v_pk_fma 4 per lane,
    # v_fma 2, v_pk_mul/add 2, v_mul/add/sub/min/max 1, transcendental 1; integer / bit / move instructions 0).
    "layer": dict(desc="CURLLayer.forward fused 3-stage (RGB->Lab->RGB->HSV->RGB + residual), bool mask all ones "
                       "(every pixel computed)", bpp=25.0, frag="OpLayer", mask="ones", bound="hbm", flop_px=198.0),
    "layer_disk": dict(desc="same kernel, bool disk mask ~70 % coverage (fully masked wavefronts take the constant "
                            "shortcut)", bpp=25.0, frag="OpLayer", mask="disk", bound="hbm", flop_px=198.0),
    "layer_disk_mask_first": dict(desc="same kernel and disk mask with CURL_F_MASK_FIRST (CURLLayer(foreground_masks=True)): "
                                       "fully masked-out wavefronts never read their pixels; GB/s against the nominal 25 B/px",
                                  bpp=25.0, frag="OpLayer", mask="disk", bound="hbm", flop_px=198.0),
    "layer_8bit": dict(desc="same kernel on spatially coherent 8-bit content (tools/synth8.py: k/255 values, gradients, grey ramps, "
                            "flat dark patches, tie palettes, saturated highlights -- what data.py:133-158 / infer.py:35-40 feed it), "
                            "float32 NCHW, bool mask all ones", bpp=25.0, frag="OpLayer", mask="ones", bound="hbm", flop_px=198.0),
    "lab_stage": dict(desc="fused RGB->Lab->3 curves->mask->RGB (the kernel BASELINE's 70 % target names), bool mask "
                           "all ones", bpp=25.0, frag="OpLabStage", mask="ones", bound="hbm", flop_px=124.0),
    "hsv_stage": dict(desc="fused RGB->HSV->4 curves->mask->RGB (model.py:163-169, the third per-colour-space kernel), "
                           "bool mask all ones", bpp=25.0, frag="OpHsvStage", mask="ones", bound="hbm", flop_px=75.0),
    "rgb_only": dict(desc="RGB-only 3 curves (adjust_rgb, BASELINE configs[1]), no mask", bpp=24.0, frag="OpAdjust3",
                     mask=None, bound="hbm", flop_px=9.0),
    "trispace": dict(desc="TriSpaceRegNet per-pixel path (SURVEY 8f-1): 3 x degree-4 polynomial layers (126 coeffs x 3 "
                          "outputs) in RGB/Lab/HSV + converters + clamp, fused", bpp=24.0, frag="OpTriSpace", mask=None,
                     bound="valu", flop_px=1464.0),
    "layer_u8": dict(desc="CURLLayer.forward on interleaved uint8 HWC in and out (SURVEY 8f-2: byte/255 and truncating "
                          "*255 fused), bool mask all ones", bpp=7.0, frag="OpLayer", mask="ones", bound="valu",
                     flop_px=210.0),
    "trispace_u8": dict(desc="TriSpaceRegNet per-pixel path on interleaved uint8 HWC in and out (infer.py:35-47 fused)",
                        bpp=6.0, frag="OpTriSpace", mask=None, bound="valu", flop_px=1473.0),
    # ---- BASELINE configs[4]'s kernels (the train step's custom HIP curve forward / backward and loss terms): VALU-bound,
    # rooflined against the 157.3 TFLOP/s vector peak; flop_px from the kernels' ISA (tools/flops_from_isa.py --all)
    "layer_bwd": dict(desc="curve-layer BACKWARD (curl_layer_bwd_f32: d img + d raw knots; autograd of model.py:137-176, the "
                           "forward's knot workspace handed back as the autograd node does) on 8 x 1500x1000 frames, bool mask all ones", bpp=37.0, frag="layer_bwd_kernel", mask="ones", bound="valu",
                      flop_px=413.5, images=8),
    "layer_bwd_crop": dict(desc="curve-layer backward on the training crop batch, 32 x 256x256 (main.py:88, data.py:86), bool "
                                "mask all ones", bpp=37.0, frag="layer_bwd_kernel", mask="ones", bound="valu", flop_px=413.5,
                           images=32, hw=(256, 256)),
    "layer_bwd_knots": dict(desc="curve-layer backward, knot gradients only (grad_img = NULL: what the training step runs, the image "
                                 "being data -- main.py:287): no RGB2LAB pullback, no gradient image written; 8 x 1500x1000 frames",
                            bpp=25.0, frag="layer_bwd_kernel", mask="ones", bound="valu", flop_px=365.8, images=8, knots_only=True),
    "layer_bwd_crop_knots": dict(desc="the same on the training crop batch, 32 x 256x256", bpp=25.0, frag="layer_bwd_kernel",
                                 mask="ones", bound="valu", flop_px=365.8, images=32, hw=(256, 256), knots_only=True),
    "loss_fwd": dict(desc="CURLLoss pointwise terms forward (model.py:89-109: RGB L1, cosine, Lab L1, HSV-cone L1 of prediction "
                          "and target + the two L planes), bool mask all ones", bpp=33.0, frag="loss_terms_kernel",
                     mask="ones", bound="valu", flop_px=241.2),
    "loss_bwd": dict(desc="CURLLoss pointwise terms backward (gradient w.r.t. the prediction)", bpp=41.0,
                     frag="loss_terms_bwd_kernel", mask="ones", bound="valu", flop_px=344.8),
    "train_fwd": dict(desc="the train step's forward in ONE pass (curl_layer_loss_fwd_f32, main.py:283-285: CURLLayer.forward + "
                           "CURLLoss' pointwise terms on the prediction in registers; out, reg, 5 sums, both L planes), bool mask all "
                           "ones", bpp=45.0, frag="layer_loss_kernel", mask="ones", bound="valu", flop_px=439.2),
    "train_fwd_two_calls": dict(desc="the same as two calls (curl_layer_fwd_f32 then curl_loss_terms_f32: 25 + 33 B/px)", bpp=58.0,
                                frag="loss_terms_kernel", mask="ones", bound="valu", flop_px=439.2),
    "trispace_bwd": dict(desc="polynomial path backward (curl_trispace_bwd_f32: d loss / d 3x3x126 coefficients, main.py:287) "
                              "on 8 x 1500x1000 frames: three kernels, 72 B/px of intermediates between the first two",
                         bpp=24.0, frag="trispace_bwd", mask=None, bound="valu", flop_px=3059.0, images=8),
}
CONFIG5 = ("layer_bwd", "layer_bwd_crop", "layer_bwd_knots", "layer_bwd_crop_knots", "loss_fwd", "loss_bwd", "trispace_bwd", "train_fwd",
           "train_fwd_two_calls")


def workload_pixels(name, B):
    w = WORKLOADS[name]
    h, wd = w.get("hw", (H_IMG, W_IMG))
    return min(w.get("images", B), B) * h * wd


def make_step(name, ops, masks, sets=None):
    mask = masks.get(WORKLOADS[name]["mask"])
    if name in CONFIG5:
        w = WORKLOADS[name]
        n = min(w.get("images", sets[0][0].shape[0]), sets[0][0].shape[0])
        need_img = not w.get("knots_only", False)
        if "hw" in w:  # the training crop batch: its own small tensors
            h, wd = w["hw"]
            dev = sets[0][0].device
            crops = [(s[0][:n, :, :h, :wd].contiguous(), s[1][:n], s[2][:n], s[3][:n]) for s in sets]
            gout = torch.rand(n, 3, h, wd, device=dev)
            m = torch.ones(n, 1, h, wd, dtype=torch.bool, device=dev)
            turn = [0]

            # as the autograd node runs it: the knot workspace the forward filled is handed back (CURL_F_WS_READY)
            wss = [ops.curl_layer_forward(c[0], m, c[1], c[2], c[3], return_workspace=True)[2] for c in crops]

            def crop_step(_s):
                k = turn[0] % len(crops)
                c = crops[k]
                turn[0] += 1
                return ops.curl_layer_backward(c[0], m, c[1], c[2], c[3], gout, workspace=wss[k], need_grad_img=need_img)
            return crop_step
        # any resident float image serves as the incoming gradient: the OTHER set's images n..2n, so that the gradient rotates
        # with the inputs and no step finds its gradient stream where the last one left it (the MALL holds 256 MiB)
        ids = {id(s): k for k, s in enumerate(sets)}
        B_have = sets[0][0].shape[0]
        lo = n if 2 * n <= B_have else 0
        gouts = [sets[(k + 1) % len(sets)][0][lo:lo + n] for k in range(len(sets))]
        m = None if mask is None else mask[:n]
        if name in ("layer_bwd", "layer_bwd_knots"):
            ws8 = [ops.curl_layer_forward(s[0][:n], m, s[1][:n], s[2][:n], s[3][:n], return_workspace=True)[2] for s in sets]
            return lambda s: ops.curl_layer_backward(s[0][:n], m, s[1][:n], s[2][:n], s[3][:n], gouts[ids[id(s)]],
                                                           workspace=ws8[ids[id(s)]], need_grad_img=need_img)
        if name == "trispace_bwd":
            return lambda s: ops.trispace_backward(s[0][:n], s[4][:n], gouts[ids[id(s)]])
        other = sets[1 % len(sets)][0]
        if name == "train_fwd":
            return lambda s: ops.layer_loss_forward(s[0], mask, s[1], s[2], s[3], other)

        if name == "train_fwd_two_calls":
            def two(s):
                out, reg = ops.curl_layer_forward(s[0], mask, s[1], s[2], s[3])
                return ops.loss_term_sums(out, other, mask)
            return two
        if name == "loss_fwd":
            return lambda s: ops.loss_term_sums(s[0], other, mask)
        w4 = torch.ones(4, device=other.device)
        gL = sets[0][0][:, :1].contiguous()
        return lambda s: ops.loss_terms_backward(s[0], other, mask, w4, gL)
    if name in ("layer", "layer_disk"):
        return lambda s: ops.curl_layer_forward(s[0], mask, s[1], s[2], s[3])
    if name == "layer_8bit":
        # the SAME entry point and kernel; only the pixel values differ (inputs rotate like the others')
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import synth8
        B, dev = sets[0][0].shape[0], sets[0][0].device
        imgs8 = []
        for k in range(len(sets)):
            distinct = synth8.coherent_8bit_frames(min(B, 8), H_IMG, W_IMG, seed=100 + k)  # 8 distinct frames, repeated
            u8 = torch.from_numpy(distinct).to(dev).repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
            imgs8.append(ops.u8hwc_to_f32chw(u8))  # to_tensor's byte / 255 (exact)
        ids = {id(s): k for k, s in enumerate(sets)}
        return lambda s: ops.curl_layer_forward(imgs8[ids[id(s)]], mask, s[1], s[2], s[3])
    if name == "layer_disk_mask_first":
        return lambda s: ops.curl_layer_forward(s[0], mask, s[1], s[2], s[3], flags=ops.F_MASK_FIRST)
    if name == "lab_stage":
        return lambda s: ops.lab_stage(s[0], mask, s[1])
    if name == "hsv_stage":
        return lambda s: ops.hsv_stage(s[0], mask, s[3])
    if name == "rgb_only":
        return lambda s: ops.adjust_rgb(s[0], s[2])
    if name == "trispace":
        return lambda s: ops.trispace_forward(s[0], s[4])
    if name == "layer_u8":
        return lambda s: ops.curl_layer_forward_u8hwc(s[5], mask, s[1], s[2], s[3])
    if name == "trispace_u8":
        return lambda s: ops.trispace_forward_u8hwc(s[5], s[4])
    raise ValueError(name)


def timed_run(step, sets, steps, warmup, dist, device):
    """W untimed warm-up steps, then EXACTLY K steps between barrier+synchronize brackets.
    Returns (wall seconds max over ranks, mean device ms per step from events on the launch stream)."""
    # Clock settle: after an idle gap the MI355X takes ~30 ms of continuous work to reach its steady clock
    # (profiles/r01/sustained_windows.log: the first ~100 launches of the arithmetic-heavy kernels run 15-25 %
    # slow).  These launches are untimed and come ON TOP of the W warm-up steps the contract asks for.
    for i in range(CLOCK_SETTLE_LAUNCHES):
        step(sets[i % len(sets)])
    for i in range(warmup):
        step(sets[i % len(sets)])
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # torch's current stream == the stream the kernels are enqueued on (ops._stream)
    for i in range(steps):
        step(sets[i % len(sets)])
    ev1.record()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / steps
    dev_min = dev_ms
    if dist is not None:
        on = device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([wall, dev_ms, -dev_ms], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms, dev_min = float(t[0]), float(t[1]), -float(t[2])
    return wall, dev_ms, dev_min


class BoardPower:
    """Board power / shader clock from the card's hwmon files, sampled in a thread while a timed region runs (context for
    the roofline: the fused kernels run at the board's power cap).  Silent if the files are not there."""

    def __init__(self, device_index=0):
        import glob
        self.files = {}
        want = None
        try:  # the host's sysfs shows every card of the node: pick the one this process computes on, by PCI address
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        except Exception:
            pass
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            dev = os.path.realpath(os.path.join(d, "..", ".."))
            if want is not None and want not in dev:
                continue
            for name in ("power1_average", "power1_input", "power1_cap", "freq1_input"):
                p = os.path.join(d, name)
                if os.path.exists(p):
                    self.files.setdefault(name, p)
        self.samples, self._stop, self._th = [], False, None

    @staticmethod
    def _read(p):
        try:
            return int(open(p).read().strip())
        except Exception:
            return None

    def __enter__(self):
        import threading
        if self.files:
            def loop():
                while not self._stop:
                    self.samples.append({k: self._read(p) for k, p in self.files.items()})
                    time.sleep(0.02)
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th is not None:
            self._th.join()

    def summary(self):
        pw = [s.get("power1_average") or s.get("power1_input") for s in self.samples]
        pw = [p for p in pw if p]
        fq = [s["freq1_input"] for s in self.samples if s.get("freq1_input")]
        cap = next((s["power1_cap"] for s in self.samples if s.get("power1_cap")), None)
        if not pw:
            return None
        return {"board_power_W_mean": sum(pw) / len(pw) / 1e6, "board_power_cap_W": cap / 1e6 if cap else None,
                "shader_clock_MHz_mean": sum(fq) / len(fq) / 1e6 if fq else None, "samples": len(pw),
                "source": "amdgpu hwmon power1_*, freq1_input sampled every 20 ms while the step runs back to back for 3 s after the timed region (second half of the samples)"}


def cold_start_run(step, sets, steps, warmup, device):
    """The contract read literally, from a cold chip: half a second idle, then EXACTLY `warmup` untimed steps and `steps`
    timed ones -- without the CLOCK_SETTLE_LAUNCHES that precede the headline's warm-up.  Reported beside the headline
    (rank-local, N = 1 semantics) so that the cost of the clock ramp is a number in the record, not a footnote."""
    torch.cuda.synchronize(device)
    time.sleep(0.5)
    for i in range(warmup):
        step(sets[i % len(sets)])
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        step(sets[i % len(sets)])
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / steps * 1e3


def cold_first_launch_us(step, sets, device):
    """What a caller issuing ONE batch after an idle gap sees: the chip idles for half a second, then one step is
    timed with events on the launch stream (the steady-state figures come after CLOCK_SETTLE_LAUNCHES launches)."""
    step(sets[0])  # code object loaded, workspace allocated
    torch.cuda.synchronize(device)
    time.sleep(0.5)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    step(sets[1 % len(sets)])
    ev1.record()
    torch.cuda.synchronize(device)
    return ev0.elapsed_time(ev1) * 1e3


def accuracy_vs_oracle(ops, device):
    """PSNR delta vs ref on one full-size frame: HIP output vs the CPU oracle (= the reference's arithmetic), with the
    yardstick beside it: how far the reference's OWN float32 result is from its float64 evaluation on the same frame
    (the chain is ill-conditioned at dark / near-grey pixels), and how far the HIP result is from that
    float64 truth.  A pixel where |HIP - ref32| > 1e-5 is expected to be one where ref32 itself is off."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import curl_oracle as O
    g = torch.Generator().manual_seed(123)
    img = torch.rand(1, 3, H_IMG, W_IMG, generator=g)
    L = torch.randn(1, 48, generator=g) * 0.1
    R = torch.randn(1, 48, generator=g) * 0.1
    Hk = torch.randn(1, 64, generator=g) * 0.1
    gt = torch.rand(1, 3, H_IMG, W_IMG, generator=g)  # arbitrary synthetic target
    mask = disk_mask(1, H_IMG, W_IMG, torch.device("cpu"))
    mf = mask.float()
    ref, ref_reg = O.curl_layer(img, mf, L, R, Hk)
    ref64, _ = O.curl_layer(img.double(), mf.double(), L.double(), R.double(), Hk.double())
    out, reg = ops.curl_layer_forward(img.to(device), mask.to(device), L.to(device), R.to(device), Hk.to(device))
    out = out.cpu()
    d = (out.double() - ref.double()).abs()
    noise = (ref.double() - ref64).abs()      # the reference's own float32 rounding on this frame
    ours = (out.double() - ref64).abs()       # the HIP result against float64 truth
    over = d > 1e-5
    # conditioning of the chain per pixel: max |d out / d in| of the float64 evaluation (finite differences)
    S = torch.zeros(1, H_IMG, W_IMG, dtype=torch.float64)
    for k in range(3):
        for sgn in (1e-6, -1e-6):
            p = img.double().clone()
            p[:, k] += sgn
            o, _ = O.curl_layer(p, mf.double(), L.double(), R.double(), Hk.double())
            S = torch.maximum(S, (o - ref64).abs().amax(1) / 1e-6)
    S3 = S[:, None].expand_as(d)
    mse = float((d ** 2).sum() / (3 * mf.sum()))
    psnr_vs_ref = None if mse == 0 else 10 * torch.log10(torch.tensor(1.0 / mse)).item()  # None: identical (strict JSON)
    p_out, p_ref = O.psnr(out, gt, mf), O.psnr(ref, gt, mf)
    return {
        "sample": "1 x 1500x1000 frame, knots N(0,0.1), bool disk mask, vs oracle (fp32 reference arithmetic)",
        "max_abs_err": float(d.max()),
        "frac_px_over_1e-5": float(over.double().mean()),
        "frac_allclose_rtol1e-5_atol1e-6": float((d <= 1e-6 + 1e-5 * ref.double().abs()).double().mean()),
        # the reference is discontinuous on the hue seam (g == b with r maximal: hue 0 <-> 1 in front of a non-periodic
        # hue curve): there its float32 and float64 evaluations land on different sides -- counted, then set aside
        "px_where_ref32_and_ref64_take_different_hue_branches": int(((noise > 1e-3).any(1)).sum()),
        "ref_self_noise_max": float(noise[noise <= 1e-3].max()),
        "ref_self_noise_frac_over_1e-5": float((noise > 1e-5).double().mean()),
        "ours_vs_f64_max": float(ours[noise <= 1e-3].max()),
        "ours_vs_f64_frac_over_1e-5": float((ours > 1e-5).double().mean()),
        "frac_of_over_1e-5_px_where_ref_noise_over_2.5e-6": (float((noise[over] > 2.5e-6).double().mean())
                                                              if bool(over.any()) else None),
        # tests/test_gpu_parity.py::test_fullsize_exception_set_is_pinned_by_conditioning: error <= max(1e-5, 2e-6 * S)
        "min_sensitivity_where_err_over_1e-5": float(S3[over].min()) if bool(over.any()) else None,
        "max_err_per_unit_sensitivity": float((d / S3.clamp_min(5.0)).max()),
        "ref_self_noise_per_unit_sensitivity": float((noise / S3.clamp_min(5.0)).max()),
        "frac_px_sensitivity_over_5": float((S > 5.0).double().mean()),
        "psnr_out_vs_ref_db": psnr_vs_ref,
        "psnr_delta_db": abs(float(p_out) - float(p_ref)),
        "reg_rel_err": float(((reg.cpu() - ref_reg).abs() / ref_reg.abs()).max()),
    }


def end_to_end(ops, device, sets, masks, steps=5):
    """BASELINE configs[1]/[2] as a user runs them: GCURLNet = encoder (timm's efficientnetv2_rw_s architecture on stock
    PyTorch-ROCm, random init, fp32, fed the 320x320 view as infer.py:32-36 does) -> 160 knots -> the fused layer at
    full resolution.  Context for the headline: how the per-pixel path compares with the encoder in one inference batch."""
    from curl_amd import model
    img = sets[0][0]
    net = model.GCURLNet(encoder_size=320).to(device).eval()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    enc_ms, lay_ms = [], []
    with torch.no_grad():
        for i in range(steps + 2):
            ev[0].record()
            knots = net.predict_knots(img)
            ev[1].record()
            L, R, Hk = knots[:, :48], knots[:, 48:96], knots[:, 96:]
            net.curllayer(img, masks["ones"], L, R, Hk)
            ev[2].record()
            torch.cuda.synchronize(device)
            if i >= 2:
                enc_ms.append(ev[0].elapsed_time(ev[1]))
                lay_ms.append(ev[1].elapsed_time(ev[2]))
    e, l = sorted(enc_ms)[len(enc_ms) // 2], sorted(lay_ms)[len(lay_ms) // 2]
    B = img.shape[0]
    out = {"model": "GCURLNet: efficientnetv2_rw_s encoder (random init, fp32, stock PyTorch-ROCm) on the 320x320 view + "
                    "fused CURLLayer at 1500x1000", "batch": B, "encoder_ms": e, "curve_layer_ms": l,
           "ms_per_batch": e + l, "images_per_s": B / ((e + l) * 1e-3), "curve_layer_share": l / (e + l)}
    # the fork's live model as infer.py:22-47 runs it: TriSpaceRegNet (efficientnetv2_rw_t backbone + the 1024-1024-512-512
    # head) on the 320x320 view -> 3 x 3 x 126 coefficients -> the fused polynomial path on the full-resolution bytes
    # (uint8 HWC in and out: to_tensor, white background and the truncating *255 inside the kernel)
    from curl_amd import infer as infer_mod
    tri = model.TriSpaceRegNet(spatial=True, is_train=False).to(device).eval()
    u8 = sets[0][5]
    ones = masks["ones"].float()
    enc_ms, px_ms = [], []
    with torch.no_grad():
        for i in range(steps + 2):
            ev[0].record()
            small, msmall = infer_mod.encoder_view(img, ones)
            coeffs = torch.stack(tri.generate_coefficients(small, msmall), 1)
            ev[1].record()
            ops.trispace_forward_u8hwc(u8, coeffs)
            ev[2].record()
            torch.cuda.synchronize(device)
            if i >= 2:
                enc_ms.append(ev[0].elapsed_time(ev[1]))
                px_ms.append(ev[1].elapsed_time(ev[2]))
    e2, p2 = sorted(enc_ms)[len(enc_ms) // 2], sorted(px_ms)[len(px_ms) // 2]
    out["live_model"] = {"model": "TriSpaceRegNet (infer.py:22-47): efficientnetv2_rw_t backbone + head (random init, fp32, stock "
                                  "PyTorch-ROCm) on the 320x320 view incl. the resize, + the fused polynomial path at 1500x1000 on "
                                  "uint8 HWC in and out", "batch": B, "encoder_ms": e2, "per_pixel_path_ms": p2,
                         "ms_per_batch": e2 + p2, "images_per_s": B / ((e2 + p2) * 1e-3), "per_pixel_share": p2 / (e2 + p2)}
    return out


def train_step_record(device, rank, world, local, dist):
    """BASELINE configs[4] in synthetic form ("report"): tools/train_step.py's measurement -- a
    data-parallel train step of the curve model, 32 crops of 256x256 per GPU (main.py:88, data.py:86), encoder forward /
    backward on stock PyTorch-ROCm, the fused HIP curve layer forward / backward, Adam; DDP over RCCL when world > 1 --
    inside the bench line.  Every rank runs it (DDP needs them all); the dict is the same shape on each."""
    import importlib.util
    from types import SimpleNamespace
    spec = importlib.util.spec_from_file_location("curl_train_step", os.path.join(ROOT, "tools", "train_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = SimpleNamespace(steps=20, warmup=8, size=256, batch=32, width=1.0)
    # DDP wants the index of the device this rank computes on (on a rehearsal box ranks wrap onto the GPUs there are)
    return mod.run(args, device, rank, world, device.index or 0, dist)


def train_step_preflight(device):
    """One forward / backward of the train step's model on this rank alone (no collective): what can fail -- building the
    encoder, MIOpen's first solver search, memory -- fails here, on its own."""
    from curl_amd import model
    net = model.GCURLNet(backbone=model.CurveEncoder(160, width=1.0)).to(device).train()
    img = torch.rand(4, 3, 256, 256, device=device)
    out, reg = net(img, torch.ones(4, 1, 256, 256, dtype=torch.bool, device=device))
    (out.mean() + 1e-6 * reg.mean()).backward()
    torch.cuda.synchronize(device)


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """The oracle (torch-eager restatement of the reference's CPU path, bit-exact vs the reference in the build
    container) timed on this box's host cores on bounded samples of the same workload: the full chain
    with all the threads this job may use and with one, and RGB-only curves at the full batch of 32."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import curl_oracle as O
    g = torch.Generator().manual_seed(0)
    img = torch.rand(32, 3, H_IMG, W_IMG, generator=g)
    L = torch.randn(32, 48, generator=g) * 0.1
    R = torch.randn(32, 48, generator=g) * 0.1
    Hk = torch.randn(32, 64, generator=g) * 0.1
    ones = torch.ones(32, 1, H_IMG, W_IMG)
    # the 1-GPU box gives this job a 16-cpu share of a 256-cpu host: more threads than that only thrash
    # (the cpus this process may run on, not torch.get_num_threads(): a launcher such as torch.distributed.run exports
    # OMP_NUM_THREADS=1, which is a default for the ranks' GPU work, not a statement about the host)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, int(os.environ.get("CURL_CPU_THREADS", 16))))

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    legs = {}
    with torch.no_grad():
        torch.set_num_threads(threads)
        O.curl_layer(img[:1], ones[:1], L[:1], R[:1], Hk[:1])  # warm
        t = timed(lambda: O.curl_layer(img[:4], ones[:4], L[:4], R[:4], Hk[:4]), 3)
        full = 4 * H_IMG * W_IMG / t / 1e6
        legs["full_chain_b4"] = {"Mpix/s": full, "threads": threads}
        t = timed(lambda: O.adjust_rgb(img, R), 2)
        legs["rgb_only_b32"] = {"Mpix/s": 32 * H_IMG * W_IMG / t / 1e6, "threads": threads}
        torch.set_num_threads(1)
        t = timed(lambda: O.curl_layer(img[:1], ones[:1], L[:1], R[:1], Hk[:1]), 2)
        legs["full_chain_b1_1thread"] = {"Mpix/s": H_IMG * W_IMG / t / 1e6, "threads": 1}
        t = timed(lambda: O.adjust_rgb(img[:1], R[:1]), 2)
        legs["rgb_only_b1_1thread"] = {"Mpix/s": H_IMG * W_IMG / t / 1e6, "threads": 1}
        torch.set_num_threads(threads)
    return {"value": full, "unit": "Mpix/s", "cores": threads, "kind": "port",
            "sample": f"4 x 1500x1000 frames through the full chain (oracle/curl_oracle.py curl_layer, all-ones mask), "
                      f"median of 3, torch {torch.__version__} CPU, {threads} threads of {os.cpu_count()} host cpus",
            "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "legs": legs}


def load_traffic(kernel_fragment, workload=None):
    """HBM bytes per launch from the builder's committed PMC passes (profiles/traffic_r*.json: FETCH_SIZE doubled per the
    gfx950 note + WRITE_SIZE, separate --pmc runs; entries keyed by bench workload -- multi-kernel calls are summed there --
    or by kernel-name fragment) and where it came from -- replayed, not observed in this run.  With it, when the SQ pass of
    the same session is on file, the VALU issue-slot utilisation ((SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2)
    quad-cycles x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))."""
    prof = os.path.join(ROOT, "profiles")
    if not os.path.isdir(prof):
        return None, None, None
    for f in sorted(os.listdir(prof), reverse=True):
        if f.startswith("traffic_") and f.endswith(".json"):
            try:
                d = json.load(open(os.path.join(prof, f)))
            except Exception:
                continue
            for key in (workload, kernel_fragment):
                if key is not None and key in d:
                    return (d[key]["hbm_bytes_per_launch"], f"profiles/{f} (builder's rocprofv3 --pmc pass, not this run)",
                            d[key].get("valu_issue_util"))
    return None, None, None


# what a scaling run (N > 1) measures beside the headline unless --full is given
SCALING_RUN_WORKLOADS = ("lab_stage", "rgb_only")
LINE_BUDGET = 4096  # bytes: the driver keeps an 8 KB tail of stdout; r04's 20.9 KB line could not be read back


def _strict(x):
    """inf / nan -> None, floats to 6 significant digits: the line must survive a strict JSON parser."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.6g}")
    if isinstance(x, dict):
        return {str(k): _strict(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict(v) for v in x]
    return x


def dump_line(line):
    return json.dumps(_strict(line), allow_nan=False, separators=(", ", ": "))


def _hbm_view(res):
    r = res["roofline"]
    return r if r["bound"] == "hbm" else r["secondary"]


def make_line(main_res, others, accuracy, cpu, meta, train=None):
    """The ONE stdout line: the contract's keys, the headline kernel's `roofline` (+ the scalars of the kernels north_star's
    targets name: the fused Lab stage the 70 % figure is stated on, the HSV stage, RGB-only curves = BASELINE configs[1],
    the layer on coherent 8-bit content), `cpu_baseline`, four accuracy scalars and the figure of the contract read
    literally (no clock-settle launches).  `others` maps workload name -> measure() record.  Nothing here grows with the
    workload table: that lives in bench_detail.json."""
    world, B = meta["n_gpus"], meta["batch_per_gpu"]
    r = main_res["roofline"]
    roof = {"bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
            "traffic": r["traffic"], "traffic_source": r.get("traffic_source"),
            "achieved_wall": r["achieved_wall"], "frac_wall": r["frac_wall"], "valu_issue_util": r.get("valu_issue_util"),
            "algorithmic_bytes_per_px": r["algorithmic_bytes_per_px"], "px_per_launch": r["px_per_launch"],
            "kernel_us": main_res["device_ms_per_step"] * 1e3,
            "timing": "HIP events on the launch stream over the timed region"}
    for n in ("lab_stage", "hsv_stage", "rgb_only", "layer_8bit"):
        o = others.get(n)
        if o is not None:
            hb = _hbm_view(o)
            roof[f"{n}_us"] = o["device_ms_per_step"] * 1e3
            roof[f"{n}_GBps"] = hb["achieved"]
            roof[f"{n}_frac"] = hb["frac"]
    cold = main_res.get("cold_start")
    line = {
        "metric": "Mpix/s through fused curve-apply at 1500x1000 bs32; PSNR delta vs ref",
        "value": main_res["value"], "unit": "Mpix/s", "n_gpus": world, "steps": meta["steps"], "warmup": meta["warmup"],
        "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": main_res["workload"], "batch_per_gpu": B, "global_batch": B * world,
                   "height": H_IMG, "width": W_IMG, "knots": "randn*0.1 (160 per image)",
                   "parallelism": f"image-sharded x{world}, no data-path collective",
                   "clock_settle_launches": CLOCK_SETTLE_LAUNCHES,
                   "protocol": f"{CLOCK_SETTLE_LAUNCHES} untimed clock-settle launches precede the warm-up steps; "
                               "literal_protocol_* = the same steps and warm-up from an idle chip without them"},
        # the contract read literally beside the headline (cold_start_run: 0.5 s idle, W warm-up, K timed, wall clock)
        "literal_protocol_ms_per_step": cold["ms_per_step"] if cold else None,
        "literal_protocol_frac": (r["px_per_launch"] * r["algorithmic_bytes_per_px"] / (cold["ms_per_step"] * 1e-3) / 1e9
                                  / HBM_PEAK_GBPS) if cold and r["bound"] == "hbm" else None,
        "device_ms_per_step": main_res["device_ms_per_step"],
        "roofline": roof,
    }
    if cpu is not None:
        line["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model") if k in cpu}
    if accuracy is not None:
        line["accuracy"] = {k: accuracy.get(k) for k in ("max_abs_err", "frac_px_over_1e-5", "psnr_delta_db",
                                                         "max_err_per_unit_sensitivity")}
    if world > 1:
        line["ranks_seen"], line["backend"] = meta.get("ranks_seen"), meta.get("backend")
        line["device_ms_per_step_min_over_ranks"] = main_res.get("device_ms_per_step_min_over_ranks")
    if train and "error" not in train:
        line["train_step"] = {k: train[k] for k in ("ms_per_step", "images_per_s", "curve_layer_share_of_step") if k in train}
        if meta.get("gpus_visible", world) < world:
            line["rehearsal"] = f"{world} ranks share {meta['gpus_visible']} GPU(s) over {meta.get('backend')}: not a scaling measurement"
    line["detail"] = "bench_detail.json (also on stderr)"
    return line


def write_detail(detail):
    """Everything that is not the line: to bench_detail.json beside bench.py, to gpurun_out/ (what a GPU box hands back)
    and to stderr."""
    txt = json.dumps(_strict(detail), allow_nan=False, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                f.write(txt + "\n")
        except OSError:
            pass
    sys.stderr.write(json.dumps(_strict(detail), allow_nan=False) + "\n")
    sys.stderr.flush()


def self_launch(args):
    """`bench.py --gpus N` started without a launcher: start N ranks as a CHILD process (never exec: this process
    may not be replaced once a GPU is initialised, and nothing here has touched one yet) and relay its result."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--workload", default="layer", choices=sorted(WORKLOADS))
    ap.add_argument("--no-extras", action="store_true", help="skip other_workloads / cpu_baseline / accuracy")
    ap.add_argument("--full", action="store_true", help="N > 1: measure the whole workload table, as N = 1 does")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world:
        # checked from the launcher's environment BEFORE anything initialises the GPU or a process group: every rank
        # leaves at once, nothing is left blocked in a collective's teardown
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}: start it plainly (it launches its own ranks) "
                 f"or with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: this path has no CPU fallback")
    # one rank per GPU.  With fewer GPUs than ranks (a rehearsal on a 1-GPU box) ranks wrap onto the GPUs there are
    # and the control plane falls back to gloo (RCCL cannot place two ranks on one device); the line says so.
    n_dev = torch.cuda.device_count()
    device = torch.device("cuda", local % n_dev)
    torch.cuda.set_device(device)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CURL_DIST_BACKEND", "nccl" if n_dev >= world else "gloo")  # "nccl" is RCCL on ROCm
        # (gloo's C++ side prints its "[Gloo] Rank n is connected to ..." lines on fd 1: they go to stderr, stdout carries the
        # ONE JSON line and nothing else)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist_mod.init_process_group(backend="nccl", device_id=device)
            else:
                dist_mod.init_process_group(backend=backend)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        dist = dist_mod

    from curl_amd import _lib, ops
    _lib.load()  # fail loudly without the HIP library

    B = args.batch
    sets = make_inputs(B, device, seed=rank)  # each rank synthesises ITS shard of the global batch
    masks = {"disk": disk_mask(B, H_IMG, W_IMG, device),
             "ones": torch.ones(B, 1, H_IMG, W_IMG, dtype=torch.bool, device=device), None: None}

    def measure(name, steps, warmup, cold=False):
        w = WORKLOADS[name]
        bpp = w["bpp"]
        step = make_step(name, ops, masks, sets)
        npx_rank = workload_pixels(name, B)
        cold_us = cold_first_launch_us(step, sets, device) if cold else None
        cold_ms = cold_start_run(step, sets, steps, warmup, device) if cold and not args.no_extras else None
        wall, dev_ms, dev_ms_min = timed_run(step, sets, steps, warmup, dist, device)
        power = None
        if cold and rank == 0 and not args.no_extras:
            # the hwmon power reading is a slow average (the ~0.3 s timed region is too short for it): the same step
            # is kept running for 3 more seconds, outside the timing, and the second half of the samples is reported
            with BoardPower(device.index or 0) as bp:
                t_end = time.perf_counter() + 3.0
                i = 0
                while time.perf_counter() < t_end:
                    for _ in range(200):
                        step(sets[i % len(sets)])
                        i += 1
                    torch.cuda.synchronize(device)
            bp.samples = bp.samples[len(bp.samples) // 2:]
            power = bp.summary()
        mpix = world * npx_rank * steps / wall / 1e6
        gbps = npx_rank * bpp / (dev_ms * 1e-3) / 1e9
        tflops = npx_rank * w["flop_px"] / (dev_ms * 1e-3) / 1e12
        # the PMC passes were taken at bs32 (the CONFIG5 rows: at their own image counts, which B = 32 leaves as they are)
        # (a kernel-fragment entry stands for the float32 forward workloads of that kernel only: the byte-edge rows and the
        # mask-first variant move other bytes)
        traffic, traffic_src, valu_util = (load_traffic(w["frag"] if bpp >= 24.0 and not name.endswith("mask_first") else None, name)
                                           if B == 32 else (None, None, None))
        hbm = {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
               "frac_of_measured_copy_ceiling_6585": gbps / COPY_CEILING_GBPS}
        valu = {"achieved": tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / VALU_PEAK_TFLOPS,
                "flop_per_px": w["flop_px"]}
        roof = dict(hbm if w["bound"] == "hbm" else valu)
        # ONE clock for the two headline figures: `achieved` (and `frac`) from the HIP events on the launch stream, as the
        # contract asks; `achieved_wall` / `frac_wall` the same quantity from the wall clock `value` is computed from
        gbps_wall = npx_rank * bpp / (wall / steps) / 1e9
        roof.update({"bound": w["bound"], "traffic": traffic, "traffic_source": traffic_src,
                     "valu_issue_util": valu_util,
                     "achieved_wall": gbps_wall if w["bound"] == "hbm" else npx_rank * w["flop_px"] / (wall / steps) / 1e12,
                     "frac_wall": (gbps_wall / HBM_PEAK_GBPS if w["bound"] == "hbm"
                                   else npx_rank * w["flop_px"] / (wall / steps) / 1e12 / VALU_PEAK_TFLOPS),
                     "value_from_event_clock_Mpix_s": world * npx_rank / (dev_ms * 1e-3) / 1e6,
                     "algorithmic_bytes_per_px": bpp, "px_per_launch": npx_rank,
                     "secondary": {"bound": "valu", **valu} if w["bound"] == "hbm" else {"bound": "hbm", **hbm}})
        res = {"workload": w["desc"], "value": mpix, "ms_per_step": wall / steps * 1e3, "device_ms_per_step": dev_ms,
               "device_ms_per_step_min_over_ranks": dev_ms_min, "roofline": roof}
        if cold_us is not None:
            res["cold_first_launch_us"] = cold_us
        if cold_ms is not None:
            res["cold_start"] = {"ms_per_step": cold_ms, "value": npx_rank / (cold_ms * 1e-3) / 1e6, "unit": "Mpix/s per GPU",
                                 "protocol": f"0.5 s idle, {warmup} warm-up steps, {steps} timed steps, wall clock, no clock-settle "
                                             "launches (this rank alone)"}
        if power is not None:
            res["power"] = power
        return res

    main_res = measure(args.workload, args.steps, args.warmup, cold=True)
    # N = 1: the whole table.  N > 1 (the scaling run): the headline, the two kernels north_star's targets name and the
    # data-parallel train step -- the time goes to the scaling number; --full restores the table
    names = [n for n in WORKLOADS if n != args.workload]
    if world > 1 and not args.full:
        names = [n for n in SCALING_RUN_WORKLOADS if n != args.workload]
    others = {}
    if not args.no_extras:
        for name in names:
            # >= 200 timed launches (>= 40 ms) each: the Lab-stage figure the 70 % target is quoted on rides here
            others[name] = measure(name, max(200, args.steps // 2), max(3, args.warmup // 2))

    def run_train_step():
        # Every rank takes part (DDP's all-reduce), so a failure on ONE rank must not leave the others waiting in a
        # collective: the ranks first agree (one MIN all-reduce) that each of them could build the model and run a step
        # alone; only then does the data-parallel measurement start.  Context only, never at the expense of the line.
        if args.no_extras or os.environ.get("CURL_BENCH_TRAIN_STEP", "1") == "0":
            return None
        train = None
        try:
            ok = 1.0
            try:
                train_step_preflight(device)
            except Exception as e:
                ok, train = 0.0, {"error": "preflight: " + repr(e)}
            if dist is not None:
                t = torch.tensor([ok], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = float(t[0])
            if ok == 1.0:
                train = train_step_record(device, rank, world, local, dist)
            elif train is None:
                train = {"error": "skipped: another rank failed the preflight"}
        except Exception as e:
            train = {"error": repr(e)}
        return train

    # N = 1: the train step is measured first and its three scalars ride in the line.  N > 1: THE LINE GOES OUT FIRST -- the
    # data-parallel train step is the one part of this program that needs every rank to answer a collective, and a rank
    # lost in it must not cost the scaling run its headline; its record goes to bench_detail.json / stderr afterwards.
    train = run_train_step() if world == 1 else None
    meta = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "batch_per_gpu": B, "backend": backend,
            "ranks_seen": world if dist is None else dist.get_world_size(), "gpus_visible": n_dev}
    detail = None
    if rank == 0:
        accuracy = end = cpu = None
        if not args.no_extras:
            accuracy = accuracy_vs_oracle(ops, device)
            if world == 1 or args.full:
                try:
                    end = end_to_end(ops, device, sets, masks)
                except Exception as e:  # context only: never at the expense of the line
                    end = {"error": repr(e)}
            # rank 0's host cores, after every timed region (the other ranks wait at the next collective)
            cpu = cpu_baseline()
        line = make_line(main_res, others, accuracy, cpu, meta, train)
        detail = {"line": line, "headline": main_res, "other_workloads": others, "accuracy": accuracy,
                  "end_to_end": end, "train_step": train, "cpu_baseline": cpu, "meta": meta}
        print(dump_line(line))
        sys.stdout.flush()
    if world > 1:
        # (nothing may follow the line on stdout: whatever a library prints from here on goes to stderr)
        sys.stdout.flush()
        os.dup2(2, 1)
        # ... and nothing may keep the scaling run from ending: the headline is out; if the data-parallel step (MIOpen's solver
        # search, DDP's bucket all-reduces over RCCL) has not come back in CURL_BENCH_TRAIN_TIMEOUT seconds, every rank's own
        # watchdog ends its process with the code of a finished run.
        import threading
        limit = float(os.environ.get("CURL_BENCH_TRAIN_TIMEOUT", 240))

        def give_up():
            sys.stderr.write(f"bench.py: rank {rank}: the data-parallel train step did not finish in {limit:.0f} s; the line is "
                             "out, ending the run without its record\n")
            sys.stderr.flush()
            if detail is not None:
                detail["train_step"] = {"error": f"timed out after {limit:.0f} s"}
                try:
                    write_detail(detail)
                except Exception:
                    pass
            os._exit(0)
        dog = threading.Timer(limit, give_up)
        dog.daemon = True
        dog.start()
        train = run_train_step()
        dog.cancel()
        if detail is not None:
            detail["train_step"] = train
    if detail is not None:
        write_detail(detail)
    if dist is not None:
        import threading
        dog2 = threading.Timer(120.0, lambda: os._exit(0))  # (a rank that never arrives must not hold the others' exit)
        dog2.daemon = True
        dog2.start()
        try:  # (the line is out: a peer that left early -- its watchdog, a failed rank -- is reported, not raised)
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            sys.stderr.write(f"bench.py: rank {rank}: closing barrier: {e!r}\n")
        dog2.cancel()


if __name__ == "__main__":
    main()
