"""Diagnostics: arithmetic-only (CURL_F_DIAG_NO_MEM) vs full timing of the fused kernels, interleaved.

Both legs see the SAME mask: CURL_F_DIAG_NO_MEM synthesises an all-ones mask in registers, so the full leg runs
with an all-ones bool mask too (round 1 ran it with the 70 % disk mask, whose masked-out waves skip the
arithmetic, and reported "full < VALU-only").  The disk-mask timing is printed as a separate, labelled row.
Measurements start after the clock has settled (SETTLE back-to-back launches) and use 100-launch windows.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import disk_mask  # noqa: E402
from curl_amd import _lib, ops  # noqa: E402

B, H, W = 32, 1000, 1500
SETTLE, WINDOW, ROUNDS = 300, 100, 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
out = torch.empty_like(imgs[0])
L = torch.randn(B, 48, device=dev) * 0.1
R = torch.randn(B, 48, device=dev) * 0.1
Hk = torch.randn(B, 64, device=dev) * 0.1
ones = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
disk = disk_mask(B, H, W, dev)
_lib.load()
cnt = [0]


def run(name, flags, mask):
    cnt[0] += 1
    img = imgs[cnt[0] & 1]
    if name == "layer":
        ops.curl_layer_forward(img, mask, L, R, Hk, flags=flags, out=out)
    elif name == "lab_stage":
        ops.lab_stage(img, mask, L, flags=flags, out=out)
    elif name == "adjust_rgb":
        ops.adjust_rgb(img, R, flags=flags)
    else:
        getattr(ops, name)(img, flags=flags)


names = sys.argv[1:] or ["layer", "lab_stage", "rgb2lab", "lab2rgb", "rgb2hsv", "hsv2rgb", "adjust_rgb"]
# (name, U, diag flag, mask label)
variants = []
for n in names:
    for u in (1, 2):
        variants += [(n, u, 0, "ones"), (n, u, _lib.F_DIAG_NO_MEM, "ones")]
    if n in ("layer", "lab_stage"):
        variants.append((n, 1, 0, "disk70"))
times = {v: [] for v in variants}
for _ in range(SETTLE):
    run("layer", 0, ones)
torch.cuda.synchronize()
for r in range(ROUNDS):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        m = ones if v[3] == "ones" else disk
        for _ in range(20):
            run(v[0], (v[1] << 8) | v[2], m)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(WINDOW):
            run(v[0], (v[1] << 8) | v[2], m)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / WINDOW)
print(f"# bs{B} x {W}x{H}; us per call incl. the ~5 us knot-prep launch; {ROUNDS} windows of {WINDOW} after {SETTLE} settle launches")
for v in variants:
    t = sorted(times[v])
    print(f"{v[0]:12s} U={v[1]} mask={v[3]:6s} {'VALU-only' if v[2] else 'full     '} median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us")
