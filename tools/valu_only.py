"""Diagnostics: arithmetic-only (CURL_F_DIAG_NO_MEM) vs full timing of the fused kernels, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import disk_mask  # noqa: E402
from curl_amd import _lib, ops  # noqa: E402

B, H, W = 32, 1000, 1500
dev = torch.device("cuda:0")
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
out = torch.empty_like(imgs[0])
L = torch.randn(B, 48, device=dev) * 0.1
R = torch.randn(B, 48, device=dev) * 0.1
Hk = torch.randn(B, 64, device=dev) * 0.1
mask = disk_mask(B, H, W, dev)
_lib.load()
cnt = [0]


def run(name, flags):
    cnt[0] += 1
    img = imgs[cnt[0] & 1]
    if name == "layer":
        ops.curl_layer_forward(img, mask, L, R, Hk, flags=flags, out=out)
    elif name == "lab_stage":
        ops.lab_stage(img, mask, L, flags=flags, out=out)
    elif name == "rgb2lab":
        ops.rgb2lab(img, flags=flags)
    elif name == "lab2rgb":
        ops.lab2rgb(img, flags=flags)
    elif name == "rgb2hsv":
        ops.rgb2hsv(img, flags=flags)
    elif name == "hsv2rgb":
        ops.hsv2rgb(img, flags=flags)
    elif name == "adjust_rgb":
        ops.adjust_rgb(img, R, flags=flags)


variants = [(n, u, d) for n in ("layer", "lab_stage", "rgb2lab", "lab2rgb", "rgb2hsv", "hsv2rgb", "adjust_rgb")
            for u in (1, 2) for d in (0, _lib.F_DIAG_NO_MEM)]
times = {v: [] for v in variants}
for v in variants:
    run(v[0], (v[1] << 8) | v[2])
torch.cuda.synchronize()
for r in range(5):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(v[0], (v[1] << 8) | v[2])
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 10)
for v in variants:
    t = sorted(times[v])
    print(f"{v[0]:12s} U={v[1]} {'VALU-only' if v[2] else 'full     '} median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us")
