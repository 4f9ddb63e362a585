"""Tile shape x occupancy sweep of the f32 streaming operators: float4 groups per lane U in {1, 2} (CURL_F_TUNE_UNROLL) x
resident workgroups per CU k in {default, 2..7} (CURL_F_TUNE_OCC), bs32 x 1500x1000, 200-launch windows, 3 interleaved
rounds, median.  Prints one row per operator and the best cell against the library default (flags = 0).

    python tools/occ_sweep.py [op ...]
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 32, 1000, 1500
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
mask = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
C = torch.exp(torch.randn(B, 16, device=dev) * 0.1)
out = torch.empty_like(imgs[0])
OPS = {
    "rgb2lab": lambda x, f: ops.rgb2lab(x, flags=f), "lab2rgb": lambda x, f: ops.lab2rgb(x, flags=f),
    "rgb2hsv": lambda x, f: ops.rgb2hsv(x, flags=f), "hsv2rgb": lambda x, f: ops.hsv2rgb(x, flags=f),
    "adjust_rgb": lambda x, f: ops.adjust_rgb(x, R, flags=f), "adjust_lab": lambda x, f: ops.adjust_lab(x, L, flags=f),
    "adjust_hsv": lambda x, f: ops.adjust_hsv(x, Hk, flags=f), "apply_curve": lambda x, f: ops.apply_curve(x, C, None, 0, 1, flags=f),
    "lab_stage": lambda x, f: ops.lab_stage(x, mask, L, flags=f, out=out), "lab_stage_nomask": lambda x, f: ops.lab_stage(x, None, L, flags=f, out=out),
    "hsv_stage": lambda x, f: ops.hsv_stage(x, mask, Hk, flags=f, out=out),
    "layer": lambda x, f: ops.curl_layer_forward(x, mask, L, R, Hk, flags=f, out=out),
}
CELLS = [(u, k) for u in (1, 2) for k in (0, 2, 3, 4, 5, 6, 7)]
LAUNCHES = int(os.environ.get("LAUNCHES", 200))


def window(fn, flags):
    for i in range(10):
        fn(imgs[i & 1], flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(LAUNCHES):
        fn(imgs[i & 1], flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / LAUNCHES * 1e3


names = sys.argv[1:] or list(OPS)
print("us per launch; columns: workgroups per CU (0 = no cap)")
print(f"{'operator':18s} {'default':>8s} | " + " ".join(f"U{u}k{k:<2d}" for u, k in CELLS))
for name in names:
    fn = OPS[name]
    for _ in range(100):
        fn(imgs[0], 0)
    t = {c: [] for c in [None] + CELLS}
    for r in range(3):
        for c in ([None] + CELLS if r % 2 == 0 else CELLS[::-1] + [None]):
            t[c].append(window(fn, 0 if c is None else (c[0] << 8) | (c[1] << 19)))
    med = {c: statistics.median(v) for c, v in t.items()}
    best = min(CELLS, key=lambda c: med[c])
    print(f"{name:18s} {med[None]:8.1f} | " + " ".join(f"{med[c]:5.1f}" for c in CELLS) +
          f" | best U={best[0]} k={best[1]}: {med[best]:.1f} us = {(med[best] / med[None] - 1) * 100:+.1f} %", flush=True)
