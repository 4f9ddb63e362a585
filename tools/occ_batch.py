"""Where does the occupancy cap start to pay?  Operators at 1..32 images of 1500x1000: library tile shape without cap
(CURL_F_TUNE_OCC = 1) against one group per lane at the op's kResident (forced through the flags), and the library default.

    python tools/occ_batch.py
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, W = 1000, 1500
OCC1 = 1 << 19
RES = {"rgb2lab": 4, "hsv2rgb": 3, "adjust_rgb": 3, "lab_stage": 7, "hsv_stage": 4}


def window(fn, n):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'operator':12s} {'B':>3s} {'tiles':>7s} {'no cap':>8s} {'capped':>8s} {'default':>8s}  capped vs no cap")
for B in (1, 2, 3, 4, 6, 8, 12, 16, 32):
    torch.manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    mask = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
    L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
    out = torch.empty_like(imgs[0])
    cnt = [0]

    def img():
        cnt[0] += 1
        return imgs[cnt[0] & 1]

    OPS = {"rgb2lab": lambda f: ops.rgb2lab(img(), flags=f), "hsv2rgb": lambda f: ops.hsv2rgb(img(), flags=f),
           "adjust_rgb": lambda f: ops.adjust_rgb(img(), R, flags=f), "lab_stage": lambda f: ops.lab_stage(img(), mask, L, flags=f, out=out),
           "hsv_stage": lambda f: ops.hsv_stage(img(), mask, Hk, flags=f, out=out)}
    n = max(100, 3200 // B)
    for name, fn in OPS.items():
        forced = (1 << 8) | (RES[name] << 19)
        t = {k: [] for k in ("off", "cap", "def")}
        for r in range(3):
            for k, f in (("off", OCC1), ("cap", forced), ("def", 0)) if r % 2 == 0 else (("def", 0), ("cap", forced), ("off", OCC1)):
                t[k].append(window(lambda: fn(f), n))
        m = {k: statistics.median(v) for k, v in t.items()}
        print(f"{name:12s} {B:3d} {B * 1465:7d} {m['off']:8.1f} {m['cap']:8.1f} {m['def']:8.1f}  {(m['cap'] / m['off'] - 1) * 100:+.1f} %", flush=True)
