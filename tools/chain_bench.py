"""adjust_rgb at bs32 x 1500x1000 in its three evaluation modes: affine collapse (default), the reference's in-order
float32 sum (CURL_F_EXACT_ORDER, bit-identical), the paper's clamped piecewise-linear curve (CURL_F_PWL)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curl_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 32, 1000, 1500
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
R = torch.randn(B, 48, device=dev) * 0.1
Hk = torch.randn(B, 64, device=dev) * 0.1
_lib.load()
for name, fn, raw in (("adjust_rgb", ops.adjust_rgb, R), ("adjust_hsv", ops.adjust_hsv, Hk)):
    for mode, flags in (("affine collapse", 0), ("exact order", _lib.F_EXACT_ORDER), ("paper PWL", _lib.F_PWL)):
        for i in range(60):
            fn(imgs[i & 1], raw, flags=flags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 100
        for i in range(n):
            fn(imgs[i & 1], raw, flags=flags)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{name:11s} {mode:16s} {ms*1e3:8.1f} us  {B*H*W*24/ms/1e6:7.0f} GB/s  {B*H*W/ms/1e3:9.0f} Mpix/s")
