"""Timing of the SURVEY 8f-3 kernels (masked PSNR, CURLLoss pointwise terms forward / backward) against their HBM bytes,
at the bench batch (32 x 1500 x 1000) and the training crop batch (32 x 256 x 256).

    python tools/f3_bench.py [variant]          (default: the product library)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:  # an experiment build instead of the product library (read when curl_amd._lib is imported)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import variants  # noqa: E402
    os.environ["CURL_HIP_LIB"] = variants.path(sys.argv[1])
from curl_amd import _lib, ops  # noqa: E402

_lib.load()
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W) in ((32, 1000, 1500), (32, 256, 256)):
    torch.manual_seed(0)
    a = torch.rand(B, 3, H, W, device=dev)
    b = torch.rand(B, 3, H, W, device=dev)
    mask = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
    w4 = torch.tensor([1.0, 1.0, 1.0, 1.0], device=dev)
    gL = torch.rand(B, 1, H, W, device=dev)
    px = B * H * W
    for name, fn, bpp in (("psnr (bool mask)", lambda: ops.psnr_per_image(a, b, mask), 25),
                          ("loss terms fwd (+2 L planes)", lambda: ops.loss_term_sums(a, b, mask), 33),
                          ("loss terms fwd (sums only)", lambda: ops.loss_term_sums(a, b, mask, want_L=False), 25),
                          ("loss terms bwd", lambda: ops.loss_terms_backward(a, b, mask, w4, gL), 41)):
        us = timeit(fn)
        print(f"{B}x{H}x{W}  {name:30s} {us:9.1f} us  {px * bpp / us / 1e6:7.2f} TB/s algorithmic ({bpp} B/px)  "
              f"= {px * bpp / us / 1e6 / 8.0:5.2f} of the HBM peak", flush=True)
    # the byte edges (transpose.py + to_tensor / mul(255).byte()), stand-alone
    u8 = (a * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    for name, fn, bpp in (("u8 HWC -> f32 CHW", lambda: ops.u8hwc_to_f32chw(u8), 15),
                          ("f32 CHW -> u8 HWC", lambda: ops.f32chw_to_u8hwc(a), 15),
                          ("compose white + u8 HWC (bool mask)", lambda: ops.compose_white_u8hwc(a, mask), 16)):
        us = timeit(fn)
        print(f"{B}x{H}x{W}  {name:30s} {us:9.1f} us  {px * bpp / us / 1e6:7.2f} TB/s algorithmic ({bpp} B/px)  "
              f"= {px * bpp / us / 1e6 / 8.0:5.2f} of the HBM peak", flush=True)
