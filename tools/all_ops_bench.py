"""Every public operator of the library at the bench batch (32 x 1500 x 1000; backward passes at 8 images), sustained, against
its algorithmic HBM bytes: one table for DESIGN.md 5 (which rows are memory-bound and how close, which are arithmetic-bound).

    python tools/all_ops_bench.py [n_launches]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import _lib, ops  # noqa: E402

_lib.load()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, H, W = 32, 1000, 1500
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
mask_b = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
mask_f = torch.ones(B, 1, H, W, device=dev)
L, R, Hk = (torch.randn(B, k, device=dev) * 0.1 for k in (48, 48, 64))
C16 = torch.randn(B, 16, device=dev) * 0.1
c126 = torch.randn(B, 3, 3, 126, device=dev) * 0.2
c35 = torch.randn(B, 3, 3, 35, device=dev) * 0.2
img5 = torch.rand(8, 5, H, W, device=dev)
u8 = (imgs[0] * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
lab = ops.rgb2lab(imgs[0])
hsv = ops.rgb2hsv(imgs[0])
w4 = torch.ones(4, device=dev)
gL = torch.rand(B, 1, H, W, device=dev)
b8 = slice(0, 8)
g8 = torch.randn(8, 3, H, W, device=dev)


def timeit(fn):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(N):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


ROWS = [  # name, images, bytes per pixel (read + written), fn
    ("rgb2lab", B, 24, lambda i: ops.rgb2lab(imgs[i & 1])),
    ("lab2rgb", B, 24, lambda i: ops.lab2rgb(lab)),
    ("rgb2hsv", B, 24, lambda i: ops.rgb2hsv(imgs[i & 1])),
    ("hsv2rgb", B, 24, lambda i: ops.hsv2rgb(hsv)),
    ("apply_curve (affine form)", B, 24, lambda i: ops.apply_curve(imgs[i & 1], C16, None, 0, 0, flags=0)),
    ("adjust_rgb", B, 24, lambda i: ops.adjust_rgb(imgs[i & 1], R)),
    ("adjust_lab", B, 24, lambda i: ops.adjust_lab(lab, L)),
    ("adjust_hsv", B, 24, lambda i: ops.adjust_hsv(hsv, Hk)),
    ("lab_stage, no mask", B, 24, lambda i: ops.lab_stage(imgs[i & 1], None, L)),
    ("lab_stage, bool mask", B, 25, lambda i: ops.lab_stage(imgs[i & 1], mask_b, L)),
    ("hsv_stage, bool mask", B, 25, lambda i: ops.hsv_stage(imgs[i & 1], mask_b, Hk)),
    ("layer, no mask", B, 24, lambda i: ops.curl_layer_forward(imgs[i & 1], None, L, R, Hk)),
    ("layer, bool mask", B, 25, lambda i: ops.curl_layer_forward(imgs[i & 1], mask_b, L, R, Hk)),
    ("layer, float mask", B, 28, lambda i: ops.curl_layer_forward(imgs[i & 1], mask_f, L, R, Hk)),
    ("layer, u8 HWC in/out", B, 7, lambda i: ops.curl_layer_forward_u8hwc(u8, mask_b, L, R, Hk)),
    ("polynomial path (126)", B, 24, lambda i: ops.trispace_forward(imgs[i & 1], c126)),
    ("polynomial path (35)", B, 24, lambda i: ops.trispace_forward(imgs[i & 1], c35)),
    ("polynomial path, u8 HWC in/out", B, 6, lambda i: ops.trispace_forward_u8hwc(u8, c126)),
    ("poly_layer (5 variables)", 8, 32, lambda i: ops.poly_layer(img5, c126[b8, 0])),
    ("u8 HWC -> f32 CHW", B, 15, lambda i: ops.u8hwc_to_f32chw(u8)),
    ("f32 CHW -> u8 HWC", B, 15, lambda i: ops.f32chw_to_u8hwc(imgs[i & 1])),
    ("white background + u8 HWC", B, 16, lambda i: ops.compose_white_u8hwc(imgs[i & 1], mask_b)),
    ("masked PSNR", B, 25, lambda i: ops.psnr_per_image(imgs[0], imgs[1], mask_b)),
    ("CURLLoss terms forward", B, 33, lambda i: ops.loss_term_sums(imgs[0], imgs[1], mask_b)),
    ("CURLLoss terms backward", B, 41, lambda i: ops.loss_terms_backward(imgs[0], imgs[1], mask_b, w4, gL)),
    ("layer backward (8 images)", 8, 37, lambda i: ops.curl_layer_backward(imgs[0][b8], mask_b[b8], L[b8], R[b8], Hk[b8], g8)),
    ("polynomial backward (8 images)", 8, 24, lambda i: ops.trispace_backward(imgs[0][b8], c126[b8], g8)),
]
print(f"# {N} launches each after 8 warm-up launches, inputs resident; B/px = algorithmic bytes read + written")
print(f"{'operator':34s} {'us':>9s} {'Gpix/s':>8s} {'B/px':>5s} {'TB/s':>6s}  of 8 TB/s")
for name, nb, bpp, fn in ROWS:
    us = timeit(fn)
    px = nb * H * W
    print(f"{name:34s} {us:9.1f} {px / us / 1e3:8.1f} {bpp:5d} {px * bpp / us / 1e6:6.2f}  {px * bpp / us / 1e6 / 8.0:5.2f}", flush=True)
