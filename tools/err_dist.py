"""Error distribution of the fused layer / Lab stage of TWO builds of the library against the oracle (float32 restatement
of the reference) and its float64 evaluation, on the same inputs -- to judge an arithmetic change by what it does to
parity, not only to time.

    python tools/err_dist.py curl_amd/lib/libcurlhip.so curl_amd/lib/variants/libcurlhip_pow24_direct.so
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import curl_oracle as O  # noqa: E402
from ab import bind  # noqa: E402

dev = torch.device("cuda:0")
libs = {"A": bind(sys.argv[1]), "B": bind(sys.argv[2])}
H, W = 512, 768
stream = torch.cuda.current_stream().cuda_stream


def layer(lib, what, img, mask, L, R, Hk):
    B = img.shape[0]
    out, reg = torch.empty_like(img), torch.empty(B, device=dev)
    nb = lib.curl_workspace_bytes(B, 160)
    ws = torch.empty(nb // 4, device=dev)
    if what == "layer":
        rc = lib.curl_layer_fwd_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), out.data_ptr(),
                                    reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16, 16, 16, 0, stream)
    else:
        rc = lib.curl_lab_stage_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), out.data_ptr(), reg.data_ptr(), ws.data_ptr(),
                                    nb, B, H, W, 16, 0, stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.cpu()


print(f"A = {sys.argv[1]}\nB = {sys.argv[2]}")
for what in ("layer", "lab_stage"):
    for sigma in (0.1, 0.3):
        for case in ("uniform", "grid8", "dark", "bright"):
            g = torch.Generator().manual_seed(int(sigma * 100) + len(case))
            img = torch.rand(1, 3, H, W, generator=g)
            if case == "grid8":
                img = (img * 255).floor() / 255
            if case == "dark":
                img = img * 0.15  # shadows: u near the sRGB knee, the largest |log2 u|
            if case == "bright":
                img = 0.7 + 0.3 * img
            L, R, Hk = (torch.randn(1, n, generator=g) * sigma for n in (48, 48, 64))
            mask = torch.ones(1, 1, H, W)
            if what == "layer":
                ref, _ = O.curl_layer(img, mask, L, R, Hk)
                r64, _ = O.curl_layer(img.double(), mask.double(), L.double(), R.double(), Hk.double())
            else:
                ref, _ = O.lab_stage(img, mask, L)
                r64, _ = O.lab_stage(img.double(), mask.double(), L.double())
            noise = (ref.double() - r64).abs()
            row = f"{what:9s} sigma {sigma} {case:8s} ref32-vs-64: max {float(noise.max()):.2e} mean {float(noise.mean()):.2e} |"
            m8 = mask.to(torch.uint8).to(dev)
            for k in ("A", "B"):
                out = layer(libs[k], what, img.to(dev), m8, L.to(dev), R.to(dev), Hk.to(dev)).double()
                d32, d64 = (out - ref.double()).abs(), (out - r64).abs()
                row += (f" {k}: vs ref32 max {float(d32.max()):.2e} q.9999 {float(np.quantile(d32.numpy(), 0.9999)):.2e} "
                        f">1e-5 {float((d32 > 1e-5).double().mean()):.1e} mean {float(d32.mean()):.2e}; vs f64 max {float(d64.max()):.2e} "
                        f"mean {float(d64.mean()):.2e} |")
            print(row, flush=True)
