"""Ad-hoc parity stress on the GPU: the conditioning bound of DESIGN.md 4 (|HIP - ref32| <= max(1e-5, 2e-6 * S)) over
knot spreads, seeds and mask kinds at 512x768, against the oracle.  Prints the worst ratio per case.

    python tools/parity_stress.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import curl_oracle as O  # noqa: E402
from curl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, W = 512, 768
worst = 0.0
for sigma in (0.1, 0.3, 0.5):
    for seed in range(4):
        g = torch.Generator().manual_seed(1000 * seed + int(sigma * 100))
        img = torch.rand(1, 3, H, W, generator=g)
        if seed == 1:
            img = (img * 255).floor() / 255  # 8-bit grid: channel ties
        L, R, Hk = (torch.randn(1, n, generator=g) * sigma for n in (48, 48, 64))
        kind = ("none", "bool", "f32", "bool")[seed]
        mask = torch.ones(1, 1, H, W) if kind == "none" else (torch.rand(1, 1, H, W, generator=g) > 0.2).float() if kind == "bool" \
            else torch.rand(1, 1, H, W, generator=g)
        ref, rreg = O.curl_layer(img, mask, L, R, Hk)
        r64, _ = O.curl_layer(img.double(), mask.double(), L.double(), R.double(), Hk.double())
        S = O.input_sensitivity(img, mask, L, R, Hk, r64=r64)
        m = None if kind == "none" else (mask.bool().to(dev) if kind == "bool" else mask.to(dev))
        out, reg = ops.curl_layer_forward(img.to(dev), m, L.to(dev), R.to(dev), Hk.to(dev))
        d = (out.cpu().double() - ref.double()).abs().amax(1)
        noise = (ref.double() - r64).abs().amax(1)
        bound = torch.clamp(2e-6 * S, min=1e-5)
        ratio = float((d / bound).max())
        nratio = float((noise / bound).max())
        worst = max(worst, ratio)
        print(f"sigma {sigma} seed {seed} mask {kind:4s}: max|d| {float(d.max()):.2e}  over1e-5 {float((d > 1e-5).double().mean()):.1e}  "
              f"max d/bound {ratio:.2f}  (reference's own float32 noise / bound {nratio:.2f})  S>5: {float((S > 5).double().mean()):.3f}  "
              f"reg rel {float(((reg.cpu() - rreg).abs() / rreg.abs().clamp_min(1e-12)).max()):.1e}", flush=True)
print("worst d/bound", worst)
