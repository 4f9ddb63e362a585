"""Time CURLLoss forward+backward at the training shape, with and without the MS-SSIM term (stock torch convs)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curl_amd import model  # noqa: E402

dev = torch.device("cuda:0")
B, S = int(os.environ.get("B", 32)), int(os.environ.get("S", 256))
torch.manual_seed(0)
tgt = torch.rand(B, 3, S, S, device=dev)
pred = (tgt + 0.05 * torch.randn(B, 3, S, S, device=dev)).clamp(0, 1).requires_grad_(True)
mask = torch.rand(B, 1, S, S, device=dev) > 0.2
for name, crit in (("with MS-SSIM", model.CURLLoss().to(dev)), ("pointwise terms only", model.CURLLoss(msssim_layer=None).to(dev))):
    for _ in range(5):
        crit(pred, tgt, mask).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        pred.grad = None
        crit(pred, tgt, mask).backward()
    torch.cuda.synchronize()
    print(f"CURLLoss fwd+bwd {name:22s} {B}x{S}x{S}: {(time.perf_counter() - t0) / n * 1e3:8.3f} ms")
