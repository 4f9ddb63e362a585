"""Timing of the polynomial path kernels (sustained), bs32 x 1500x1000."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curl_amd import _lib, ops  # noqa: E402

B, H, W = int(os.environ.get("B", 32)), 1000, 1500
dev = torch.device("cuda:0")
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
c126 = torch.randn(B, 3, 3, 126, device=dev) * 0.2
c35 = torch.randn(B, 3, 3, 35, device=dev) * 0.2
_lib.load()
for name, c in (("trispace spatial (126)", c126), ("trispace non-spatial (35)", c35)):
    for nomem in (0, _lib.F_DIAG_NO_MEM):
        for i in range(10):
            ops.trispace_forward(imgs[i & 1], c, flags=nomem)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 40
        for i in range(n):
            ops.trispace_forward(imgs[i & 1], c, flags=nomem)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{name:28s} {'VALU-only' if nomem else 'full     '} {ms*1e3:8.1f} us  {B*H*W/ms/1e6:7.1f} Gpix/s  "
              f"{B*H*W*24/ms/1e6:7.0f} GB/s algorithmic")

# backward (coefficient gradients): the training shape (main.py crops) and the full-frame batch
for (b, h, w) in ((32, 256, 256), (8, 1000, 1500)):
    img = torch.rand(b, 3, h, w, device=dev)
    g = torch.randn(b, 3, h, w, device=dev)
    for name, c in (("trispace bwd (126)", c126[:b]), ("trispace bwd (35)", c35[:b])):
        for i in range(3):
            ops.trispace_backward(img, c, g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for i in range(n):
            ops.trispace_backward(img, c, g)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{name:20s} {b}x{h}x{w}  {ms*1e3:9.1f} us  {b*h*w/ms/1e6:7.2f} Gpix/s")
