"""The process rocprofv3 wraps for a look at the polynomial backward alone: N calls of curl_trispace_bwd_f32 at one shape.
    rocprofv3 --kernel-trace --pmc ... -- python3 tools/poly_bwd_run.py [B H W] [calls]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import ops  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 1000, 1500)
n = int(sys.argv[4]) if len(sys.argv) >= 5 else 40
dev = torch.device("cuda:0")
torch.manual_seed(0)
img = torch.rand(B, 3, H, W, device=dev)
gout = torch.randn(B, 3, H, W, device=dev)
c = torch.randn(B, 3, 3, 126, device=dev) * 0.2
for _ in range(n):
    g = ops.trispace_backward(img, c, gout)
torch.cuda.synchronize()
print("ok", float(g.abs().sum()))
