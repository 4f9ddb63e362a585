"""Sustained behaviour: 300 back-to-back launches per workload, mean time per window of 20 launches.
Shows the DVFS give-back of the arithmetic-heavy kernels (first windows fast, then the clock drops)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curl_amd import _lib, ops  # noqa: E402

B, H, W = 32, 1000, 1500
dev = torch.device("cuda:0")
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
out = torch.empty_like(imgs[0])
L = torch.randn(B, 48, device=dev) * 0.1
R = torch.randn(B, 48, device=dev) * 0.1
Hk = torch.randn(B, 64, device=dev) * 0.1
ones = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
_lib.load()
cnt = [0]
FLAGS = int(os.environ.get("CURL_FLAGS", "0"), 0)
names = sys.argv[1:] or ["rgb_only", "rgb2lab", "lab2rgb", "rgb2hsv", "lab_stage", "layer", "layer_nomem", "copy"]


def run(name):
    cnt[0] += 1
    img = imgs[cnt[0] & 1]
    if name == "layer":
        ops.curl_layer_forward(img, ones, L, R, Hk, out=out, flags=FLAGS)
    elif name == "layer_nomem":
        ops.curl_layer_forward(img, ones, L, R, Hk, out=out, flags=_lib.F_DIAG_NO_MEM | FLAGS)
    elif name == "lab_stage":
        ops.lab_stage(img, ones, L, out=out, flags=FLAGS)
    elif name == "rgb_only":
        ops.adjust_rgb(img, R, flags=FLAGS)
    elif name == "copy":
        out.copy_(img)
    else:
        getattr(ops, name)(img, flags=FLAGS)


for name in names:
    for _ in range(3):
        run(name)
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2e8))  # let the chip idle ~0.1 s so every workload starts from the same state
    torch.cuda.synchronize()
    n_win, per = 15, 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
    ev[0].record()
    for w in range(n_win):
        for _ in range(per):
            run(name)
        ev[w + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) / per * 1e3 for i in range(n_win)]
    print(f"{name:12s} us/launch per window of {per}: " + " ".join(f"{x:6.1f}" for x in t))
