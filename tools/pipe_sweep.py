"""(Needs tools/experiments/patches/r04_pipelined_tile_walk.patch applied: the variant lost and is not in the product.)
Sweep of the pipelined tile walk (CURL_F_TUNE_PIPE) against the library default, per operator, at bs32 x 1500x1000:
paired per-round differences of alternating 200-launch windows (tools/ab.py's protocol), one library, flags only.

    python tools/pipe_sweep.py [op ...]        ops: lab_stage hsv_stage adjust_rgb rgb2lab layer
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("B", 32)), 1000, 1500
torch.manual_seed(0)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
out = torch.empty_like(imgs[0])
mask = torch.ones(B, 1, H, W, dtype=torch.uint8, device=dev)
L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
reg = torch.empty(B, device=dev)
nb = lib.curl_workspace_bytes(B, 160)
ws = torch.empty(nb // 4, device=dev)
stream = torch.cuda.current_stream().cuda_stream
cnt = [0]


def run(op, flags):
    cnt[0] += 1
    img = imgs[cnt[0] & 1]
    if op == "layer":
        rc = lib.curl_layer_fwd_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), out.data_ptr(),
                                    reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16, 16, 16, flags, stream)
    elif op == "lab_stage":
        rc = lib.curl_lab_stage_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb,
                                    B, H, W, 16, flags, stream)
    elif op == "hsv_stage":
        rc = lib.curl_hsv_stage_f32(img.data_ptr(), mask.data_ptr(), 1, Hk.data_ptr(), out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb,
                                    B, H, W, 16, flags, stream)
    elif op == "adjust_rgb":
        rc = lib.curl_adjust_rgb_f32(img.data_ptr(), R.data_ptr(), out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16, flags, stream)
    else:
        rc = lib.curl_rgb2lab_f32(img.data_ptr(), out.data_ptr(), B, H, W, flags, stream)
    assert rc == 0, lib.curl_last_error()


def window(op, flags, n=200):
    for _ in range(20):
        run(op, flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run(op, flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


U1 = 1 << 8
ops = sys.argv[1:] or ["lab_stage", "hsv_stage", "adjust_rgb", "rgb2lab", "layer"]
rounds = int(os.environ.get("ROUNDS", 5))
for op in ops:
    for _ in range(150):
        run(op, 0)
    torch.cuda.synchronize()
    ref = out.clone()
    run(op, 0)
    torch.cuda.synchronize()
    want = out.clone()
    variants = [(p, k) for p in (1, 2) for k in ((2, 3, 4, 6, 1) if op != "layer" else (6, 7, 1))]
    res = {}
    for (p, k) in variants:
        fl = U1 | (k << 19) | (p << 25)
        cnt[0] = 1  # same input as `want`
        run(op, fl)
        torch.cuda.synchronize()
        cnt[0] = 0
        run(op, 0)
        torch.cuda.synchronize()
        base = out.clone()
        cnt[0] = 0
        run(op, fl)
        torch.cuda.synchronize()
        same = torch.equal(out, base)
        d, a, b = [], [], []
        for r in range(rounds):
            order = ((0, "a"), (fl, "b")) if r % 2 == 0 else ((fl, "b"), (0, "a"))
            t = {}
            for f, nm in order:
                t[nm] = window(op, f)
            a.append(t["a"]), b.append(t["b"]), d.append((t["b"] - t["a"]) / t["a"] * 100)
        res[(p, k)] = statistics.median(d)
        print(f"{op:10s} T={1 << p} resident={'uncapped' if k == 1 else k}: default {statistics.median(a):7.1f} us, pipelined "
              f"{statistics.median(b):7.1f} us, paired {statistics.median(d):+.2f} % [{min(d):+.2f}, {max(d):+.2f}]  bit-identical: {same}", flush=True)
