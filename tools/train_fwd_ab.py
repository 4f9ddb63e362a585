"""The train step's forward as ONE pass (curl_layer_loss_fwd_f32) against the two calls it replaces (curl_layer_fwd_f32 then
curl_loss_terms_f32), C ABI on preallocated buffers, alternating windows in one process.

    python tools/train_fwd_ab.py
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
for (B, H, W) in ((32, 1000, 1500), (8, 1000, 1500), (1, 1000, 1500), (32, 256, 256)):
    torch.manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    tgts = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    mask = torch.ones(B, 1, H, W, dtype=torch.uint8, device=dev)
    L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
    out = torch.empty_like(imgs[0])
    reg = torch.empty(B, device=dev)
    sums = torch.empty(B, 5, dtype=torch.float64, device=dev)
    Lp, Lt = (torch.empty(B, 1, H, W, device=dev) for _ in range(2))
    nb = lib.curl_workspace_bytes(B, 160)
    ws = torch.empty(nb // 4, device=dev)
    sb = lib.curl_loss_terms_scratch_bytes(B, H, W)
    scratch = torch.empty(sb // 4, device=dev)
    cnt = [0]

    def fused():
        cnt[0] += 1
        k = cnt[0] & 1
        rc = lib.curl_layer_loss_fwd_f32(imgs[k].data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), tgts[k].data_ptr(),
                                         out.data_ptr(), reg.data_ptr(), sums.data_ptr(), Lp.data_ptr(), Lt.data_ptr(), ws.data_ptr(), nb,
                                         scratch.data_ptr(), sb, B, H, W, 16, 16, 16, 0, stream)
        assert rc == 0, lib.curl_last_error()

    def two():
        cnt[0] += 1
        k = cnt[0] & 1
        rc = lib.curl_layer_fwd_f32(imgs[k].data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), out.data_ptr(),
                                    reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16, 16, 16, 0, stream)
        assert rc == 0, lib.curl_last_error()
        rc = lib.curl_loss_terms_f32(out.data_ptr(), tgts[k].data_ptr(), mask.data_ptr(), 1, sums.data_ptr(), Lp.data_ptr(), Lt.data_ptr(),
                                     scratch.data_ptr(), sb, B, H, W, stream)
        assert rc == 0, lib.curl_last_error()

    n = 400 if B * H * W < 16e6 else 100
    for _ in range(150):
        two()
    t = {"one pass": [], "two calls": []}
    for r in range(9):
        for name, fn in ((("one pass", fused), ("two calls", two)) if r % 2 == 0 else (("two calls", two), ("one pass", fused))):
            for _ in range(20):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t[name].append(e0.elapsed_time(e1) / n * 1e3)
    a, b = statistics.median(t["one pass"]), statistics.median(t["two calls"])
    d = sorted((x - y) / y * 100 for x, y in zip(t["one pass"], t["two calls"]))
    print(f"{B:3d}x{H:4d}x{W:4d}  one pass {a:8.1f} us   two calls {b:8.1f} us   per-round difference: median {d[len(d) // 2]:+.1f} %  "
          f"({d[len(d) // 4]:+.1f} .. {d[3 * len(d) // 4]:+.1f})", flush=True)
