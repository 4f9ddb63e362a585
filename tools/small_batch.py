"""Small launches, where the LAUNCH is the cost: CURLLayer forward / backward at one 1500x1000 frame
(infer.py:44 processes ONE image), two frames, and the training crop batch 32 x 256x256 (main.py:88, data.py:86).

Four clocks per shape, because a 10-20 us step is as much host as device:
  window_us   back-to-back calls through the Python surface (curl_amd.ops), HIP events around the window / calls
  host_us     wall time the host needs to ENQUEUE one call (no sync inside the window): if window_us ~ host_us the figure is
              host-bound and says nothing about the kernels
  cabi_us     the same window with the C ABI called directly on preallocated buffers (no tensor allocation, no checks)
  latency_us  one call between two events after a device sync (what a caller issuing ONE call sees)

    python tools/small_batch.py [--flags-fwd N] [--flags-bwd N]
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--flags-fwd", type=lambda s: int(s, 0), default=0)
ap.add_argument("--flags-bwd", type=lambda s: int(s, 0), default=0)
ap.add_argument("--n", type=int, default=400)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()


def window(fn, n):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, t_host / n * 1e6


def latency(fn, n=40):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


print(f"{'shape':>16s} {'pass':>5s} {'window_us':>10s} {'host_us':>8s} {'cabi_us':>8s} {'cabi_host':>9s} {'latency_us':>10s}")
for (B, H, W) in ((1, 1000, 1500), (2, 1000, 1500), (4, 1000, 1500), (32, 256, 256), (32, 1000, 1500)):
    torch.manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    gout = torch.rand(B, 3, H, W, device=dev)
    mask = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
    m8 = mask.view(torch.uint8)
    L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
    out = torch.empty_like(imgs[0])
    reg = torch.empty(B, device=dev)
    nbytes = lib.curl_workspace_bytes(B, 160)
    ws = torch.empty(nbytes // 4, device=dev)
    sb = lib.curl_layer_bwd_scratch_bytes(B, H, W)
    scratch = torch.empty(sb // 4, device=dev)
    gimg = torch.empty_like(imgs[0])
    gL, gR, gH = torch.empty_like(L), torch.empty_like(R), torch.empty_like(Hk)
    stream = torch.cuda.current_stream().cuda_stream
    cnt = [0]

    def img():
        cnt[0] += 1
        return imgs[cnt[0] & 1]

    def fwd_py():
        return ops.curl_layer_forward(img(), mask, L, R, Hk, flags=args.flags_fwd, out=out)

    def fwd_c():
        rc = lib.curl_layer_fwd_f32(img().data_ptr(), m8.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), out.data_ptr(),
                                    reg.data_ptr(), ws.data_ptr(), nbytes, B, H, W, 16, 16, 16, args.flags_fwd, stream)
        assert rc == 0, lib.curl_last_error()

    _, _, ws_py = ops.curl_layer_forward(imgs[0], mask, L, R, Hk, return_workspace=True)

    def bwd_py():
        return ops.curl_layer_backward(img(), mask, L, R, Hk, gout, workspace=ws_py, flags=args.flags_bwd)

    fwd_c()

    def bwd_c():
        rc = lib.curl_layer_bwd_f32(img().data_ptr(), m8.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), gout.data_ptr(), 0,
                                    gimg.data_ptr(), gL.data_ptr(), gR.data_ptr(), gH.data_ptr(), ws.data_ptr(), nbytes,
                                    scratch.data_ptr(), sb, B, H, W, 16, 16, 16, _lib.F_WS_READY | args.flags_bwd, stream)
        assert rc == 0, lib.curl_last_error()

    def bwdk_py():  # knot gradients only: what autograd asks for when the image needs no gradient (the training step)
        return ops.curl_layer_backward(img(), mask, L, R, Hk, gout, workspace=ws_py, flags=args.flags_bwd, need_grad_img=False)

    def bwdk_c():
        rc = lib.curl_layer_bwd_f32(img().data_ptr(), m8.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), gout.data_ptr(), 0,
                                    0, gL.data_ptr(), gR.data_ptr(), gH.data_ptr(), ws.data_ptr(), nbytes,
                                    scratch.data_ptr(), sb, B, H, W, 16, 16, 16, _lib.F_WS_READY | args.flags_bwd, stream)
        assert rc == 0, lib.curl_last_error()

    # the live model's per-pixel path between bytes (infer.py:35-47: ONE image per call): curl_trispace_fwd_u8hwc
    u8s = [(im * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous() for im in imgs]
    coeffs = torch.randn(B, 3, 3, 126, device=dev) * 0.2
    out8 = torch.empty_like(u8s[0])

    def tri_py():
        cnt[0] += 1
        return ops.trispace_forward_u8hwc(u8s[cnt[0] & 1], coeffs)

    def tri_c():
        cnt[0] += 1
        rc = lib.curl_trispace_fwd_u8hwc(u8s[cnt[0] & 1].data_ptr(), coeffs.data_ptr(), 0, out8.data_ptr(), B, H, W, 126, 0, stream)
        assert rc == 0, lib.curl_last_error()

    n = args.n if B * H * W < 16e6 else 100
    for name, py, c in (("fwd", fwd_py, fwd_c), ("bwd", bwd_py, bwd_c), ("bwdk", bwdk_py, bwdk_c), ("tri8", tri_py, tri_c)):
        w, h = window(py, n)
        wc, hc = window(c, n)
        lat = latency(c)
        print(f"{B:3d}x{H:4d}x{W:4d}   {name:>5s} {w:10.1f} {h:8.1f} {wc:8.1f} {hc:9.1f} {lat:10.1f}", flush=True)
