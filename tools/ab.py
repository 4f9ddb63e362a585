"""A/B timing of two builds of libcurlhip.so in ONE process on ONE GPU (box-to-box clock differences are +-2 %,
more than most single optimisations): alternating rounds, full and arithmetic-only (CURL_F_DIAG_NO_MEM).

    git stash / checkout the baseline, python -m curl_amd.build --force, cp curl_amd/lib/libcurlhip.so /tmp/base.so ...
    python tools/ab.py curl_amd/lib/libcurlhip_base.so curl_amd/lib/libcurlhip.so [layer|lab_stage|trispace]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curl_amd import _lib  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def main():
    pa, pb = sys.argv[1], sys.argv[2]
    what = sys.argv[3] if len(sys.argv) > 3 else "layer"
    B, H, W = int(os.environ.get("B", 32)), int(os.environ.get("H", 1000)), int(os.environ.get("W", 1500))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lo = float(os.environ.get("IMG_LO", 0.0))  # IMG_LO=0.2: no dark pixels (a photograph's mid-tones): pixels = lo + (1 - lo) U
    imgs = [lo + (1.0 - lo) * torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    out = torch.empty_like(imgs[0])
    mask = torch.ones(B, 1, H, W, dtype=torch.uint8, device=dev)
    if os.environ.get("MASK") == "disk":  # bench.py's disk: ~70 % coverage, whole wavefronts inside and outside
        yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        r2 = ((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2
        mask = (r2 <= float(os.environ.get("DISK_R2", 0.9))).to(torch.uint8).expand(B, 1, H, W).contiguous()
        print("mask coverage", float(mask.float().mean()))
    L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
    poly = torch.randn(B, 3, 3, 126, device=dev) * 0.2
    reg = torch.empty(B, device=dev)
    libs = {"A": bind(pa), "B": bind(pb)}
    nb = libs["A"].curl_workspace_bytes(B, 160)
    ws = torch.empty(nb // 4, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cnt = [0]
    gout = torch.rand(B, 3, H, W, device=dev) if what == "layer_bwd" else None
    gin = torch.empty_like(out)
    no_gin = bool(os.environ.get("NO_GIN"))  # layer_bwd: grad_img = NULL, the knots-only kernel
    gL, gR, gH = torch.empty_like(L), torch.empty_like(R), torch.empty_like(Hk)
    sb = libs["A"].curl_layer_bwd_scratch_bytes(B, H, W)
    scratch = torch.empty(max(4, sb) // 4, device=dev)

    loss_sums = torch.empty(B, 5, dtype=torch.float64, device=dev)
    loss_L = [torch.empty(B, 1, H, W, device=dev) for _ in range(2)] if what == "loss_fwd" else [out, out]
    loss_nb = libs["A"].curl_loss_terms_scratch_bytes(B, H, W)
    loss_scratch = torch.empty(max(4, loss_nb) // 4, device=dev)

    w4 = torch.tensor([1.0, 1e-3, 1.0, 1.0], device=dev)
    gLp = torch.rand(B, 1, H, W, device=dev)
    psnr_nb = libs["A"].curl_psnr_scratch_bytes(B, H, W)
    psnr_scratch = torch.empty(max(4, psnr_nb) // 4, device=dev)
    bytes_ = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev)

    def run(lib, flags):
        cnt[0] += 1
        img = imgs[cnt[0] & 1]
        if what == "layer":
            rc = lib.curl_layer_fwd_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(),
                                        out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16, 16, 16, flags, stream)
        elif what == "layer_bwd":
            rc = lib.curl_layer_bwd_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(),
                                        gout.data_ptr(), 0, 0 if no_gin else gin.data_ptr(), gL.data_ptr(), gR.data_ptr(), gH.data_ptr(),
                                        ws.data_ptr(), nb, scratch.data_ptr(), sb, B, H, W, 16, 16, 16, flags & 0x400000, stream)
        elif what == "loss_fwd":
            rc = lib.curl_loss_terms_f32(img.data_ptr(), imgs[1 - (cnt[0] & 1)].data_ptr(), mask.data_ptr(), 1, loss_sums.data_ptr(),
                                         loss_L[0].data_ptr(), loss_L[1].data_ptr(), loss_scratch.data_ptr(), loss_nb, B, H, W, stream)
        elif what == "loss_bwd":
            rc = lib.curl_loss_terms_bwd_f32(img.data_ptr(), imgs[1 - (cnt[0] & 1)].data_ptr(), mask.data_ptr(), 1, w4.data_ptr(),
                                             gLp.data_ptr(), out.data_ptr(), B, H, W, stream)
        elif what == "psnr":
            rc = lib.curl_psnr_f32(img.data_ptr(), imgs[1 - (cnt[0] & 1)].data_ptr(), mask.data_ptr(), 1, reg.data_ptr(),
                                   psnr_scratch.data_ptr(), psnr_nb, B, H, W, 1.0, stream)
        elif what == "to_u8":
            rc = lib.curl_f32chw_to_u8hwc(img.data_ptr(), bytes_.data_ptr(), B, H, W, stream)
        elif what == "from_u8":
            rc = lib.curl_u8hwc_to_f32chw(bytes_.data_ptr(), out.data_ptr(), B, H, W, 3, stream)
        elif what == "rgb2lab":
            rc = lib.curl_rgb2lab_f32(img.data_ptr(), out.data_ptr(), B, H, W, flags, stream)
        elif what == "adjust_rgb":
            rc = lib.curl_adjust_rgb_f32(img.data_ptr(), R.data_ptr(), out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16,
                                         flags, stream)
        elif what == "hsv_stage":
            rc = lib.curl_hsv_stage_f32(img.data_ptr(), mask.data_ptr(), 1, Hk.data_ptr(), out.data_ptr(), reg.data_ptr(),
                                        ws.data_ptr(), nb, B, H, W, 16, flags, stream)
        elif what == "lab_stage":
            rc = lib.curl_lab_stage_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), out.data_ptr(), reg.data_ptr(),
                                        ws.data_ptr(), nb, B, H, W, 16, flags, stream)
        else:
            rc = lib.curl_trispace_fwd_f32(img.data_ptr(), poly.data_ptr(), out.data_ptr(), B, H, W, 126, flags, stream)
        assert rc == 0, rc

    extra = {"A": int(os.environ.get("FLAGS_A", "0"), 0), "B": int(os.environ.get("FLAGS_B", "0"), 0)}  # e.g. 0x200 = U=2
    full_only = what in ("layer_bwd", "loss_fwd", "loss_bwd", "psnr", "to_u8", "from_u8") or os.environ.get("FULL_ONLY")  # FULL_ONLY=1: skip the arithmetic-only (no-memory) legs
    variants = [(k, d) for d in ((0,) if full_only else (0, _lib.F_DIAG_NO_MEM)) for k in ("A", "B")]
    LAUNCHES = int(os.environ.get("LAUNCHES", 100))  # per timed window
    times = {v: [] for v in variants}
    for _ in range(150):  # clock settle
        run(libs["A"], 0)
    torch.cuda.synchronize()
    for r in range(int(os.environ.get('ROUNDS', 11))):
        for v in (variants if r % 2 == 0 else variants[::-1]):
            for _ in range(20):
                run(libs[v[0]], v[1] | extra[v[0]])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(LAUNCHES):
                run(libs[v[0]], v[1] | extra[v[0]])
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / LAUNCHES)
    # paired view: A and B of one round run within ~50 ms of each other, so the board's slow power / thermal drift (which
    # moves a power-capped kernel by +-2 % over a session) cancels in the per-round difference
    for d in sorted({v[1] for v in variants}):
        diffs = sorted((b - a) / a * 100 for a, b in zip(times[("A", d)], times[("B", d)]))
        print(f"{what:10s} B vs A {'VALU-only' if d else 'full     '} per-round difference: median {diffs[len(diffs)//2]:+.2f} %  "
              f"quartiles {diffs[len(diffs)//4]:+.2f} .. {diffs[3*len(diffs)//4]:+.2f} %  ({len(diffs)} rounds)")
    for v in variants:
        t = sorted(times[v])
        print(f"{what:10s} {v[0]} ({os.path.basename(pa if v[0] == 'A' else pb)}) {'VALU-only' if v[1] else 'full     '} "
              f"median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us")


if __name__ == "__main__":
    main()
