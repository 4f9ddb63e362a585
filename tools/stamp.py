"""In-kernel clock and per-SIMD accounting of the streaming kernels.

Uses the `stamp` experiment build (tools/variants.py: -DCURL_DIAG_STAMP): every wave records s_memtime (shader
clock) and s_memrealtime (100 MHz) at entry, when its loads have landed and after its stores are issued, and the
SIMD it ran on.  From one launch taken after the clock has settled:

    clock        = d(s_memtime) / d(s_memrealtime) x 100 MHz        (median over waves)
    kernel span  = max(end) - min(start) of the 100 MHz stamps
    cycles/wave  = span x clock / waves per SIMD                    (SIMD time one wave costs)
    concurrency  = sum of compute spans per SIMD / span             (waves in their arithmetic phase at once)

    python tools/stamp.py [layer layer_nomem lab_stage rgb_only ...]
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab import bind  # noqa: E402
from curl_amd import _lib  # noqa: E402
import variants  # noqa: E402

B, H, W = int(os.environ.get("B", 32)), int(os.environ.get("H", 1000)), int(os.environ.get("W", 1500))
SETTLE = int(os.environ.get("SETTLE", 400))


def main():
    names = sys.argv[1:] or ["layer", "layer_nomem", "lab_stage", "lab_stage_nomem", "rgb_only"]
    lib = bind(variants.path(os.environ.get("VARIANT", "stamp")))
    lib.curl_diag_set_stamps.restype = ctypes.c_int
    lib.curl_diag_set_stamps.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    out = torch.empty_like(imgs[0])
    mask = torch.ones(B, 1, H, W, dtype=torch.uint8, device=dev)
    L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
    reg = torch.empty(B, device=dev)
    poly = torch.randn(B, 3, 3, 126, device=dev) * 0.2
    nb = lib.curl_workspace_bytes(B, 160)
    ws = torch.empty(nb // 4, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    n_waves_max = B * max(((H * W // 4 + 255) // 256) * 4, H * 2 * 4)  # row-tiled ops: <= 2 blocks of <= 4 waves per row
    stamps = torch.zeros(n_waves_max * 8, dtype=torch.int64, device=dev)
    assert lib.curl_diag_set_stamps(stamps.data_ptr()) == 0
    cnt = [0]

    def run(name):
        cnt[0] += 1
        img = imgs[cnt[0] & 1]
        flags = _lib.F_DIAG_NO_MEM if name.endswith("_nomem") else 0
        base = name.replace("_nomem", "")
        if base == "layer":
            rc = lib.curl_layer_fwd_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(),
                                        out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16, 16, 16, flags,
                                        stream)
        elif base == "lab_stage":
            rc = lib.curl_lab_stage_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), out.data_ptr(), reg.data_ptr(),
                                        ws.data_ptr(), nb, B, H, W, 16, flags, stream)
        elif base == "trispace":
            rc = lib.curl_trispace_fwd_f32(img.data_ptr(), poly.data_ptr(), out.data_ptr(), B, H, W, 126, flags, stream)
        elif base == "rgb_only":
            rc = lib.curl_adjust_rgb_f32(img.data_ptr(), R.data_ptr(), out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb,
                                         B, H, W, 16, flags, stream)
        else:
            rc = getattr(lib, f"curl_{base}_f32")(img.data_ptr(), out.data_ptr(), B, H, W, flags, stream)
        assert rc == 0, (name, rc, lib.curl_last_error())

    results = {}
    for name in names:
        for _ in range(SETTLE):
            run(name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run(name)
        e1.record()
        torch.cuda.synchronize()
        us_event = e0.elapsed_time(e1) / 50 * 1e3
        stamps.zero_()
        torch.cuda.synchronize()
        for _ in range(20):  # keep the clock where it was; the LAST launch's stamps are the ones read
            run(name)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(-1, 8)
        s = s[s[:, 0] != 0]
        t0, t1, t2, r0, r1, r2 = (s[:, i].astype(np.float64) for i in range(6))
        hw, xcc = s[:, 6], s[:, 7] & 0xF
        life, comp, wait = t2 - t0, t2 - t1, t1 - t0
        ok = (r2 - r0) >= 200
        clock = np.median(life[ok] / (r2 - r0)[ok]) * 0.1  # GHz
        span_us = (r2.max() - r0.min()) / 100.0
        simd_key = (xcc.astype(np.int64) << 16) | ((hw >> 4) & 0x3) | (((hw >> 8) & 0xFF) << 2)  # simd | cu,sh,se
        keys, inv, counts = np.unique(simd_key, return_inverse=True, return_counts=True)
        comp_per_simd = np.bincount(inv, weights=comp)
        life_per_simd = np.bincount(inv, weights=life)
        span_cyc = span_us * 1e3 * clock
        res = {
            "event_us_per_launch_incl_prep": round(us_event, 2),
            "waves": int(len(s)), "simds_seen": int(len(keys)),
            "waves_per_simd_mean": round(float(counts.mean()), 1), "waves_per_simd_max": int(counts.max()),
            "in_kernel_clock_GHz": round(float(clock), 3),
            "kernel_span_us": round(float(span_us), 2),
            "simd_cycles_per_wave": round(float(span_cyc / counts.mean()), 1),
            "wave_life_cycles_mean": round(float(life.mean()), 0),
            "wave_compute_cycles_mean": round(float(comp.mean()), 0),
            "wave_loadwait_cycles_mean": round(float(wait.mean()), 0),
            "compute_concurrency_per_simd": round(float((comp_per_simd / span_cyc).mean()), 2),
            "resident_waves_per_simd": round(float((life_per_simd / span_cyc).mean()), 2),
        }
        results[name] = res
        print(name, json.dumps(res), flush=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "stamp.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
