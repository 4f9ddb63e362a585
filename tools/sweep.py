"""Launch-geometry sweep for the streaming kernels (exploration tool, writes gpurun_out/sweep.json).
Interleaved rounds in one process (guide rule 24); times with events on the launch stream."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curl_amd import _lib, ops  # noqa: E402


def main():
    B, H, W = int(os.environ.get("B", 32)), 1000, 1500
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    img = imgs[0]
    counter = [0]
    out = torch.empty_like(img)
    L = torch.randn(B, 48, device=dev) * 0.1
    R = torch.randn(B, 48, device=dev) * 0.1
    Hk = torch.randn(B, 64, device=dev) * 0.1
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import disk_mask
    maskb = disk_mask(B, H, W, dev)
    lib = _lib.load()
    npx = B * H * W

    def run(name, flags):
        counter[0] += 1
        img = imgs[counter[0] & 1]  # rotate inputs: nothing is re-read from the 256 MiB MALL
        if name == "layer":
            ops.curl_layer_forward(img, None, L, R, Hk, flags=flags, out=out)
        elif name == "layer_mask_u8":
            ops.curl_layer_forward(img, maskb, L, R, Hk, flags=flags, out=out)
        elif name == "lab_stage_mask_u8":
            ops.lab_stage(img, maskb, L, flags=flags, out=out)
        elif name == "lab_stage":
            ops.lab_stage(img, None, L, flags=flags, out=out)
        elif name == "adjust_rgb":
            ops.adjust_rgb(img, R, flags=flags)
        elif name == "adjust_rgb_exact":
            ops.adjust_rgb(img, R, flags=flags | 1)
        elif name == "rgb2lab":
            ops.rgb2lab(img, flags=flags)
        elif name == "rgb2hsv":
            ops.rgb2hsv(img, flags=flags)
        elif name == "hsv2rgb":
            ops.hsv2rgb(img, flags=flags)
        elif name == "lab2rgb":
            ops.lab2rgb(img, flags=flags)
        elif name == "copy":
            out.copy_(img)

    variants = []
    for u in (1, 2, 4):
        for iters in (0,):
            for nt in (0, _lib.F_TUNE_NO_NT):
                for xcd in (0,):
                    variants.append((u, nt, xcd))
    names = os.environ.get("WORKLOADS", "layer_mask_u8,lab_stage_mask_u8,adjust_rgb,layer").split(",")
    results = {}
    rounds = 5
    iters = 10
    for name in names + ["copy"]:
        vs = variants if name != "copy" else [(0, 0, 0)]
        times = {v: [] for v in vs}
        for v in vs:  # warm
            run(name, (v[0] << 8) | v[1] | v[2])
        torch.cuda.synchronize()
        for r in range(rounds):
            for v in vs:
                flags = (v[0] << 8) | v[1] | v[2]
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run(name, flags)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / iters)
        res = []
        for v in vs:
            t = sorted(times[v])
            med = t[len(t) // 2]
            res.append({"unroll": v[0], "nt": hex(v[1]), "xcd": int(bool(v[2])), "ms_med": med, "ms_min": t[0],
                        "gpix_s": npx / med / 1e6, "GBps_24": npx * 24 / med / 1e6})
        res.sort(key=lambda r: r["ms_med"])
        results[name] = res
        print(name)
        for r in res[:6]:
            print("   ", r)
        sys.stdout.flush()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(results, open("gpurun_out/sweep.json", "w"), indent=1)


if __name__ == "__main__":
    main()
