"""Board power and clock while each kernel runs back to back (is the sustained layer time set by the
power the chip can hold?).  Reads the amdgpu hwmon files (power1_average / power1_input, power1_cap, freq1_input) of the
card in a sampler thread while the main thread keeps one workload running for a few seconds.

    python tools/power_sample.py [seconds per workload]
"""
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import _lib, ops  # noqa: E402


def hwmon_files():
    out = {}
    pr = torch.cuda.get_device_properties(0)  # sysfs lists every card of the host: match this process's by PCI address
    want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if want not in os.path.realpath(os.path.join(d, "..", "..")):
            continue
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input",
                     "temp2_input"):
            p = os.path.join(d, name)
            if os.path.exists(p) and name not in out:
                out[name] = p
    return out


def read(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    files = hwmon_files()
    print("hwmon files:", json.dumps(files))
    B, H, W = 32, 1000, 1500
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    out = torch.empty_like(imgs[0])
    ones = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
    L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
    poly = torch.randn(B, 3, 3, 126, device=dev) * 0.2
    _lib.load()
    work = {
        "idle": None,
        "rgb_only": lambda i: ops.adjust_rgb(imgs[i & 1], R),
        "lab_stage": lambda i: ops.lab_stage(imgs[i & 1], ones, L, out=out),
        "layer": lambda i: ops.curl_layer_forward(imgs[i & 1], ones, L, R, Hk, out=out),
        "layer_nomem": lambda i: ops.curl_layer_forward(imgs[i & 1], ones, L, R, Hk, out=out, flags=_lib.F_DIAG_NO_MEM),
        "trispace": lambda i: ops.trispace_forward(imgs[i & 1], poly),
        # round 3
        "hsv_stage": lambda i: ops.hsv_stage(imgs[i & 1], ones, Hk, out=out),
        "layer_bright_pixels": lambda i: ops.curl_layer_forward(bright[i & 1], ones, L, R, Hk, out=out),
        "layer_bwd": lambda i: ops.curl_layer_backward(imgs[i & 1], ones, L, R, Hk, imgs[1 - (i & 1)]),
        "layer_exact_order": lambda i: ops.curl_layer_forward(imgs[i & 1], ones, L, R, Hk, out=out, flags=_lib.F_EXACT_ORDER),
    }
    bright = [0.2 + 0.8 * t for t in imgs]  # no dark values: every wave skips the four linear branches
    res = {}
    for name, fn in work.items():
        samples, stop = [], [False]

        def sampler():
            while not stop[0]:
                samples.append({k: read(p) for k, p in files.items()})
                time.sleep(0.05)
        th = threading.Thread(target=sampler)
        t0 = time.time()
        n = 0
        if fn is None:
            th.start()
            time.sleep(min(secs, 2.0))
        else:
            for i in range(200):
                fn(i)
            torch.cuda.synchronize()
            th.start()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            while time.time() - t0 < secs:
                for i in range(100):
                    fn(n + i)
                n += 100
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
        stop[0] = True
        th.join()
        pw = [s.get("power1_average") or s.get("power1_input") for s in samples]
        pw = [p for p in pw if p]
        fq = [s.get("freq1_input") for s in samples if s.get("freq1_input")]
        r = {"samples": len(samples), "power_W_mean": round(sum(pw) / len(pw) / 1e6, 1) if pw else None,
             "power_W_max": round(max(pw) / 1e6, 1) if pw else None,
             "power_cap_W": (samples[0].get("power1_cap") or 0) / 1e6 if samples else None,
             "sclk_MHz_mean": round(sum(fq) / len(fq) / 1e6, 0) if fq else None}
        if fn is not None:
            r["us_per_call"] = round(e0.elapsed_time(e1) / n * 1e3, 1)
        res[name] = r
        print(name, json.dumps(r), flush=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "power_sample.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
