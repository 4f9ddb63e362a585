"""A/B of the polynomial path's backward (curl_trispace_bwd_f32, d loss / d coeffs) between experiment builds
(tools/variants.py), interleaved rounds, at the training crop batch and two full-frame shapes; also checks the builds
against each other (different accumulation order: same sums to float32 rounding).

    python tools/bwd_ab.py base bwd_plain [bwd_s32 ...]
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab import bind  # noqa: E402
import variants  # noqa: E402

SHAPES = [(32, 256, 256), (32, 512, 512), (8, 1000, 1500), (4, 300, 450)]
if os.environ.get("SHAPES") == "crop":  # one shape per process: rocprofv3's per-kernel averages then belong to it
    SHAPES = SHAPES[:1]


def main():
    names = sys.argv[1:] or ["base", "bwd_plain"]
    libs = {n: bind(variants.path(n)) for n in names}
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for (B, H, W) in SHAPES:
        torch.manual_seed(B + H)
        img = torch.rand(B, 3, H, W, device=dev)
        gout = torch.randn(B, 3, H, W, device=dev)
        c = torch.randn(B, 3, 3, 126, device=dev) * 0.2
        res, times = {}, {n: [] for n in names}

        def run(n, out, scratch, nbytes):
            rc = libs[n].curl_trispace_bwd_f32(img.data_ptr(), c.data_ptr(), gout.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                               nbytes, B, H, W, 126, 0, stream)
            assert rc == 0, (n, rc)

        bufs = {}
        for n in names:
            nbytes = libs[n].curl_trispace_bwd_scratch_bytes(B, H, W, 126)
            bufs[n] = (torch.empty_like(c), torch.empty(nbytes // 4, device=dev), nbytes)
            for _ in range(5):
                run(n, *bufs[n])
            torch.cuda.synchronize()
            res[n] = bufs[n][0].clone()
        for _ in range(int(os.environ.get("ROUNDS", 5))):
            for n in names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(n, *bufs[n])
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / 20 * 1e3)
        ref = res[names[-1]].double()
        for n in names:
            d = float((res[n].double() - ref).abs().max() / ref.abs().max())
            print(f"{B}x{H}x{W}  {n:12s} median {statistics.median(times[n]):8.1f} us  min {min(times[n]):8.1f} us   "
                  f"max rel diff vs {names[-1]}: {d:.2e}", flush=True)


if __name__ == "__main__":
    main()
