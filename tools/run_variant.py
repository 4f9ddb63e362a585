"""Run one experiment build of the library (tools/variants.py) on the bench workload -- the process rocprofv3 wraps
when a counter pass of a VARIANT is wanted (kernel names are the same in every build, so one process per build).

    python tools/run_variant.py <variant|path.so> [layer|lab_stage] [launches]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab import bind  # noqa: E402
import variants  # noqa: E402


def main():
    v = sys.argv[1]
    what = sys.argv[2] if len(sys.argv) > 2 else "layer"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    flags = int(os.environ.get("FLAGS", "0"), 0)
    lib = bind(v if v.endswith(".so") else variants.path(v))
    B, H, W = 32, 1000, 1500
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
    out = torch.empty_like(imgs[0])
    mask = torch.ones(B, 1, H, W, dtype=torch.uint8, device=dev)
    L, R, Hk = (torch.randn(B, k, device=dev) * 0.1 for k in (48, 48, 64))
    reg = torch.empty(B, device=dev)
    poly = torch.randn(B, 3, 3, 126, device=dev) * 0.2
    nb = lib.curl_workspace_bytes(B, 160)
    ws = torch.empty(nb // 4, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(i):
        img = imgs[i & 1]
        if what == "layer":
            rc = lib.curl_layer_fwd_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), R.data_ptr(), Hk.data_ptr(),
                                        out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nb, B, H, W, 16, 16, 16, flags, stream)
        elif what == "trispace":
            rc = lib.curl_trispace_fwd_f32(img.data_ptr(), poly.data_ptr(), out.data_ptr(), B, H, W, 126, flags, stream)
        else:
            rc = lib.curl_lab_stage_f32(img.data_ptr(), mask.data_ptr(), 1, L.data_ptr(), out.data_ptr(), reg.data_ptr(),
                                        ws.data_ptr(), nb, B, H, W, 16, flags, stream)
        assert rc == 0, rc

    for i in range(200):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(v)} {what} flags={flags:#x}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call (incl. knot prep)")


if __name__ == "__main__":
    main()
