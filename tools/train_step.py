"""BASELINE configs[4] in synthetic form: a data-parallel train step of the curve model -- encoder fwd/bwd on
PyTorch-ROCm (MIOpen/rocBLAS), fused HIP curve layer fwd/bwd -- 32 crops of 256x256 per GPU (main.py:88,
data.py:86), Adam (main.py:236), DDP over RCCL when launched with torchrun.  Prints one JSON line with the step
time and the share of the HIP curve kernels.

    python tools/train_step.py [--steps 30] [--size 256] [--batch 32] [--width 1.0]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_step.py
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curl_amd import model, ops  # noqa: E402


def run(args, dev, rank=0, world=1, local=0, dist=None):
    """One rank's share of the measurement; `dist` = an initialised torch.distributed module (RCCL) or None.
    Returns the result dict (every rank computes it; rank 0 reports).  bench.py calls this for `end_to_end.train_step`."""
    torch.manual_seed(rank)
    net = model.GCURLNet(backbone=model.CurveEncoder(160, width=args.width)).to(dev).train()
    if dist is not None and world > 1:
        net = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local])  # main.py:225
    opt = torch.optim.Adam(net.parameters(), lr=5e-7, betas=(0.5, 0.999))  # main.py:236
    B, S = args.batch, args.size
    img = torch.rand(B, 3, S, S, device=dev)
    gt = torch.rand(B, 3, S, S, device=dev)
    mask = torch.rand(B, 1, S, S, device=dev) > 0.1

    def step():
        out, reg = net(img, mask)
        loss = ((out - gt).abs() * mask).sum() / (3 * mask.sum()) + 1e-6 * reg.mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    # the curve layer alone, fwd + bwd, same shapes
    knots = torch.randn(B, 160, device=dev) * 0.1
    L, R, Hk = knots[:, :48].contiguous(), knots[:, 48:96].contiguous(), knots[:, 96:].contiguous()
    g = torch.rand(B, 3, S, S, device=dev)
    ws = ops.curl_layer_forward(img, mask, L, R, Hk, return_workspace=True)[2]  # what the autograd node keeps for its backward
    for _ in range(5):
        ops.curl_layer_forward(img, mask, L, R, Hk)
        ops.curl_layer_backward(img, mask, L, R, Hk, g, None, need_grad_img=False, workspace=ws)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(20):
        ops.curl_layer_forward(img, mask, L, R, Hk)
    e[1].record()
    for _ in range(20):
        ops.curl_layer_backward(img, mask, L, R, Hk, g, None, need_grad_img=False, workspace=ws)
    e[2].record()
    torch.cuda.synchronize()
    fwd_ms, bwd_ms = e[0].elapsed_time(e[1]) / 20, e[1].elapsed_time(e[2]) / 20
    return {"metric": "train step (synthetic, curve model)", "n_gpus": world, "batch_per_gpu": B, "crop": S,
            "ms_per_step": dt * 1e3, "images_per_s": world * B / dt, "mpix_per_s": world * B * S * S / dt / 1e6,
            "curve_layer_fwd_ms": fwd_ms, "curve_layer_bwd_ms": bwd_ms,
            "curve_layer_share_of_step": (fwd_ms + bwd_ms) / (dt * 1e3), "loss": float(loss.detach()),
            "model": "GCURLNet: efficientnetv2_rw_s architecture (random init, fp32, stock PyTorch-ROCm convolutions) + fused "
                     "HIP CURLLayer forward / backward; masked L1 + 1e-6 reg; Adam (main.py:236)",
            "steps": args.steps, "warmup": args.warmup}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--width", type=float, default=1.0)
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    res = run(args, dev, rank, world, local, dist)
    if rank == 0:
        print(json.dumps(res))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
