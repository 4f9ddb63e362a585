"""Data-parallel training driver in the shape of the reference's main.py:212-340 (BASELINE configs[4]).

    python -m curl_amd.train --num_epoch 4 --batch_size 32                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m curl_amd.train --parallel_mode ddp --num_epoch 4 --batch_size 32   # one process per GPU, RCCL

What is kept from main.py: one process per GPU under DDP with SyncBatchNorm (main.py:98-124,222-225, model.py:457),
`DistributedSampler` + `set_epoch` (main.py:213-214,262), the model/criterion/optimiser/scheduler of main.py:221-239
(TriSpaceRegNet(polynomial_order=4, spatial=True), CURLLoss(ssim_window_size=5), Adam(lr=5e-7, betas=(0.5, 0.999)),
OneCycleLR(max_lr=1e-4, total_steps=num_epoch)), the step of main.py:283-289, the per-epoch loss gather of
main.py:301-305, validation every `--valid_every` epochs with masked PSNR, and checkpoints with the reference's keys
(main.py:332-338) that `--checkpoint_filepath` resumes from (main.py:241-250).
Encoder forward/backward is stock PyTorch-ROCm; every per-pixel piece (polynomial or curve layer forward and
backward, the loss's colour terms, PSNR) is the HIP library.  Gradients of the per-image coefficients are local
to the rank that owns the image; the only collective is DDP's bucketed all-reduce of the encoder's gradients.

Data: `--training_img_dirpath <dir>` reads the reference's folder layout through curl_amd.data (data.py without
torchvision); the Adobe-5k-DPE folders are not available offline, so the default `synthetic` draws seeded random crops
whose ground truth is a fixed smooth retouch of the input.
"""
import argparse
import contextlib
import json
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from . import data, evaluate, model


class SyntheticPairs(Dataset):
    """Items shaped like data.py:160-207: {'input_img','output_img','mask','name'}; float32 CHW in [0,1], mask
    [1,H,W] bool.  The target is a fixed, learnable retouch (gamma + saturation + a warm tint) of the input."""

    def __init__(self, n, crop, seed):
        self.n, self.crop, self.seed = n, crop, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        S = self.crop
        low = torch.rand(3, S // 8, S // 8, generator=g)
        x = nn.functional.interpolate(low[None], size=(S, S), mode="bilinear", align_corners=False)[0]
        x = (x + 0.05 * torch.randn(3, S, S, generator=g)).clamp(0, 1)
        grey = x.mean(0, keepdim=True)
        y = (grey + 1.25 * (x - grey)).clamp(0, 1) ** 0.8
        y = (y * torch.tensor([1.05, 1.0, 0.93]).view(3, 1, 1)).clamp(0, 1)
        yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
        c = torch.rand(2, generator=g) * S
        mask = (((yy - c[0]) ** 2 + (xx - c[1]) ** 2) < (0.6 * S) ** 2)[None]
        return {"input_img": x, "output_img": y, "mask": mask, "name": f"synthetic_{i:05d}"}


def build_net(arch, width, sync_bn, foreground_masks=False):
    if arch == "trispace":  # main.py:221
        net = model.TriSpaceRegNet(polynomial_order=4, spatial=True, use_sync_bn=sync_bn,
                                   backbone=model.CurveEncoder(num_outputs=1, num_features=1024, width=width,
                                                               variant="efficientnetv2_rw_t"))  # model.py:456
    else:
        net = model.GCURLNet(backbone=model.CurveEncoder(160, width=width), foreground_masks=foreground_masks)
        if sync_bn:
            net = nn.SyncBatchNorm.convert_sync_batchnorm(net)
    return net


def forward_image(net, img, mask):
    out = net(img, mask)
    return out[0] if isinstance(out, tuple) else out  # the curve model also returns its regulariser


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train the enhancement model (shape of the reference's main.py)")
    ap.add_argument("--num_epoch", type=int, default=4)
    ap.add_argument("--valid_every", type=int, default=2)
    ap.add_argument("--checkpoint_filepath", type=str, default=None)
    ap.add_argument("--training_img_dirpath", type=str, default="synthetic")
    ap.add_argument("--inference_img_dirpath", type=str, default=None,
                    help="with --checkpoint_filepath: evaluate that checkpoint on <dir> (images_inference.txt) and dump "
                         "the outputs, main.py:147-193; no training")
    ap.add_argument("--batch_size", type=int, default=32, help="per process, as main.py:117 after its division")
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--parallel_mode", type=str, default=None, choices=["ddp"])
    ap.add_argument("--local_rank", type=int, default=0, help="accepted for torch.distributed.launch (main.py:91); "
                                                              "the LOCAL_RANK environment variable wins")
    ap.add_argument("--arch", choices=("trispace", "curl"), default="trispace")
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL over xGMI; gloo for rehearsals")
    ap.add_argument("--crop", type=int, default=256, help="data.py:86 crops 256x256 .. 'resize to 320' variants")
    ap.add_argument("--train_items", type=int, default=256)
    ap.add_argument("--valid_items", type=int, default=64)
    ap.add_argument("--width", type=float, default=1.0, help="encoder width multiplier")
    ap.add_argument("--log_dirpath", type=str, default=None, help="where checkpoints go (rank 0); none = no files")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--foreground_masks", action="store_true",
                    help="--arch curl: the masks have sizeable empty regions (data.py:186-190) -- masked-out wavefronts skip "
                         "their pixel loads (CURL_F_MASK_FIRST; same results)")
    ap.add_argument("--fused_forward", action="store_true",
                    help="--arch curl: the layer and CURLLoss' pointwise terms as ONE forward kernel (curl_layer_loss_fwd_f32); the "
                         "same loss and gradients as main.py:283-285's two calls")
    ap.add_argument("--save_images", action="store_true", help="dump validation outputs under --log_dirpath (evaluate.py:49-66)")
    ap.add_argument("--amp", choices=("off", "bf16"), default="off",
                    help="bf16: torch.autocast around the encoder (stock PyTorch-ROCm); the per-pixel HIP kernels and "
                         "the loss always run in float32 (their autograd nodes cast inputs back)")
    ap.add_argument("--channels_last", action=argparse.BooleanOptionalAction, default=True,
                    help="NHWC memory format for the encoder's convolutions (same float32 results; measured 136 -> 85 ms "
                         "per 32x256x256 step on MI355X, 74 ms with --amp bf16)")
    ap.add_argument("--miopen_benchmark", action=argparse.BooleanOptionalAction, default=False,
                    help="torch.backends.cudnn.benchmark: let MIOpen time its solvers per convolution shape")
    args = ap.parse_args(argv)
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    ddp = args.parallel_mode == "ddp" and world > 1
    n_dev = torch.cuda.device_count()
    device = torch.device("cuda", local % max(1, n_dev))
    torch.cuda.set_device(device)
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, **({"device_id": device} if args.backend == "nccl" else {}))
    torch.manual_seed(args.seed)  # same initial weights on every rank (DDP broadcasts rank 0's anyway)

    if args.checkpoint_filepath and args.inference_img_dirpath:            # main.py:147-193
        if args.parallel_mode is not None:
            raise ValueError("Inference not supported with DP or DDP. Do not pass --parallel_mode parameter.")
        from .convert_state import convert_state_dict
        root = os.path.join(args.inference_img_dirpath, "")
        dd = data.filter_data_dict(data.get_data_dict(root), data.get_data_ids(os.path.join(root, "images_inference.txt")))
        loader = DataLoader(data.Dataset(dd, normaliser=1, is_train=False, crop_h=args.crop, crop_w=args.crop),
                            batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers)
        net = build_net(args.arch, args.width, sync_bn=False, foreground_masks=args.foreground_masks)
        ckpt = torch.load(args.checkpoint_filepath, map_location="cpu")
        net.load_state_dict(convert_state_dict(ckpt["model_state_dict"]))  # DP/DDP "module." prefixes removed
        net = net.to(device).eval()
        log_dir = args.log_dirpath or "."
        os.makedirs(log_dir, exist_ok=True)
        ev = evaluate.Evaluator(model.CURLLoss().to(device), loader, "test", log_dir, local_rank=rank)
        loss, psnr, msssim = ev.evaluate(net, epoch=0, save_images=True)
        print(json.dumps({"mode": "inference", "arch": args.arch, "images": len(dd), "test_loss": loss, "test_psnr": psnr,
                          "test_msssim": msssim, "images_dir": os.path.join(log_dir, "test", "1")}))
        return

    if args.training_img_dirpath == "synthetic":
        train_set = SyntheticPairs(args.train_items, args.crop, seed=1)
        valid_set = SyntheticPairs(args.valid_items, args.crop, seed=2)
    else:  # main.py:196-210: <dir>/{*input*,*output*,*mask*}/<id>.<ext> + images_train.txt / images_valid.txt
        root = os.path.join(args.training_img_dirpath, "")
        data_dict = data.get_data_dict(root)
        train_ids = data.get_data_ids(os.path.join(root, "images_train.txt"))
        valid_ids = data.get_data_ids(os.path.join(root, "images_valid.txt"))
        train_set = data.Dataset(data.filter_data_dict(data_dict, train_ids), normaliser=1, is_train=True,
                                 crop_h=args.crop, crop_w=args.crop, seed=args.seed * 7919 + rank)
        valid_set = data.Dataset(data.filter_data_dict(data_dict, valid_ids), normaliser=1, is_train=False,
                                 crop_h=args.crop, crop_w=args.crop)
    train_sampler = DistributedSampler(train_set) if ddp else None           # main.py:213
    valid_sampler = DistributedSampler(valid_set, shuffle=False) if ddp else None
    train_loader = DataLoader(train_set, batch_size=args.batch_size, shuffle=(train_sampler is None), pin_memory=True,
                              num_workers=args.num_workers, sampler=train_sampler, drop_last=True)
    valid_loader = DataLoader(valid_set, batch_size=args.batch_size, shuffle=False, pin_memory=True,
                              num_workers=args.num_workers, sampler=valid_sampler)

    net = build_net(args.arch, args.width, sync_bn=ddp and args.backend == "nccl", foreground_masks=args.foreground_masks).to(device)
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
    autocast = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if args.amp == "bf16" else contextlib.nullcontext
    if ddp:
        net = nn.parallel.DistributedDataParallel(net, device_ids=[device.index], output_device=device.index)  # main.py:225
    criterion = model.CURLLoss(ssim_window_size=5).to(device)                 # main.py:228
    if args.fused_forward and args.arch != "curl":
        raise SystemExit("--fused_forward is the curve model's (--arch curl)")
    validation_evaluator = evaluate.Evaluator(criterion, valid_loader, "valid", args.log_dirpath, local_rank=rank)  # main.py:233
    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, net.parameters()), lr=5e-7, betas=(0.5, 0.999))
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=1e-4, total_steps=args.num_epoch)
    start_epoch = 0
    if args.checkpoint_filepath:                                              # main.py:241-250
        ckpt = torch.load(args.checkpoint_filepath, map_location=device)
        net.load_state_dict(ckpt["model_state_dict"])
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        start_epoch = ckpt["epoch"]
    optimizer.zero_grad()
    net.train()

    history, step_ms = [], []
    for epoch in range(start_epoch, args.num_epoch):
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)                                    # main.py:262
        running_loss, batches = 0.0, 0
        for batch in train_loader:
            t0 = time.perf_counter()
            img, gt, mask = (batch[k].to(device, non_blocking=True) for k in ("input_img", "output_img", "mask"))
            if args.fused_forward:
                # the curve model's layer and the loss' pointwise terms as ONE forward kernel (curl_layer_loss_fwd_f32)
                with autocast():
                    out, _reg, loss = net(img, mask, target=gt, criterion=criterion)
            else:
                with autocast():
                    out = forward_image(net, img, mask)                       # main.py:283
                loss = criterion(out.float(), gt, mask)                       # main.py:285
            optimizer.zero_grad()
            loss.backward()                                                   # main.py:287
            optimizer.step()
            running_loss += loss.item()                                       # main.py:290 (synchronises)
            batches += 1
            step_ms.append((time.perf_counter() - t0) * 1e3)
        totals = [None] * world
        if ddp:
            dist.all_gather_object(totals, (running_loss, batches))           # main.py:301-303
        else:
            totals[0] = (running_loss, batches)
        entry = {"epoch": epoch + 1, "lr": optimizer.param_groups[0]["lr"],
                 "train_loss": sum(t[0] for t in totals) / max(1, sum(t[1] for t in totals))}
        scheduler.step()
        if (epoch + 1) % args.valid_every == 0:                               # main.py:313-340
            entry["valid_loss"], entry["valid_psnr"], entry["valid_msssim"] = validation_evaluator.evaluate(
                net, epoch, save_images=args.save_images and bool(args.log_dirpath))
            if rank == 0 and args.log_dirpath:
                os.makedirs(args.log_dirpath, exist_ok=True)
                path = os.path.join(args.log_dirpath, "curl_validpsnr_{}_validloss_{}_epoch_{}_model.pt".format(
                    entry["valid_psnr"], entry["valid_loss"], epoch + 1))
                torch.save({"epoch": epoch + 1, "model_state_dict": net.state_dict(),
                            "optimizer_state_dict": optimizer.state_dict(),
                            "scheduler_state_dict": scheduler.state_dict(), "loss": entry["valid_loss"]}, path)
                entry["checkpoint"] = path
        history.append(entry)

    # replicas must hold identical weights after training: a checksum every rank computes and rank 0 compares
    checksum = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum().reshape(1)
    sums = [torch.zeros_like(checksum) for _ in range(world)] if ddp else [checksum]
    if ddp:
        dist.all_gather(sums, checksum)
    if rank == 0:
        steady = sorted(step_ms[len(step_ms) // 4:]) or [float("nan")]
        print(json.dumps({"arch": args.arch, "amp": args.amp, "channels_last": args.channels_last,
                          "world_size": world if ddp else 1, "batch_per_gpu": args.batch_size,
                          "crop": args.crop, "epochs": history, "ms_per_step_median": steady[len(steady) // 2],
                          "images_per_s": (world if ddp else 1) * args.batch_size / (steady[len(steady) // 2] / 1e3),
                          "param_checksums": [float(s) for s in sums],
                          "replicas_identical": all(float(s) == float(sums[0]) for s in sums)}))
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
