"""GPU tests at the sizes BASELINE configs[3] and infer.py reach, and the first execution of this repo's code on RCCL.

* RCCL: `init_process_group("nccl", world_size=1, device_id=cuda:0)` in a fresh child process (main.py:98-100 is the
  reference's call); shard.broadcast_knots / shard.allgather_knots / bench.timed_run's MAX all-reduce on device tensors.
  A 1-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device); N > 1 is the driver's to run.
* B = 256 x 1500x1000 on ONE GPU (config 4's global batch; 4.6 GB per tensor: every byte offset crosses 2^32) and
  B = 960 (4.3 G elements per tensor: every ELEMENT offset crosses 2^32 too): first / last image equal the image
  processed alone, bit for bit.
* one 10000 x 10000 frame (infer.py:32-33 max_resolution; 1.2 GB per image, n = 25 M float4 groups per plane) against the
  oracle on 64-row bands.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, max_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from curl_amd import ops as _ops
    from curl_amd import _lib
    _lib.load()
    return _ops


_RCCL_CHILD = r"""
import json, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["CURL_ROOT"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=dev)   # "nccl" IS RCCL on ROCm
from curl_amd import shard, ops, _lib
_lib.load()
import bench
res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
g = torch.Generator().manual_seed(0)
knots = torch.randn(5, 160, generator=g).to(dev)
buf = knots.clone()
shard.broadcast_knots(buf, src=0)                       # ncclBroadcast on a device tensor
res["bcast"] = bool(torch.equal(buf, knots))
res["gather"] = bool(torch.equal(shard.allgather_knots(knots[1:4].clone()), knots[1:4]))   # ncclAllGather x2
# bench.py's timing protocol with the process group in place: barrier, K steps, barrier, MAX all-reduce on the device
img = torch.rand(2, 3, 64, 96, generator=g).to(dev)
L, R, H = ((torch.randn(2, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
bench.CLOCK_SETTLE_LAUNCHES = 3
wall, dev_ms, dev_min = bench.timed_run(lambda s: ops.curl_layer_forward(s[0], None, s[1], s[2], s[3]),
                                        [(img, L, R, H)], 5, 2, dist, dev)
res["timed"] = [wall, dev_ms, dev_min]
# split-pixels layout end to end on one rank: broadcast knots -> this rank's row slab in place
from curl_amd import model
layer = model.CURLLayer()
out, reg, (r0, r1) = shard.apply_row_slab(layer, img, None, buf[:2, :48], buf[:2, 48:96], buf[:2, 96:], 0, 1)
full, freg = ops.curl_layer_forward(img, None, buf[:2, :48].contiguous(), buf[:2, 48:96].contiguous(), buf[:2, 96:].contiguous())
res["slab"] = bool((r0, r1) == (0, 64) and torch.equal(out, full) and torch.equal(reg, freg))
dist.barrier()
dist.destroy_process_group()
print("RCCL_CHILD " + json.dumps(res))
"""


def test_rccl_single_rank_collectives_in_a_child_process(dev, tmp_path):
    """RCCL initialised by this repo's code for the first time (VERDICT r2 missing 2): a fresh process, one rank."""
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD)
    env = dict(os.environ, CURL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL_CHILD ")][-1]
    res = json.loads(line[len("RCCL_CHILD "):])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["bcast"] and res["gather"] and res["slab"]
    wall, dev_ms, dev_min = res["timed"]
    assert wall > 0 and dev_ms > 0 and abs(dev_ms - dev_min) < 1e-9   # MAX and MIN over one rank agree


def _batch(B, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    img = torch.rand(B, 3, 1000, 1500, device=dev, generator=g)
    L, R, Hk = (torch.randn(B, n, device=dev, generator=g) * 0.1 for n in (48, 48, 64))
    return img, L, R, Hk


def test_bs256_one_gpu_first_and_last_image_equal_the_image_alone(ops, dev):
    """BASELINE configs[3]'s global batch on ONE GPU: 256 x 1500x1000 = 384 Mpix, 4.6 GB per tensor -- image 255 starts
    4.59 GB into every tensor, past 2^32 bytes in every offset expression of every kernel touched here."""
    B, H, W = 256, 1000, 1500
    img, L, R, Hk = _batch(B, dev, 256)
    assert img.numel() * 4 > 2 ** 32
    mask = torch.rand(B, 1, H, W, device=dev) > 0.25
    ends = (0, B - 1)

    out, reg = ops.curl_layer_forward(img, mask, L, R, Hk)
    for b in ends:
        o1, r1 = ops.curl_layer_forward(img[b:b + 1], mask[b:b + 1], L[b:b + 1], R[b:b + 1], Hk[b:b + 1])
        assert torch.equal(o1[0], out[b]) and torch.equal(r1[0], reg[b]), b
    assert (out * (~mask) == 0).all()

    u8 = ops.f32chw_to_u8hwc(out)
    for b in ends:
        assert torch.equal(ops.f32chw_to_u8hwc(out[b:b + 1])[0], u8[b]), b
    back = ops.u8hwc_to_f32chw(u8)
    for b in ends:
        assert torch.equal(ops.u8hwc_to_f32chw(u8[b:b + 1])[0], back[b]), b
    del back, u8

    p = ops.psnr_per_image(out, img, mask)
    for b in ends:
        assert torch.equal(ops.psnr_per_image(out[b:b + 1], img[b:b + 1], mask[b:b + 1])[0], p[b]), b

    for name, fn in (("lab_stage", lambda i, m, s: ops.lab_stage(i, m, L[s])), ("hsv_stage", lambda i, m, s: ops.hsv_stage(i, m, Hk[s])),
                     ("adjust_rgb", lambda i, m, s: ops.adjust_rgb(i, R[s]))):
        o, r = fn(img, mask, slice(None))
        for b in ends:
            o1, r1 = fn(img[b:b + 1], mask[b:b + 1], slice(b, b + 1))
            assert torch.equal(o1[0], o[b]) and torch.equal(r1[0], r[b]), (name, b)
        del o
    del out

    coeffs = torch.randn(B, 3, 3, 126, device=dev) * 0.2
    tri = ops.trispace_forward(img, coeffs)
    for b in ends:
        assert torch.equal(ops.trispace_forward(img[b:b + 1], coeffs[b:b + 1])[0], tri[b]), b
    # the fused byte path at the same batch
    u8in = (img * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    t8 = ops.trispace_forward_u8hwc(u8in, coeffs)
    l8, _ = ops.curl_layer_forward_u8hwc(u8in, mask, L, R, Hk)
    for b in ends:
        assert torch.equal(ops.trispace_forward_u8hwc(u8in[b:b + 1], coeffs[b:b + 1])[0], t8[b]), b
        assert torch.equal(ops.curl_layer_forward_u8hwc(u8in[b:b + 1], mask[b:b + 1], L[b:b + 1], R[b:b + 1], Hk[b:b + 1])[0][0], l8[b]), b


def test_bs256_backward_first_and_last_image(ops, dev):
    """config 5's kernels at config 4's batch: the layer backward's image gradient and knot gradients of images 0 and
    255 equal those of the image alone (per-image reductions, fixed order: bit-reproducible)."""
    B = 256
    img, L, R, Hk = _batch(B, dev, 257)
    mask = torch.rand(B, 1, 1000, 1500, device=dev) > 0.25
    gout = torch.rand_like(img)
    greg = torch.rand(B, device=dev)
    gi, gL, gR, gH = ops.curl_layer_backward(img, mask, L, R, Hk, gout, greg)
    for b in (0, B - 1):
        s = slice(b, b + 1)
        gi1, gL1, gR1, gH1 = ops.curl_layer_backward(img[s], mask[s], L[s], R[s], Hk[s], gout[s], greg[s])
        assert torch.equal(gi1[0], gi[b]), b
        # the second reduction pass walks the block partials in an order that depends only on the image's own blocks
        assert torch.equal(gL1[0], gL[b]) and torch.equal(gR1[0], gR[b]) and torch.equal(gH1[0], gH[b]), b


def test_bs960_element_offsets_cross_2_to_32(ops, dev):
    """4.32 G floats per tensor (17.3 GB): the element index of image 959 exceeds 2^32 as well as its byte offset."""
    B, H, W = 960, 1000, 1500
    g = torch.Generator(device=dev).manual_seed(960)
    img = torch.rand(B, 3, H, W, device=dev, generator=g)
    assert img.numel() > 2 ** 32
    L, R, Hk = (torch.randn(B, n, device=dev, generator=g) * 0.1 for n in (48, 48, 64))
    out, reg = ops.curl_layer_forward(img, None, L, R, Hk)
    for b in (0, 700, B - 1):
        o1, r1 = ops.curl_layer_forward(img[b:b + 1], None, L[b:b + 1], R[b:b + 1], Hk[b:b + 1])
        assert torch.equal(o1[0], out[b]) and torch.equal(r1[0], reg[b]), b
    ops.curl_layer_forward(img, None, L, R, Hk, out=img)  # in place over the whole 17 GB
    assert torch.equal(img[B - 1], out[B - 1]) and torch.equal(img[0], out[0])


def test_one_frame_10000x10000_vs_oracle_bands(ops, dev):
    """infer.py:32-33 / model.py:442,491: max_resolution = 10000.  One 10000 x 10000 frame (100 Mpix, 1.2 GB, 25 M float4
    groups per plane -- the top of the range `__builtin_assume(a.n <= 1 << 28)` speaks about) through the curve layer,
    the polynomial path (whose y = row / H must be the full image's) and the byte edge, against the oracle on 64-row
    bands at the top, in the middle and at the bottom."""
    import curl_oracle as O
    H = W = 10000
    g = torch.Generator(device=dev).manual_seed(10000)
    img = torch.rand(1, 3, H, W, device=dev, generator=g)
    L, R, Hk = (torch.randn(1, n, device=dev, generator=g) * 0.1 for n in (48, 48, 64))
    mask = torch.rand(1, 1, H, W, device=dev, generator=g) > 0.2
    coeffs = torch.randn(1, 3, 3, 126, device=dev, generator=g) * 0.2
    out, reg = ops.curl_layer_forward(img, mask, L, R, Hk)
    tri = ops.trispace_forward(img, coeffs)
    u8 = ops.f32chw_to_u8hwc(tri)
    c = coeffs.cpu()
    for r0 in (0, 4968, H - 64):
        band = img[:, :, r0:r0 + 64].cpu()
        mb = mask[:, :, r0:r0 + 64].cpu().float()
        ref, rreg = O.curl_layer(band, mb, L.cpu(), R.cpu(), Hk.cpu())
        S = O.input_sensitivity(band, mb, L.cpu(), R.cpu(), Hk.cpu())
        d = (out[:, :, r0:r0 + 64].cpu().double() - ref.double()).abs().amax(1)
        bound = torch.clamp(2e-6 * S, min=1e-5)   # north_star's 1e-5 off the ill-conditioned pixels
        assert int((d > bound).sum()) == 0, (r0, float((d / bound).max()))
        np.testing.assert_allclose(reg.cpu().numpy(), rreg.numpy(), rtol=2e-6)
        res = O.trispace_residual(band, c[:, 0], c[:, 1], c[:, 2], rows=(r0, H))
        want = O.generate_image(band, res)
        got = tri[:, :, r0:r0 + 64].cpu()
        assert max_err(got.numpy(), want.numpy()) <= 2e-5, r0   # the polynomial path's bar
        assert np.array_equal(u8[:, r0:r0 + 64].cpu().numpy()[0], O.f32chw_to_u8hwc(got[0])), r0
    # the row-slab entry at this size: the last 1250 rows (an 8-way split's last slab) in place == the whole-image call
    slab = torch.zeros_like(img)
    ops.curl_layer_forward_rows(img, mask, L, R, Hk, (8750, 10000), slab)
    assert torch.equal(slab[:, :, 8750:], out[:, :, 8750:]) and not slab[:, :, :8750].any()


def test_one_image_at_the_api_limit_of_2_to_30_pixels(ops, dev):
    """check_img admits H*W <= 2^30 pixels, and the streaming kernels rely on it: a lane's BYTE offset inside a plane is 32-bit
    (stream.inc at(): 2^28 float4 groups x 16 B, or 2^30 floats x 4 B on the scalar path).  One 32768 x 32768 image
    (12.9 GB per tensor): the last rows -- the largest offsets -- and the first must equal the same rows processed as an
    image of their own (the path is pixel-wise; the knots are the same), float4 and scalar kernels, layer / PSNR / byte edge;
    one pixel more is refused."""
    H = W = 32768
    g = torch.Generator(device=dev).manual_seed(30)
    buf = torch.empty(3 * H * W + 1, device=dev)
    buf.uniform_(generator=g)
    img = buf[:-1].view(1, 3, H, W)
    mask = torch.rand(1, 1, H, W, device=dev, generator=g) > 0.2
    L, R, Hk = (torch.randn(1, n, device=dev, generator=g) * 0.1 for n in (48, 48, 64))
    out, reg = ops.curl_layer_forward(img, mask, L, R, Hk)
    bands = (slice(0, 8), slice(H - 8, H))

    def alone(x, rows):
        return x[:, :, rows].contiguous()

    for rows in bands:
        o1, _ = ops.curl_layer_forward(alone(img, rows), alone(mask, rows), L, R, Hk)
        assert torch.equal(o1, out[:, :, rows]), rows
    u8 = ops.f32chw_to_u8hwc(out)
    for rows in bands:
        assert torch.equal(ops.f32chw_to_u8hwc(alone(out, rows)), u8[:, rows]), rows
    del u8
    # PSNR of (out, img) under the mask: the reduction walks every block of the 2^30 pixels; against torch on the GPU
    p = ops.psnr_per_image(out, img, mask)
    m = mask.float()
    se = torch.zeros((), dtype=torch.float64, device=dev)
    for c in range(3):  # channel by channel: no 12.9 GB temporaries
        se += (((out[0, c].clamp(0, 1) - img[0, c].clamp(0, 1)) * m[0, 0]) ** 2).sum(dtype=torch.float64)
    want = 10.0 * torch.log10(1.0 / (se / (3.0 * m.sum(dtype=torch.float64))))
    assert abs(float(p[0]) - float(want)) <= 1e-4 * abs(float(want))
    del m
    # scalar kernels: the same pixels from a base pointer 4 bytes off (every float4 test fails -> VEC = 1, i up to 2^30 - 1)
    first = img.flatten()[0].clone()
    view = buf[1:].view(1, 3, H, W)
    assert view.data_ptr() % 16 == 4
    del out
    o_s, _ = ops.curl_layer_forward(view, mask, L, R, Hk)
    for rows in bands:
        small = torch.empty(3 * 8 * W + 1, device=dev)[1:].view(1, 3, 8, W)  # the band alone, off the boundary as well
        small.copy_(view[:, :, rows])
        o1, _ = ops.curl_layer_forward(small, alone(mask, rows), L, R, Hk)
        assert torch.equal(o1, o_s[:, :, rows]), rows
    assert float(first) == float(buf[0])
    del o_s, view, img, buf, mask
    torch.cuda.empty_cache()
    with pytest.raises(ValueError, match="2\\^30"):  # CURL_E_SHAPE
        ops.curl_layer_forward(torch.empty(1, 3, 1, 2 ** 30 + 1, device=dev), None, L, R, Hk)
