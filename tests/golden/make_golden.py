"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

Every array written here is either a seeded input or the output of the
reference's own code on that input:
  curves.apply_curve            /root/reference/curves.py:4-38
  colors.RGB2LAB/LAB2RGB/...    /root/reference/colors.py
  transpose.swapimdims_*        /root/reference/transpose.py
  metric.PSNRMetric             /root/reference/metric.py:28-72
The adjust_* wrappers and CURLLayer.forward cannot be run as written
(SURVEY.md section 0.2); for those the stage ORDER is restated below
(curves.py:53-80,105-126,152-173; model.py:150-176 minus the `feat` lines)
over the imported primitives, with the regulariser seeded by zeros(B).
Nothing from oracle/ or curl_amd/ is used: these files pin both.
"""
import hashlib
import os
import sys

import numpy as np
import torch

REF = os.environ.get("CURL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import colors  # noqa: E402
import curves  # noqa: E402
import metric  # noqa: E402
import transpose  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def npy(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ reference-side stage order
def ref_adjust(img, raw, pairs):
    img = img.contiguous()
    parts = [torch.exp(p) for p in torch.chunk(raw, len(pairs), dim=1)]
    reg = torch.zeros(img.shape[0])
    for C, (ci, co) in zip(parts, pairs):
        img, reg = curves.apply_curve(img, C, reg, ci, co)
    return img.clone().contiguous(), reg


RGB_PAIRS = [(0, 0), (1, 1), (2, 2)]
HSV_PAIRS = [(0, 0), (0, 1), (1, 1), (2, 2)]


def ref_layer(img, mask, L, R, H, stages=None):
    rgb2lab, lab2rgb = colors.RGB2LAB(), colors.LAB2RGB()
    rgb2hsv, hsv2rgb = colors.RGB2HSV(), colors.HSV2RGB()
    lab = rgb2lab(img)
    lab, reg_lab = ref_adjust(lab, L[:, :48], RGB_PAIRS)
    lab = lab * mask
    rgb = lab2rgb(lab)
    if stages is not None:
        stages["after_lab_stage"] = rgb
    rgb, reg_rgb = ref_adjust(rgb, R[:, :48], RGB_PAIRS)
    rgb = rgb * mask
    if stages is not None:
        stages["after_rgb_stage"] = rgb
    hsv = rgb2hsv(rgb)
    hsv, reg_hsv = ref_adjust(hsv, H[:, :64], HSV_PAIRS)
    hsv = hsv * mask
    if stages is not None:
        stages["after_hsv_stage"] = hsv
    res = hsv2rgb(hsv)
    out = torch.clamp(img + res, 0.0, 1.0) * mask
    if stages is not None:
        stages["reg_lab"], stages["reg_rgb"], stages["reg_hsv"] = reg_lab, reg_rgb, reg_hsv
    return out, reg_rgb + reg_lab + reg_hsv


# ------------------------------------------------------------------ 1. apply_curve
def gen_apply_curve():
    g = torch.Generator().manual_seed(1234)
    B, H, W = 2, 16, 24
    store = {}
    inputs = {
        "unit": torch.rand(B, 3, H, W, generator=g),
        "wide": torch.rand(B, 3, H, W, generator=g) * 2.0 - 0.5,
    }
    for k, v in inputs.items():
        store[f"in_{k}"] = npy(v)
    sigmas = [0.1, 0.5, 1.0]
    pairs = [(0, 0), (0, 1), (1, 1), (2, 2), (2, 0)]
    case = 0
    meta = []
    for K in (2, 4, 16, 17):
        for (ci, co) in pairs:
            for rng in ("unit", "wide"):
                sigma = sigmas[case % 3]
                C = torch.exp(torch.randn(B, K, generator=g) * sigma)
                reg0 = torch.rand(B, generator=g)
                reg = reg0.clone()
                out, reg = curves.apply_curve(inputs[rng], C, reg, ci, co)
                store[f"c{case}_C"] = npy(C)
                store[f"c{case}_reg0"] = npy(reg0)
                store[f"c{case}_reg"] = npy(reg)
                store[f"c{case}_out"] = npy(out)
                meta.append([case, K, ci, co, 0 if rng == "unit" else 1])
                case += 1
    store["meta"] = np.array(meta, dtype=np.int32)  # case, K, cin, cout, input(0 unit / 1 wide)
    np.savez_compressed(os.path.join(OUT, "apply_curve.npz"), **store)
    print("apply_curve cases:", case)


# ------------------------------------------------------------------ 2. converters
def f32_neighbours(v, n=3):
    """v and its +-1..n float32 neighbours."""
    a = np.float32(v)
    out = [a]
    lo = hi = a
    for _ in range(n):
        lo = np.nextafter(lo, np.float32(-np.inf), dtype=np.float32)
        hi = np.nextafter(hi, np.float32(np.inf), dtype=np.float32)
        out += [lo, hi]
    return out


def pixel_set_rgb(g):
    px = []
    # ties, grey, black, white, primaries/secondaries
    px += [(0.7, 0.7, 0.2), (0.2, 0.6, 0.6), (0.6, 0.2, 0.6), (0.5, 0.5, 0.5), (0, 0, 0), (1, 1, 1),
           (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1),
           (0.3, 0.3, 0.3000001), (0.8, 0.3, 0.3000001), (0.8, 0.3000001, 0.3), (1e-9, 1e-9, 1e-9),
           (1e-10, 0.5, 2e-10), (1.0, 0.9999999, 1.0)]
    # sRGB threshold neighbours
    for v in f32_neighbours(0.04045):
        px += [(v, 0.5, 0.01), (0.01, v, 0.9), (v, v, v)]
    for v in (1e-5, 1e-4, 0.00011, 0.003, 0.01, 0.02):
        px += [(v, v, v), (v, 0.0, 1.0)]
    px = torch.tensor(px, dtype=torch.float32)
    # 8-bit grid sample
    grid = torch.randint(0, 256, (600, 3), generator=g).float() / 255
    # a grey ramp on the 8 bit grid (all ties)
    ramp = (torch.arange(0, 256, 5).float() / 255)[:, None].repeat(1, 3)
    rnd = torch.rand(900, 3, generator=g)
    allpx = torch.cat([px, grid, ramp, rnd], 0)
    return allpx


def to_img(px, W=32):
    n = px.shape[0]
    H = (n + W - 1) // W
    pad = H * W - n
    if pad:
        px = torch.cat([px, px[:pad]], 0)
    return px.t().reshape(1, 3, H, W).contiguous()


def gen_converters():
    g = torch.Generator().manual_seed(4321)
    store = {}
    rgb = to_img(pixel_set_rgb(g))
    store["rgb_in"] = npy(rgb)
    store["rgb2lab_out"] = npy(colors.RGB2LAB()(rgb.clone()))
    store["rgb2hsv_out"] = npy(colors.RGB2HSV()(rgb.clone()))
    # out-of-range rgb into RGB2HSV (what LAB2RGB + curves hand it is clamped, but the op accepts anything)
    wide = torch.rand(1, 3, 16, 32, generator=g) * 3 - 1
    store["rgbwide_in"] = npy(wide)
    store["rgb2hsv_wide_out"] = npy(colors.RGB2HSV()(wide.clone()))
    store["rgb2lab_wide_out"] = npy(colors.RGB2LAB()(wide.clone()))

    # Lab inputs: images of real colours, plus threshold neighbours, plus random (out of gamut)
    lab_px = [npy(colors.RGB2LAB()(rgb.clone()))[0].reshape(3, -1).T]
    eps = 6.0 / 29.0
    thr = []
    for f in f32_neighbours(eps):
        # fy = (100 L + 16)/116 = f  -> L = (116 f - 16)/100
        Lh = (116.0 * float(f) - 16.0) / 100.0
        thr += [(Lh, 0.5, 0.5), (Lh, 0.6, 0.4), (Lh, 0.45, 0.52)]
    for L_ in (0.0, 0.01, 0.05, 0.0799, 0.08, 0.0801, 0.2, 1.0):
        thr += [(L_, 0.5, 0.5), (L_, 0.3, 0.7), (L_, 0.9, 0.1)]
    lab_px.append(np.array(thr, dtype=np.float32))
    lab_px.append(npy(torch.rand(700, 3, generator=g)))
    lab = to_img(torch.from_numpy(np.concatenate(lab_px, 0).astype(np.float32)))
    store["lab_in"] = npy(lab)
    store["lab2rgb_out"] = npy(colors.LAB2RGB()(lab.clone()))
    # linear-rgb threshold 0.0031308: greys around it through the full inverse
    hsv_px = [npy(colors.RGB2HSV()(rgb.clone()))[0].reshape(3, -1).T]
    sext = []
    for hdeg in (0, 59.99999, 60, 60.00001, 120, 180, 240, 300, 359.9999, 360):
        for s_, v_ in ((1.0, 1.0), (0.5, 0.7), (1e-9, 0.3), (0.999, 1e-9)):
            sext.append((hdeg / 360.0, s_, v_))
    hsv_px.append(np.array(sext, dtype=np.float32))
    hsv_px.append(npy(torch.rand(600, 3, generator=g)))
    hsv_px.append(npy(torch.rand(200, 3, generator=g) * 3 - 1))
    hsv = to_img(torch.from_numpy(np.concatenate(hsv_px, 0).astype(np.float32)))
    store["hsv_in"] = npy(hsv)
    store["hsv2rgb_out"] = npy(colors.HSV2RGB()(hsv.clone()))
    np.savez_compressed(os.path.join(OUT, "converters.npz"), **store)
    print("converters: rgb", tuple(rgb.shape), "lab", tuple(lab.shape), "hsv", tuple(hsv.shape))


# ------------------------------------------------------------------ 3. adjust_* and the full chain
def make_masks(B, H, W, g):
    ones = torch.ones(B, 1, H, W)
    holes = (torch.rand(B, 1, H, W, generator=g) > 0.3)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    disk = (((yy - H / 2) ** 2 / (H / 2) ** 2 + (xx - W / 2) ** 2 / (W / 2) ** 2) < 0.8)
    soft = torch.rand(B, 1, H, W, generator=g)  # fractional f32 mask (the ops accept any float)
    return {"ones": ones, "holes": holes, "disk": disk[None, None].repeat(B, 1, 1, 1), "soft": soft}


def gen_chain():
    g = torch.Generator().manual_seed(777)
    B, H, W = 2, 32, 48
    store = {}
    img = torch.rand(B, 3, H, W, generator=g)
    img8 = torch.randint(0, 256, (B, 3, H, W), generator=g).float() / 255
    store["img"] = npy(img)
    store["img8"] = npy(img8)
    masks = make_masks(B, H, W, g)
    for k, m in masks.items():
        store[f"mask_{k}"] = npy(m)
    for sig_name, sigma in (("s01", 0.1), ("s05", 0.5)):
        L = torch.randn(B, 48, generator=g) * sigma
        R = torch.randn(B, 48, generator=g) * sigma
        Hh = torch.randn(B, 64, generator=g) * sigma
        store[f"{sig_name}_L"], store[f"{sig_name}_R"], store[f"{sig_name}_H"] = npy(L), npy(R), npy(Hh)
        # adjust_* alone, on in-range and (for rgb) out-of-range input
        o, r = ref_adjust(img, R, RGB_PAIRS)
        store[f"{sig_name}_adjust_rgb_out"], store[f"{sig_name}_adjust_rgb_reg"] = npy(o), npy(r)
        wide = img * 2 - 0.5
        o, r = ref_adjust(wide, R, RGB_PAIRS)
        store[f"{sig_name}_adjust_rgbwide_out"] = npy(o)
        o, r = ref_adjust(img, L, RGB_PAIRS)
        store[f"{sig_name}_adjust_lab_out"], store[f"{sig_name}_adjust_lab_reg"] = npy(o), npy(r)
        o, r = ref_adjust(img, Hh, HSV_PAIRS)
        store[f"{sig_name}_adjust_hsv_out"], store[f"{sig_name}_adjust_hsv_reg"] = npy(o), npy(r)
        for in_name, x in (("img", img), ("img8", img8)):
            for mk, m in masks.items():
                if in_name == "img8" and mk not in ("ones", "disk"):
                    continue
                stages = {}
                mf = m.float() if m.dtype == torch.bool else m
                out, reg = ref_layer(x, mf, L, R, Hh, stages)
                key = f"{sig_name}_{in_name}_{mk}"
                store[f"{key}_out"] = npy(out)
                store[f"{key}_reg"] = npy(reg)
                store[f"{key}_lab_stage"] = npy(stages["after_lab_stage"])
                if mk == "ones" and in_name == "img":
                    store[f"{key}_rgb_stage"] = npy(stages["after_rgb_stage"])
                    store[f"{key}_hsv_stage"] = npy(stages["after_hsv_stage"])
                    store[f"{key}_reg_lab"] = npy(stages["reg_lab"])
                    store[f"{key}_reg_rgb"] = npy(stages["reg_rgb"])
                    store[f"{key}_reg_hsv"] = npy(stages["reg_hsv"])
        # gradients (autograd through the reference primitives): loss = sum(out*w) + sum(reg*wr)
        x = img.clone().requires_grad_(True)
        Lg, Rg, Hg = (t.clone().requires_grad_(True) for t in (L, R, Hh))
        w = torch.rand(B, 3, H, W, generator=g)
        wr = torch.rand(B, generator=g)
        out, reg = ref_layer(x, masks["disk"].float(), Lg, Rg, Hg)
        ((out * w).sum() + (reg * wr).sum()).backward()
        store[f"{sig_name}_grad_w"], store[f"{sig_name}_grad_wr"] = npy(w), npy(wr)
        store[f"{sig_name}_grad_img"] = npy(x.grad)
        store[f"{sig_name}_grad_L"], store[f"{sig_name}_grad_R"], store[f"{sig_name}_grad_H"] = \
            npy(Lg.grad), npy(Rg.grad), npy(Hg.grad)
    # masked PSNR (metric.py)
    a = torch.rand(B, 3, H, W, generator=g)
    b = torch.clamp(a + 0.05 * torch.randn(B, 3, H, W, generator=g), -0.2, 1.2)
    store["psnr_a"], store["psnr_b"] = npy(a), npy(b)
    store["psnr_val"] = npy(metric.PSNRMetric()(a, b, masks["disk"].float()))
    np.savez_compressed(os.path.join(OUT, "chain.npz"), **store)
    print("chain: keys", len(store))


# ------------------------------------------------------------------ 4. config 1 (image stays in the reference)
def gen_config1():
    try:
        from PIL import Image
    except ImportError:
        print("PIL missing: config-1 fixture skipped")
        return
    rel = "adobe5k_dpe/curl_example_test_input/a4723-_DGW7894_input.png"
    im = Image.open(os.path.join(REF, rel)).convert("RGB")  # file is RGBA 512x341
    arr = np.asarray(im)
    Hh, Ww = arr.shape[:2]
    top, left = (Hh - 256) // 2, (Ww - 256) // 2
    crop = arr[top:top + 256, left:left + 256]
    x = torch.from_numpy(np.ascontiguousarray(transpose.swapimdims_HW3_3HW(crop))).float().div(255)[None]
    g = torch.Generator().manual_seed(99)
    L = torch.randn(1, 48, generator=g) * 0.1
    R = torch.randn(1, 48, generator=g) * 0.1
    Hk = torch.randn(1, 64, generator=g) * 0.1
    out, reg = ref_layer(x, torch.ones(1, 1, 256, 256), L, R, Hk)
    u8 = (npy(out)[0] * 255).astype("uint8")  # evaluate.py:64
    hwc = np.ascontiguousarray(transpose.swapimdims_3HW_HW3(u8))  # evaluate.py:66
    np.savez_compressed(
        os.path.join(OUT, "config1.npz"),
        image_relpath=np.array(rel), crop_top_left=np.array([top, left]),
        L=npy(L), R=npy(R), H=npy(Hk), reg=npy(reg),
        in_sha256=np.array(hashlib.sha256(npy(x).tobytes()).hexdigest()),
        out_f32_sha256=np.array(hashlib.sha256(npy(out).tobytes()).hexdigest()),
        out_u8hwc_sha256=np.array(hashlib.sha256(hwc.tobytes()).hexdigest()),
        in_corner=npy(x)[0, :, :16, :16], out_corner=npy(out)[0, :, :16, :16], out_u8_corner=hwc[:16, :16],
        out_mean=np.array(float(out.double().mean())))
    print("config1 ok", hwc.shape)


# ------------------------------------------------------------------ 5. layout edges
def gen_layout():
    g = torch.Generator().manual_seed(5)
    store = {}
    chw = torch.rand(3, 7, 11, generator=g).numpy()
    bchw = torch.rand(2, 3, 5, 6, generator=g).numpy()
    store["chw"], store["bchw"] = chw, bchw
    store["chw_to_hwc"] = np.ascontiguousarray(transpose.swapimdims_3HW_HW3(chw))
    store["bchw_to_bhwc"] = np.ascontiguousarray(transpose.swapimdims_3HW_HW3(bchw))
    hwc = store["chw_to_hwc"]
    store["hwc_to_chw"] = np.ascontiguousarray(transpose.swapimdims_HW3_3HW(hwc))
    store["bhwc_to_bchw"] = np.ascontiguousarray(transpose.swapimdims_HW3_3HW(store["bchw_to_bhwc"]))
    # truncating egress (evaluate.py:64-66): values k/255 +- a hair, 0, 1
    vals = np.concatenate([np.arange(256, dtype=np.float32) / 255,
                           np.nextafter(np.arange(256, dtype=np.float32) / 255, np.float32(0)),
                           np.nextafter(np.arange(256, dtype=np.float32) / 255, np.float32(2)),
                           torch.rand(384, generator=g).numpy()]).astype(np.float32)
    vals = np.clip(vals, 0, 1)
    f = vals[: (vals.size // (3 * 16)) * 3 * 16].reshape(3, -1, 16)
    store["egress_in"] = f
    u8 = (f * 255).astype("uint8")
    store["egress_out"] = np.ascontiguousarray(transpose.swapimdims_3HW_HW3(u8))
    # ingest: u8 HWC(A) -> f32 CHW /255 (what PIL + to_tensor do at infer.py:37; torchvision is absent,
    # so this one is numpy arithmetic, not a run of the reference)
    rgba = torch.randint(0, 256, (9, 13, 4), generator=g, dtype=torch.uint8).numpy()
    store["ingest_rgba"] = rgba
    store["ingest_out"] = np.ascontiguousarray(
        transpose.swapimdims_HW3_3HW(rgba[..., :3])).astype(np.float32) / np.float32(255)
    np.savez_compressed(os.path.join(OUT, "layout.npz"), **store)
    print("layout ok")


if __name__ == "__main__":
    gen_apply_curve()
    gen_converters()
    gen_chain()
    gen_config1()
    gen_layout()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
