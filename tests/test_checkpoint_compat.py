"""A checkpoint saved by the reference's train loop (main.py:332-338: DDP-wrapped TriSpaceRegNet around timm's
efficientnetv2_rw_t) must load into curl_amd's model key for key.

timm is not installed here, so the expected key list is written down from timm 0.5.4's public definition
(efficientnet.py `_gen_efficientnetv2_s`, efficientnet_blocks.py) independently of curl_amd/model.py, and pinned by
the parameter counts timm publishes for the two variants (13.65 M and 23.94 M)."""
import math
import os
import re

import pytest
import torch

from curl_amd import infer, model
from curl_amd.convert_state import convert_state_dict

BN = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")


def _divisible(v, d=8):
    n = max(d, int(v + d / 2) // d * d)
    return n + d if n < 0.9 * v else n


def timm_v2_keys(variant):
    """{key: shape} of timm.create_model(variant).state_dict(), written from the architecture string."""
    if variant == "efficientnetv2_rw_t":
        arch, cm, dm, feat = ["cn_r2_k3_s1_e1_c24_skip", "er_r4_k3_s2_e4_c48", "er_r4_k3_s2_e4_c64",
                              "ir_r6_k3_s2_e4_c128_se0.25", "ir_r9_k3_s1_e6_c160_se0.25",
                              "ir_r15_k3_s2_e6_c256_se0.25"], 0.8, 0.9, 1280
    else:
        arch, cm, dm, feat = ["er_r2_k3_s1_e1_c24", "er_r4_k3_s2_e4_c48", "er_r4_k3_s2_e4_c64",
                              "ir_r6_k3_s2_e4_c128_se0.25", "ir_r9_k3_s1_e6_c160_se0.25",
                              "ir_r15_k3_s2_e6_c272_se0.25"], 1.0, 1.0, 1792
    keys = {}

    def bn(prefix, c):
        for k in BN:
            keys[f"{prefix}.{k}"] = () if k == "num_batches_tracked" else (c,)

    stem = _divisible(24 * cm)
    keys["conv_stem.weight"] = (stem, 3, 3, 3)
    bn("bn1", stem)
    cin = stem
    for s, spec in enumerate(arch):
        f = dict(re.match(r"([a-z]+)(.*)", p).groups() for p in spec.split("_")[1:] if p != "skip")
        kind, reps, k, exp, cout = spec.split("_")[0], int(f["r"]), int(f["k"]), int(f["e"]), _divisible(int(f["c"]) * cm)
        se = float(f.get("se", 0))
        for i in range(math.ceil(reps * dm)):
            p = f"blocks.{s}.{i}"
            mid = _divisible(cin * exp)
            if kind == "cn":
                keys[f"{p}.conv.weight"] = (cout, cin, k, k)
                bn(f"{p}.bn1", cout)
            elif kind == "er":
                keys[f"{p}.conv_exp.weight"] = (mid, cin, k, k)
                bn(f"{p}.bn1", mid)
                keys[f"{p}.conv_pwl.weight"] = (cout, mid, 1, 1)
                bn(f"{p}.bn2", cout)
            else:
                keys[f"{p}.conv_pw.weight"] = (mid, cin, 1, 1)
                bn(f"{p}.bn1", mid)
                keys[f"{p}.conv_dw.weight"] = (mid, 1, k, k)
                bn(f"{p}.bn2", mid)
                rd = round(mid * se / exp)
                keys[f"{p}.se.conv_reduce.weight"], keys[f"{p}.se.conv_reduce.bias"] = (rd, mid, 1, 1), (rd,)
                keys[f"{p}.se.conv_expand.weight"], keys[f"{p}.se.conv_expand.bias"] = (mid, rd, 1, 1), (mid,)
                keys[f"{p}.conv_pwl.weight"] = (cout, mid, 1, 1)
                bn(f"{p}.bn3", cout)
            cin = cout
    nf = _divisible(feat * cm)
    keys["conv_head.weight"] = (nf, cin, 1, 1)
    bn("bn2", nf)
    keys["classifier.weight"], keys["classifier.bias"] = (1000, nf), (1000,)
    return keys


@pytest.mark.parametrize("variant,published_mparams", [("efficientnetv2_rw_t", 13.65), ("efficientnetv2_rw_s", 23.94)])
def test_encoder_has_timm_keys_shapes_and_size(variant, published_mparams):
    net = model.EfficientNetV2(variant)
    have = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    want = timm_v2_keys(variant)
    assert list(have) == list(want)  # same keys in the same registration order
    assert have == want
    n = sum(p.numel() for p in net.parameters()) / 1e6
    assert abs(n - published_mparams) < 0.006, n  # timm's model table
    assert net.eval()(torch.rand(1, 3, 64, 64)).shape == (1, 1000)


def _reference_trispace_checkpoint(spatial=True, prefix="module."):
    """What main.py:332-338 saves for the DDP-wrapped live model (model.py:439-484): timm backbone with the 4-layer
    head, the polynomial layer's `powers`, the colour constants (colors.py), and the frozen `x` / `y` ramps."""
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shape in timm_v2_keys("efficientnetv2_rw_t").items():
        if k.startswith("classifier."):
            continue
        sd["backbone." + k] = torch.zeros(shape, dtype=torch.long) if k.endswith("num_batches_tracked") \
            else torch.randn(shape, generator=g) * 0.05
    nc = 126 if spatial else 35
    for i, (a, b) in enumerate(((1024, 1024), (1024, 512), (512, 512), (512, 9 * nc))):  # model.py:459-463
        sd[f"backbone.classifier.{i}.weight"] = torch.randn(b, a, generator=g) * 0.01
        sd[f"backbone.classifier.{i}.bias"] = torch.zeros(b)
    sd["polylayer.powers"] = torch.zeros(nc, 5 if spatial else 3)
    for k, shape in (("rgb2lab.rgb_to_xyz", (1, 1, 3, 3)), ("rgb2lab.fxfyfz_to_lab", (1, 1, 3, 3)),
                     ("rgb2lab.xyz_to_rgb_mult", (1, 1, 1, 3)), ("rgb2lab.lab_to_fxfyfz_offset", (1, 1, 1, 3)),
                     ("lab2rgb.xyz_to_rgb", (1, 1, 3, 3)), ("lab2rgb.lab_to_fxfyfz", (1, 1, 3, 3)),
                     ("lab2rgb.xyz_to_rgb_mult", (1, 1, 1, 3)), ("lab2rgb.lab_to_fxfyfz_offset", (1, 1, 1, 3)),
                     ("rgb2hsv.comparison_zero", None)):
        sd[k] = None  # filled from the module below (values are constants; shapes are what is checked)
    if spatial:  # model.py:476-484
        sd["x"] = torch.arange(0, 10000).reshape(1, 1, 1, 10000)
        sd["y"] = torch.arange(0, 10000).reshape(1, 1, 10000, 1)
    else:
        sd["x"], sd["y"] = torch.zeros(1, 0, 1, 10000), torch.zeros(1, 0, 10000, 1)
    return {prefix + k: v for k, v in sd.items()}


@pytest.mark.parametrize("spatial", [True, False])
def test_reference_trispace_checkpoint_loads_strict(spatial, tmp_path):
    net = model.TriSpaceRegNet(polynomial_order=4, spatial=spatial, is_train=False,
                               polylayer=model.Deg4MobilePolyLayer() if spatial else None)
    mine = net.state_dict()
    ckpt = _reference_trispace_checkpoint(spatial)
    for k in list(ckpt):  # colour constants: take the values of our modules (equal to the reference's, tested elsewhere)
        if ckpt[k] is None:
            ckpt[k] = mine[k[len("module."):]].clone()
    # every key of a reference checkpoint exists here with the same shape and dtype class, and nothing is left over
    conv = convert_state_dict(ckpt)
    assert set(conv) == set(mine), (sorted(set(conv) - set(mine))[:5], sorted(set(mine) - set(conv))[:5])
    for k, v in conv.items():
        assert tuple(v.shape) == tuple(mine[k].shape), k
        assert v.is_floating_point() == mine[k].is_floating_point(), k
    net.load_state_dict(conv, strict=True)
    assert torch.equal(net.backbone.blocks[3][1].se.conv_reduce.weight, conv["backbone.blocks.3.1.se.conv_reduce.weight"])
    assert net.x.dtype == (torch.int64 if spatial else torch.float32) and not net.x.requires_grad
    # the infer.py path: torch.load -> convert_state_dict -> load_state_dict (infer.py:25-29)
    if spatial:
        path = os.path.join(tmp_path, "ref_ckpt.pt")
        torch.save({"epoch": 3, "model_state_dict": ckpt}, path)
        loaded = infer.build_net(path, torch.device("cpu"), arch="trispace")
        assert torch.equal(loaded.backbone.conv_stem.weight, conv["backbone.conv_stem.weight"])


def test_non_backbone_keys_equal_the_reference_class():
    """The reference's TriSpaceRegNet cannot be imported (timm, torchvision); its non-backbone state is what its
    constructor registers (model.py:451-484): read the registrations from the source text and compare the names."""
    path = "/root/reference/model.py"
    if not os.path.isfile(path):
        pytest.skip("reference tree not present")
    src = open(path).read()
    body = src[src.index("class TriSpaceRegNet"):]
    body = body[:body.index("def cat_coords")]
    attrs = set(re.findall(r"self\.(\w+)\s*=\s*(?:torch\.nn\.Parameter|nn\.Parameter|colors\.\w+\(|polylayer|timm\.)", body))
    assert {"x", "y", "rgb2lab", "lab2rgb", "rgb2hsv", "hsv2rgb", "polylayer", "backbone"} <= attrs
    mine = {k.split(".")[0] for k in model.TriSpaceRegNet(spatial=True).state_dict()}
    assert mine == {"x", "y", "rgb2lab", "lab2rgb", "rgb2hsv", "polylayer", "backbone"}  # hsv2rgb holds no tensor
