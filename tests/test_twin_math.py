"""The kernel arithmetic (curl_amd/csrc/curl_math.h) compiled for the host by tests/twin, against the
reference's golden vectors and the oracle.  libm replaces v_log_f32 / v_exp_f32 / v_rcp_f32, everything
else (thresholds, tie handling, folded constants, curve collapse, cascade summation) is the code the
GPU runs.  This is the CPU-side gate before a kernel change goes to the GPU box."""
import numpy as np
import pytest
import torch

import curl_oracle as O
from conftest import max_err


@pytest.mark.parametrize("op,inp,outp", [
    ("rgb2lab", "rgb_in", "rgb2lab_out"), ("rgb2lab", "rgbwide_in", "rgb2lab_wide_out"),
    ("lab2rgb", "lab_in", "lab2rgb_out"), ("rgb2hsv", "rgb_in", "rgb2hsv_out"),
    ("rgb2hsv", "rgbwide_in", "rgb2hsv_wide_out"), ("hsv2rgb", "hsv_in", "hsv2rgb_out")])
def test_converters(twin, golden, op, inp, outp):
    g = golden("converters")
    assert max_err(twin.convert(op, g[inp]), g[outp]) <= 1e-6


def test_apply_curve_exact_order_is_bitexact(twin, golden):
    g = golden("apply_curve")
    for case, K, ci, co, which in g["meta"]:
        x = g["in_wide" if which else "in_unit"]
        out, reg = twin.apply_curve(x, g[f"c{case}_C"], g[f"c{case}_reg0"], int(ci), int(co), mode=1)
        assert np.array_equal(out, g[f"c{case}_out"]), (case, K)
        np.testing.assert_allclose(reg, g[f"c{case}_reg"], rtol=2e-6)


@pytest.mark.parametrize("K", [18, 19, 33, 40, 100, 256])
def test_exact_order_follows_torch_cascade_sum(twin, K):
    """K-2 > 16 terms: ATen's cascade sum flushes every 16 terms; the kernel mimics it (curl_math.h)."""
    g = torch.Generator().manual_seed(K)
    x = torch.rand(1, 3, 8, 16, generator=g) * 2 - 0.5
    C = torch.exp(torch.randn(1, K, generator=g) * 0.3)
    ref, _ = O.apply_curve(x, C, torch.zeros(1), 1, 2)
    out, _ = twin.apply_curve(x.numpy(), C.numpy(), None, 1, 2, mode=1)
    assert np.array_equal(out, ref.numpy())


def test_affine_collapse_within_reference_noise(twin, golden):
    g = golden("apply_curve")
    for case, K, ci, co, which in g["meta"]:
        x = g["in_wide" if which else "in_unit"]
        out, _ = twin.apply_curve(x, g[f"c{case}_C"], None, int(ci), int(co), mode=0)
        assert max_err(out, g[f"c{case}_out"]) <= 1e-5, case


def test_pwl_mode_is_the_paper_curve(twin):
    """Non-parity option: scale = piecewise-linear interpolation of the knots at x (paper eq. 1)."""
    g = torch.Generator().manual_seed(3)
    K = 16
    x = torch.rand(1, 3, 4, 32, generator=g)
    C = torch.exp(torch.randn(1, K, generator=g) * 0.3)
    out, _ = twin.apply_curve(x.numpy(), C.numpy(), None, 0, 0, mode=2)
    xs = x[0, 0].numpy().astype(np.float64)
    scale = np.interp(xs * (K - 1), np.arange(K), C[0].numpy().astype(np.float64))
    want = np.clip(xs * scale, 0, 1)
    assert np.abs(out[0, 0] - want).max() < 1e-5


@pytest.mark.parametrize("sig,tol", [("s01", 2e-6), ("s05", 1e-5)])
def test_adjust(twin, golden, sig, tol):
    c = golden("chain")
    for name, n, key in (("rgb", 3, "_R"), ("lab", 3, "_L"), ("hsv", 4, "_H")):
        for mode in (0, 1):
            out, reg = twin.adjust(n, c["img"], c[sig + key], mode)
            assert max_err(out, c[f"{sig}_adjust_{name}_out"]) <= tol, (name, mode)
            np.testing.assert_allclose(reg, c[f"{sig}_adjust_{name}_reg"], rtol=2e-6)


def test_layer_sigma01(twin, golden):
    c = golden("chain")
    L, R, H = c["s01_L"], c["s01_R"], c["s01_H"]
    for inn in ("img", "img8"):
        for mk in ("ones", "holes", "disk", "soft"):
            key = f"s01_{inn}_{mk}"
            if key + "_out" not in c:
                continue
            m = c["mask_" + mk].astype(np.float32)
            out, reg = twin.layer(1, c[inn], m, L, R, H)
            assert max_err(out, c[key + "_out"]) <= 1e-5, key
            np.testing.assert_allclose(reg, c[key + "_reg"], rtol=2e-6)
            ls, _ = twin.layer(0, c[inn], m, L, R, H)
            assert max_err(ls, c[key + "_lab_stage"]) <= 1e-5, key
            if mk != "soft":  # 0/1 masks: the binary-mask specialisation (skipped multiplies, masked-out shortcut)
                outb, _ = twin.layer(1, c[inn], m, L, R, H, binary=True)
                lsb, _ = twin.layer(0, c[inn], m, L, R, H, binary=True)
                assert np.array_equal(lsb, ls), key
                # the layer's binary form also drops the 1e-9 floors of colors.py:205,240 inside the HSV stage
                # (inputs and outputs there are already in [0,1]): not bit-identical, but far below the bar
                assert max_err(outb, out) <= 3e-7 and max_err(outb, c[key + "_out"]) <= 1e-5, key


def test_layer_error_is_reference_noise_sized(twin):
    """On 0.5 Mpix the kernel arithmetic is no further from float64 truth than the reference's own float32
    evaluation is (DESIGN.md 'Parity'): the chain is ill-conditioned at dark / near-grey pixels."""
    g = torch.Generator().manual_seed(0)
    B, H, W = 2, 256, 512
    img = torch.rand(B, 3, H, W, generator=g)
    m = torch.ones(B, 1, H, W)
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (48, 48, 64))
    for stage in (0, 1):
        if stage == 0:
            o32, _ = O.lab_stage(img, m, L)
            o64, _ = O.lab_stage(img.double(), m.double(), L.double())
        else:
            o32, _ = O.curl_layer(img, m, L, R, Hk)
            o64, _ = O.curl_layer(img.double(), m.double(), L.double(), R.double(), Hk.double())
        tw, _ = twin.layer(stage, img.numpy(), m.numpy(), L.numpy(), R.numpy(), Hk.numpy())
        ref_noise = np.abs(o32.numpy() - o64.numpy())
        tw_noise = np.abs(tw - o64.numpy())
        assert tw_noise.max() <= 1.5 * ref_noise.max() + 1e-6
        assert tw_noise.mean() <= 1.5 * ref_noise.mean() + 1e-8
        d = np.abs(tw - o32.numpy())
        assert (d > 1e-5).mean() < 2e-4 and d.max() < 1e-4


def test_reference_sum_order_depends_on_shape(twin):
    """Why 'bit-exact' carries a shape condition: ATen's sum kernel handles the last H*W mod 32 pixels of an
    image outside its vector body, with a different accumulation order.  Exact-order mode reproduces the
    vector-body order for EVERY pixel: identical bits where H*W % 32 == 0, 1-ulp-level noise elsewhere."""
    for shape, must_match in (((2, 16, 24), True), ((1, 33, 64), True), ((2, 8, 8), True), ((2, 7, 9), False),
                              ((3, 16, 17), False)):
        B, H, W = shape
        g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
        x = torch.rand(B, 3, H, W, generator=g) * 1.2 - 0.1
        C = torch.exp(torch.randn(B, 16, generator=g) * 0.1)
        ref = O.apply_curve(x, C, torch.zeros(B), 2, 0)[0].numpy()
        out, _ = twin.apply_curve(x.numpy(), C.numpy(), None, 2, 0, mode=1)
        if must_match:
            assert np.array_equal(out, ref), shape
        assert max_err(out, ref) <= 1e-6, shape


def test_file_edge_scalars(twin):
    """u8_to_unit is to_tensor's byte/255 for every byte (a Newton-corrected multiply, no division), and
    unit_to_u8 is (x*255).astype('uint8') on [0,1] (evaluate.py:64), saturating outside."""
    g = np.random.default_rng(0)
    x = np.concatenate([g.random(4096, dtype=np.float32), np.arange(256, dtype=np.float32) / np.float32(255),
                        np.array([0.0, 1.0, -0.3, 1.7, 0.99999994], np.float32)])
    unit, q = twin.u8_edges(x)
    want = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255).numpy()
    assert np.array_equal(unit, want)
    inside = (x >= 0) & (x <= 1)
    assert np.array_equal(q[inside], (x[inside] * np.float32(255)).astype(np.uint8))
    assert q[-3] == 0 and q[-2] == 255


@pytest.mark.parametrize("seed", [3034, 1, 2])
def test_layer_binary_form_on_ties_and_out_of_range_inputs(twin, seed):
    """Inputs in [-0.1, 1.1] drive channels into the clamps, so the HSV stage sees exact ties (g == b == 1 adds up
    to hue sextant 6 = hue 1.0, which must NOT wrap to 0: the hue curves are not periodic) and exact zeros (where
    the binary form has dropped the reference's 1e-9 floors).  Caught a fract()-based hue wrap once."""
    B, H, W = 2, 1, 1024
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g) * 1.2 - 0.1
    img[0, :, 0, :64] = torch.round(img[0, :, 0, :64] * 4) / 4          # coarse grid: many exact ties
    mask = torch.rand(B, 1, H, W, generator=g) > 0.25
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (48, 48, 64))
    ref, _ = O.curl_layer(img, mask.float(), L, R, Hk)
    r64, _ = O.curl_layer(img.double(), mask.double(), L.double(), R.double(), Hk.double())
    noise = max_err(ref.numpy(), r64.numpy())  # the reference's own float32 rounding on these inputs (~1e-5)
    for binary in (True, False):
        out, _ = twin.layer(1, img.numpy(), mask.float().numpy(), L.numpy(), R.numpy(), Hk.numpy(), binary=binary)
        assert max_err(out, r64.numpy()) <= 1.5 * noise + 1e-6, binary
        assert max_err(out, ref.numpy()) <= 3e-5, binary


def test_coordinate_division_is_exact(twin):
    """cat_coords divides column by width (model.py:487-497); the kernels use a corrected reciprocal multiply."""
    assert twin.div_small_mismatches(8192) == 0


@pytest.mark.parametrize("sig,tol", [("s01", 3e-6), ("s05", 1e-5)])
def test_hsv_stage(twin, golden, sig, tol):
    """hsv_stage_n (curl_math.h; model.py:163-169) on the host against the reference's outputs, every input kind (unit
    range, 8-bit grid with channel ties, out of range) and mask kind; the binary-mask form is bit-identical."""
    c = golden("hsv_stage")
    H = c[sig + "_H"]
    z = np.zeros((2, 48), np.float32)
    for inn in ("img", "img8", "wide"):
        for mk in ("ones", "holes", "disk", "soft"):
            key = f"{sig}_{inn}_{mk}"
            m = c["mask_" + mk].astype(np.float32)
            out, reg = twin.layer(2, c[inn], m, z, z, H)
            ref = c[key + "_out"]
            d = np.abs(out.astype(np.float64) - ref)
            assert d.max() <= tol, (key, d.max())  # exact channel ties (img8) included: tie terms are reproduced exactly
            np.testing.assert_allclose(reg, c[key + "_reg"], rtol=2e-6)
            if mk != "soft":
                outb, _ = twin.layer(2, c[inn], m, z, z, H, binary=True)
                assert np.array_equal(outb, out), key
