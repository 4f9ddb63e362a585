"""Backward arithmetic (curl_amd/csrc/curl_math_bwd.h) on the host twin against (1) the golden gradients
produced by autograd THROUGH THE REFERENCE's primitives and (2) autograd through the oracle on random data,
incl. tie / clamp / mask corner cases."""
import numpy as np
import pytest
import torch

import curl_oracle as O


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def oracle_grads(img, mask, L, R, H, w, wr):
    x = img.clone().requires_grad_(True)
    Lg, Rg, Hg = (t.clone().requires_grad_(True) for t in (L, R, H))
    out, reg = O.curl_layer(x, mask, Lg, Rg, Hg)
    ((out * w).sum() + (reg * wr).sum()).backward()
    return x.grad, Lg.grad, Rg.grad, Hg.grad


@pytest.mark.parametrize("sig", ["s01", "s05"])
def test_golden_gradients(twin, golden, sig):
    c = golden("chain")
    gi, gL, gR, gH = twin.layer_bwd(c["img"], c["mask_disk"].astype(np.float32), c[sig + "_L"], c[sig + "_R"],
                                    c[sig + "_H"], c[sig + "_grad_w"], c[sig + "_grad_wr"])
    tol = 2e-4 if sig == "s01" else 5e-3  # strong curves: the reference's own fp32 gradient noise grows
    assert rel(gL, c[sig + "_grad_L"]) <= tol
    assert rel(gR, c[sig + "_grad_R"]) <= tol
    assert rel(gH, c[sig + "_grad_H"]) <= tol
    d = np.abs(gi - c[sig + "_grad_img"])
    scale = np.abs(c[sig + "_grad_img"]).max()
    assert np.quantile(d, 0.999) <= tol * scale and d.max() <= 20 * tol * scale


@pytest.mark.parametrize("case", ["random", "grid8", "saturated", "softmask"])
def test_vs_oracle_autograd(twin, case):
    g = torch.Generator().manual_seed({"random": 11, "grid8": 22, "saturated": 33, "softmask": 44}[case])
    B, H, W = 2, 24, 32
    img = torch.rand(B, 3, H, W, generator=g)
    mask = torch.ones(B, 1, H, W)
    if case == "grid8":
        img = torch.randint(0, 256, (B, 3, H, W), generator=g).float() / 255  # ties on the 8-bit grid
        img[:, 1, :4] = img[:, 0, :4]
        img[:, 2, 4:8] = img[:, 1, 4:8]
    if case == "saturated":
        img = (img * 1.6 - 0.3)  # out-of-range inputs: clamps active
    if case == "softmask":
        mask = torch.rand(B, 1, H, W, generator=g)
    else:
        mask = (torch.rand(B, 1, H, W, generator=g) > 0.2).float()
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (48, 48, 64))
    w = torch.randn(B, 3, H, W, generator=g)
    wr = torch.rand(B, generator=g)
    ref = oracle_grads(img, mask, L, R, Hk, w, wr)
    got = twin.layer_bwd(img.numpy(), mask.numpy(), L.numpy(), R.numpy(), Hk.numpy(), w.numpy(), wr.numpy())
    for name, a, b in zip(("img", "L", "R", "H"), got, ref):
        b = b.numpy()
        if name == "img":
            d = np.abs(a - b)
            scale = np.abs(b).max()
            # isolated pixels sit on a discontinuity of the reference (tie / threshold flips under rounding)
            assert np.quantile(d, 0.995) <= 3e-4 * scale, (case, name)
        else:
            assert rel(a, b) <= 1e-3, (case, name, rel(a, b))


STRUCTURED_COLOURS = np.array([[0, 0, 0], [1, 1, 1], [0, 0, 1], [1, 0, 0], [0, 1, 0], [2, 2, 2], [10, 10, 10], [128, 128, 128],
                               [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255],
                               [0, 0, 128], [128, 0, 0], [0, 128, 0], [10, 10, 0], [0, 10, 10]], np.float32) / 255


def structured_case(seed, shift):
    n = len(STRUCTURED_COLOURS)
    g = torch.Generator().manual_seed(seed)
    img = torch.from_numpy(STRUCTURED_COLOURS.T.reshape(1, 3, 1, n).copy())
    mask = torch.ones(1, 1, 1, n)
    L, R, Hk = (torch.randn(1, k, generator=g) * 0.1 + shift for k in (48, 48, 64))
    w = torch.randn(1, 3, 1, n, generator=g)
    wr = torch.rand(1, generator=g)
    return img, mask, L, R, Hk, w, wr


@pytest.mark.parametrize("shift", [-0.7, -0.3, 0.0])
def test_black_and_the_corners_of_the_colour_cube(twin, shift):
    """Colours every photograph is full of and random floats never hit: BLACK, near-black, greys, white, the primaries and
    secondaries, single dark channels.  At black the reference's L is exactly 0 in float32 (116 * float32(4/29) rounds to 16.0,
    colors.py:50-56) and >= 0 in float64, and torch.clamp's gate at 0 passes the gradient on; an L of -1.6e-9 -- the single-fma
    form 1.16 fy - 0.16 the backward had until round 4 -- closed it on all 2 194 black pixels of the `dark` photograph, and
    the exception set of the float64-pinned GPU test (a gradient jump within 1e-6 of the input: it IS one) hid that.  Here
    nothing is excused: every colour within 5e-6 of the gradient scale of the float64 gradient, as the reference's own float32
    autograd is."""
    for seed in (3, 4, 5):
        img, mask, L, R, Hk, w, wr = structured_case(seed, shift)
        g64 = O.layer_gradients(img, mask, L, R, Hk, w, wr)[0].numpy()
        g32 = O.layer_gradients(img, mask, L, R, Hk, w, wr, dtype=torch.float32)[0].numpy()
        got = twin.layer_bwd(img.numpy(), mask.numpy(), L.numpy(), R.numpy(), Hk.numpy(), w.numpy(), wr.numpy())[0]
        tol = 5e-6 * max(1.0, float(np.abs(g64).max()))
        assert np.abs(g32 - g64).max() <= tol, "the reference's own float32 gradient is unambiguous on these colours"
        d = np.abs(got - g64)[0, :, 0]
        assert d.max() <= tol, (shift, seed, (STRUCTURED_COLOURS[d.max(0).argmax()] * 255).tolist(), float(d.max()))


def test_regulariser_gradient_only(twin):
    """gout = 0: the knot gradient is the regulariser's alone (curves.py:19,24 through exp)."""
    g = torch.Generator().manual_seed(5)
    B = 2
    img = torch.rand(B, 3, 4, 4, generator=g)
    mask = torch.ones(B, 1, 4, 4)
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.3 for n in (48, 48, 64))
    wr = torch.rand(B, generator=g) + 0.5
    ref = oracle_grads(img, mask, L, R, Hk, torch.zeros(B, 3, 4, 4), wr)
    got = twin.layer_bwd(img.numpy(), mask.numpy(), L.numpy(), R.numpy(), Hk.numpy(), np.zeros((B, 3, 4, 4)), wr.numpy())
    for a, b in zip(got[1:], ref[1:]):
        assert rel(a, b.numpy()) <= 2e-5
    assert np.abs(got[0]).max() == 0


@pytest.mark.parametrize("case", ["random", "grid8", "saturated"])
def test_binary_mask_specialisation_equals_the_general_path(twin, case):
    """bool / uint8 masks run curl_layer_bwd<BINARY>: no intermediate `* mask`, input clamps of the HSV stages known to
    pass.  On 0/1 masks it must give what the general (float-mask) path gives -- the same values, pixel for pixel."""
    g = torch.Generator().manual_seed({"random": 1, "grid8": 2, "saturated": 3}[case])
    B, H, W = 2, 16, 24
    img = torch.rand(B, 3, H, W, generator=g)
    if case == "grid8":
        img = torch.randint(0, 256, (B, 3, H, W), generator=g).float() / 255
        img[:, 1, :4] = img[:, 0, :4]
        img[:, 2, 4:8] = img[:, 1, 4:8]
    if case == "saturated":
        img = img * 1.6 - 0.3
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).float()
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (48, 48, 64))
    w = torch.randn(B, 3, H, W, generator=g)
    wr = torch.rand(B, generator=g)
    a = twin.layer_bwd(img.numpy(), mask.numpy(), L.numpy(), R.numpy(), Hk.numpy(), w.numpy(), wr.numpy())
    b = twin.layer_bwd(img.numpy(), mask.numpy(), L.numpy(), R.numpy(), Hk.numpy(), w.numpy(), wr.numpy(), binary=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.abs(a[0] * (1 - mask.numpy())).max() == 0
