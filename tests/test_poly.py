"""The polynomial per-pixel path: oracle vs the golden vectors produced by running the
reference's own classes, the kernel arithmetic (host twin) vs both."""
import os
import re
import sys

import numpy as np
import pytest
import torch

import curl_oracle as O
from conftest import ROOT, max_err


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_power_table_matches_reference(golden):
    g = golden("poly")
    for d, v in ((4, 5), (4, 3), (3, 2)):
        assert np.array_equal(O.poly_powers(d, v).numpy().astype(np.int32), g[f"powers_d{d}_v{v}"])
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT + "/tools")
    from gen_poly_horner import powers
    assert [list(r) for r in powers(4, 5)] == g["powers_d4_v5"].tolist()  # the generated Horner code's indexing


def test_oracle_poly_layers_golden(golden):
    g = golden("poly")
    assert np.allclose(O.channel_poly_layer(t(g["x5"]), t(g["c5"]), 4).numpy(), g["channel_poly_d4v5"], rtol=0, atol=2e-6)
    assert np.allclose(O.deg4_mobile_poly_layer(t(g["x5"]), t(g["c5"])).numpy(), g["mobile_poly"], rtol=0, atol=2e-6)
    assert np.allclose(O.channel_poly_layer(t(g["x3"]), t(g["c3"]), 4).numpy(), g["channel_poly_d4v3"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("s", ["s02", "s1"])
def test_oracle_trispace_golden(golden, s):
    g = golden("poly")
    c = t(g[s + "_coeffs"])
    for nm in ("img", "img8"):
        r = O.trispace_residual(t(g[nm]), c[:, 0], c[:, 1], c[:, 2])
        assert max_err(r.numpy(), g[f"{s}_{nm}_residual"]) <= 3e-6
        assert max_err(O.generate_image(t(g[nm]), r).numpy(), g[f"{s}_{nm}_image"]) <= 3e-6
    c35 = t(g[s + "_coeffs35"])
    r = O.trispace_residual(t(g["img"]), c35[:, 0], c35[:, 1], c35[:, 2], spatial=False)
    assert max_err(r.numpy(), g[f"{s}_img_residual_nonspatial"]) <= 3e-6


def test_twin_poly_layer(twin, golden):
    """Horner form vs the reference's sum of monomials: same polynomial, different rounding; the two reference
    layers themselves differ by 1e-6 at this coefficient scale (model.py:404-409 claims equality)."""
    g = golden("poly")
    assert max_err(twin.poly_layer(g["x5"], g["c5"]), g["mobile_poly"]) <= 3e-6
    assert max_err(twin.poly_layer(g["x5"], g["c5"]), g["channel_poly_d4v5"]) <= 3e-6
    assert max_err(twin.poly_layer(g["x3"], g["c3"]), g["channel_poly_d4v3"]) <= 3e-6


@pytest.mark.parametrize("s,tol", [("s02", 1e-5), ("s1", 2e-5)])
def test_twin_trispace(twin, golden, s, tol):
    g = golden("poly")
    for nm in ("img", "img8"):
        assert max_err(twin.trispace(g[nm], g[s + "_coeffs"], residual_only=True), g[f"{s}_{nm}_residual"]) <= tol, nm
        assert max_err(twin.trispace(g[nm], g[s + "_coeffs"]), g[f"{s}_{nm}_image"]) <= tol, nm
    assert max_err(twin.trispace(g["img"], g[s + "_coeffs35"], residual_only=True), g[f"{s}_img_residual_nonspatial"]) <= tol


@pytest.mark.parametrize("s,tol", [("s02", 1e-5), ("s1", 2e-5)])
def test_twin_trispace_row_collapsed(twin, golden, s, tol):
    """The spatial kernel evaluates a 4-variable polynomial per row (y folded into 70 coefficients per row):
    same values as the reference's 5-variable form."""
    g = golden("poly")
    for nm in ("img", "img8"):
        assert max_err(twin.trispace_rows(g[nm], g[s + "_coeffs"], residual_only=True), g[f"{s}_{nm}_residual"]) <= tol, nm
        assert max_err(twin.trispace_rows(g[nm], g[s + "_coeffs"]), g[f"{s}_{nm}_image"]) <= tol, nm
        assert max_err(twin.trispace_rows(g[nm], g[s + "_coeffs"]), twin.trispace(g[nm], g[s + "_coeffs"])) <= 3e-6, nm


@pytest.mark.parametrize("nc,residual_only", [(126, False), (126, True), (35, False)])
def test_twin_trispace_backward_vs_oracle_autograd(twin, nc, residual_only):
    """d loss / d coeffs of the fused polynomial path vs autograd through the oracle (= the reference's ops)."""
    g = torch.Generator().manual_seed(nc + residual_only)
    B, H, W = 2, 12, 20
    img = torch.rand(B, 3, H, W, generator=g)
    coeffs = (torch.randn(B, 3, 3, nc, generator=g) * 0.3).requires_grad_(True)
    w = torch.randn(B, 3, H, W, generator=g)
    res = O.trispace_residual(img, coeffs[:, 0], coeffs[:, 1], coeffs[:, 2], spatial=(nc == 126))
    out = res if residual_only else O.generate_image(img, res)
    (out * w).sum().backward()
    got = twin.trispace_bwd(img.numpy(), coeffs.detach().numpy(), w.numpy(), residual_only)
    ref = coeffs.grad.numpy()
    assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max()


def test_foldx_index_table_covers_every_coefficient_once():
    """The x-fold of the coefficient gradient (tools/gen_poly_horner.py:gen_foldx): its (monomial, x power) pairs name each
    of the 126 reference coefficients exactly once, and each pair multiplies out to that coefficient's monomial."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_poly_horner as gph
    full, small = gph.powers(4, 5), gph.powers(4, 4)
    text = open(os.path.join(ROOT, "curl_amd", "csrc", "poly_horner.inc")).read()
    body = text[text.index("kPolyFoldXIndex[2][2][46]"):]
    idx = [int(x, 0) for x in re.findall(r"0xFFFF|\d+", body[body.index("=") + 1:body.index(";")])]
    live = [t for t in idx if t != 0xFFFF]
    assert sorted(live) == list(range(126))
    assigns = re.findall(r"if constexpr \(C == (\d) && S == (\d)\) \{(.*?)\n  \}", text[text.index("void foldx_expand"):], re.S)
    seen = 0
    for c, sl, blk in assigns:
        for i, jl, j in re.findall(r"e\[(\d+)\] = a\[(\d+)\](?: \* xp\[(\d)\])?;", blk):
            m = small[int(c) * 35 + int(jl)]
            t = idx[(int(c) * 2 + int(sl)) * 46 + int(i)]
            assert full[t] == m[:3] + (int(j or 0),) + m[3:], (c, sl, i)
            seen += 1
    assert seen == 126


@pytest.mark.parametrize("residual_only", [False, True])
def test_twin_trispace_backward_column_strips(twin, residual_only):
    """The accumulation order of the HIP kernel's spatial backward (per column, x folded out, expanded at the end) gives
    the gradient of the plain 126-monomial form, and the oracle's autograd."""
    g = torch.Generator().manual_seed(7 + residual_only)
    B, H, W = 2, 9, 14
    img = torch.rand(B, 3, H, W, generator=g)
    coeffs = (torch.randn(B, 3, 3, 126, generator=g) * 0.3).requires_grad_(True)
    w = torch.randn(B, 3, H, W, generator=g)
    res = O.trispace_residual(img, coeffs[:, 0], coeffs[:, 1], coeffs[:, 2], spatial=True)
    out = res if residual_only else O.generate_image(img, res)
    (out * w).sum().backward()
    ref = coeffs.grad.numpy()
    got = twin.trispace_bwd_foldx(img.numpy(), coeffs.detach().numpy(), w.numpy(), residual_only)
    plain = twin.trispace_bwd(img.numpy(), coeffs.detach().numpy(), w.numpy(), residual_only)
    assert np.abs(got - plain).max() <= 2e-5 * np.abs(plain).max()
    assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("residual_only", [False, True])
def test_twin_trispace_backward_on_8bit_content(twin, residual_only, golden):
    """The same on what the path is fed in production (infer.py:35-40: bytes / 255): a piece of the reference's own photograph
    with black, white, grey and primary pixels written into it -- exact ties, exact zeros, generate_image's clamp at work --
    against autograd through the oracle in FLOAT64."""
    real = golden("real8")
    u8 = real["crop_u8"][100:112, 60:84].copy()
    u8[0, :6] = [[0, 0, 0], [255, 255, 255], [128, 128, 128], [255, 0, 0], [0, 255, 0], [0, 0, 255]]
    img = O.u8hwc_to_f32chw(u8)[None].repeat(2, 1, 1, 1)
    g = torch.Generator().manual_seed(5 + residual_only)
    coeffs = torch.randn(2, 3, 3, 126, generator=g) * 0.3
    w = torch.randn(2, 3, 12, 24, generator=g)
    c64 = coeffs.double().requires_grad_(True)
    res = O.trispace_residual(img.double(), c64[:, 0], c64[:, 1], c64[:, 2], spatial=True)
    out = res if residual_only else O.generate_image(img.double(), res)
    (out * w.double()).sum().backward()
    ref = c64.grad.numpy()
    got = twin.trispace_bwd(img.numpy(), coeffs.numpy(), w.numpy(), residual_only)
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), float(np.abs(got - ref).max() / np.abs(ref).max())
