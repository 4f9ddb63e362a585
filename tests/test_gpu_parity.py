"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the reference and
against the oracle on seeded inputs.  Tolerances: SURVEY.md section 8(d) / BASELINE.json 1e-5
(max|out-ref| / max(1,max|ref|)); the reference's own float32-vs-float64 noise is the yardstick
where the chain is ill-conditioned ("Parity")."""
import numpy as np
import pytest
import torch

from conftest import max_err

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
    _lib.load()  # fail loudly if the HIP library is missing
    return _ops


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ converters
@pytest.mark.parametrize("op,inp,outp,tol", [
    ("rgb2lab", "rgb_in", "rgb2lab_out", 2e-6), ("rgb2lab", "rgbwide_in", "rgb2lab_wide_out", 2e-6),
    ("lab2rgb", "lab_in", "lab2rgb_out", 2e-6),
    ("rgb2hsv", "rgb_in", "rgb2hsv_out", 1e-6), ("rgb2hsv", "rgbwide_in", "rgb2hsv_wide_out", 1e-6),
    ("hsv2rgb", "hsv_in", "hsv2rgb_out", 1e-6)])
def test_converters_golden(ops, dev, golden, op, inp, outp, tol):
    g = golden("converters")
    out = getattr(ops, op)(T(g[inp], dev))
    assert max_err(N(out), g[outp]) <= tol


# ------------------------------------------------------------------ apply_curve
def test_apply_curve_golden_exact_order_bitexact(ops, dev, golden):
    g = golden("apply_curve")
    for case, K, ci, co, which in g["meta"]:
        x = T(g["in_wide" if which else "in_unit"], dev)
        reg = T(g[f"c{case}_reg0"], dev).clone()
        out, reg = ops.apply_curve(x, T(g[f"c{case}_C"], dev), reg, int(ci), int(co))
        assert np.array_equal(N(out), g[f"c{case}_out"]), f"case {case} K={K} {ci}->{co}"
        np.testing.assert_allclose(N(reg), g[f"c{case}_reg"], rtol=2e-6)


def test_apply_curve_golden_affine(ops, dev, golden):
    g = golden("apply_curve")
    for case, K, ci, co, which in g["meta"]:
        x = T(g["in_wide" if which else "in_unit"], dev)
        out, _ = ops.apply_curve(x, T(g[f"c{case}_C"], dev), None, int(ci), int(co), flags=0)
        assert max_err(N(out), g[f"c{case}_out"]) <= 1e-5, f"case {case}"


# ------------------------------------------------------------------ adjust_* and the fused chain
@pytest.mark.parametrize("sig,tol", [("s01", 2e-6), ("s05", 1e-5)])
def test_adjust_golden(ops, dev, golden, sig, tol):
    c = golden("chain")
    img = T(c["img"], dev)
    for name, fn, key in (("rgb", ops.adjust_rgb, "_R"), ("lab", ops.adjust_lab, "_L"), ("hsv", ops.adjust_hsv, "_H")):
        for flags in (0, 1):
            out, reg = fn(img, T(c[sig + key], dev), flags=flags)
            assert max_err(N(out), c[f"{sig}_adjust_{name}_out"]) <= tol, (name, flags)
            np.testing.assert_allclose(N(reg), c[f"{sig}_adjust_{name}_reg"], rtol=2e-6)
    out, _ = ops.adjust_rgb(T(c["img"] * 2 - 0.5, dev), T(c[sig + "_R"], dev))
    assert max_err(N(out), c[f"{sig}_adjust_rgbwide_out"]) <= tol


def _mask_variants(c, mk, dev):
    m = c["mask_" + mk]
    yield T(m.astype(np.float32), dev)
    if m.dtype == np.bool_:
        yield T(m, dev)
    if mk == "ones":
        yield None


def test_layer_golden_sigma01(ops, dev, golden):
    """Realistic near-identity curves (raw knots ~ N(0, 0.1), the bench configuration): 1e-5."""
    c = golden("chain")
    L, R, H = (T(c["s01" + k], dev) for k in ("_L", "_R", "_H"))
    for inn in ("img", "img8"):
        for mk in ("ones", "holes", "disk", "soft"):
            key = f"s01_{inn}_{mk}"
            if key + "_out" not in c:
                continue
            for m in _mask_variants(c, mk, dev):
                out, reg = ops.curl_layer_forward(T(c[inn], dev), m, L, R, H)
                assert max_err(N(out), c[key + "_out"]) <= 1e-5, key
                np.testing.assert_allclose(N(reg), c[key + "_reg"], rtol=2e-6)
                ls, rl = ops.lab_stage(T(c[inn], dev), m, L)
                assert max_err(N(ls), c[key + "_lab_stage"]) <= 1e-5, key


def test_layer_golden_sigma05_conditioning(ops, dev, golden):
    """Strong curves (sigma 0.5) make the chain ill-conditioned: the reference's own float32 result is
    ~2e-4 from its float64 evaluation on a handful of pixels.  Bar: all but a few pixels within 1e-5 of
    the reference, and no pixel further from float64 truth than 4x the reference's own worst pixel."""
    import curl_oracle as O
    c = golden("chain")
    L, R, H = (c["s05" + k] for k in ("_L", "_R", "_H"))
    img = c["img"]
    o64, _ = O.curl_layer(*(torch.from_numpy(a).double() for a in (img, c["mask_ones"].astype(np.float64), L, R, H)))
    ref = c["s05_img_ones_out"]
    out, reg = ops.curl_layer_forward(T(img, dev), None, T(L, dev), T(R, dev), T(H, dev))
    out = N(out)
    d = np.abs(out.astype(np.float64) - ref)
    assert (d > 1e-5).mean() < 5e-3
    ref_noise = np.abs(ref.astype(np.float64) - o64.numpy()).max()
    assert np.abs(out.astype(np.float64) - o64.numpy()).max() <= 4 * ref_noise + 1e-5
    np.testing.assert_allclose(N(reg), c["s05_img_ones_reg"], rtol=2e-6)


# ------------------------------------------------------------------ mask kinds agree bit for bit
def test_binary_mask_paths_agree(ops, dev, golden):
    """bool / uint8 masks take the binary specialisation (dropped multiplies, masked-out shortcut);
    a float mask holding the same 0/1 values takes the general path.  Same bits out for the Lab stage and between
    the two binary kinds; the layer's binary form also drops the 1e-9 floors inside its HSV stage (values there
    are already in [0,1]), which moves results by < 3e-7."""
    c = golden("chain")
    L, R, H = (T(c["s01" + k], dev) for k in ("_L", "_R", "_H"))
    img = T(c["img"], dev)
    for mk in ("holes", "disk", "ones"):
        mb = T(c["mask_" + mk].astype(np.bool_), dev)
        mf = mb.float()
        mu = mb.to(torch.uint8)
        o_b, _ = ops.curl_layer_forward(img, mb, L, R, H)
        o_f, _ = ops.curl_layer_forward(img, mf, L, R, H)
        o_u, _ = ops.curl_layer_forward(img, mu, L, R, H)
        assert torch.equal(o_b, o_u), mk
        assert float((o_b - o_f).abs().max()) <= 3e-7, mk
        l_b, _ = ops.lab_stage(img, mb, L)
        l_f, _ = ops.lab_stage(img, mf, L)
        assert torch.equal(l_b, l_f), mk
    o_none, _ = ops.curl_layer_forward(img, None, L, R, H)
    o_ones, _ = ops.curl_layer_forward(img, torch.ones(2, 1, 32, 48, dtype=torch.bool, device=dev), L, R, H)
    assert torch.equal(o_none, o_ones)


def test_fully_masked_waves_take_the_shortcut(ops, dev):
    """A mask with large all-zero regions (whole wavefronts masked out) must still equal the oracle."""
    import curl_oracle as O
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 64, 256  # 16384 px per image = 16 tiles of 1024 px
    img = torch.rand(B, 3, H, W, generator=g)
    mask = torch.zeros(B, 1, H, W, dtype=torch.bool)
    mask[0, :, 10:20, :] = True          # a band: most waves fully masked, some fully live, some mixed
    mask[1, :, :, 100:103] = True        # thin column: every wave mixed
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (48, 48, 64))
    ref, _ = O.curl_layer(img, mask.float(), L, R, Hk)
    ref_ls, _ = O.lab_stage(img, mask.float(), L)
    for m in (mask, mask.float()):
        out, _ = ops.curl_layer_forward(img.to(dev), m.to(dev), L.to(dev), R.to(dev), Hk.to(dev))
        assert max_err(N(out), ref.numpy()) <= 1e-5
        assert (N(out)[~mask.expand(B, 3, H, W).numpy()] == 0).all()
        ls, _ = ops.lab_stage(img.to(dev), m.to(dev), L.to(dev))
        assert max_err(N(ls), ref_ls.numpy()) <= 1e-5
    zero = torch.zeros(B, 1, H, W, dtype=torch.bool, device=dev)
    out, _ = ops.curl_layer_forward(img.to(dev), zero, L.to(dev), R.to(dev), Hk.to(dev))
    assert (out == 0).all()


# ------------------------------------------------------------------ shapes, alignment, knots
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 7, 9), (1, 3, 5), (3, 16, 17), (1, 33, 64), (2, 1, 1024), (1, 31, 33)])
def test_odd_shapes_and_tails(ops, dev, shape):
    """H*W not a multiple of 4 (scalar kernels), tiles with ragged tails, single pixels."""
    import curl_oracle as O
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    img = torch.rand(B, 3, H, W, generator=g) * 1.2 - 0.1
    mask = torch.rand(B, 1, H, W, generator=g) > 0.25
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (48, 48, 64))
    ref, rreg = O.curl_layer(img, mask.float(), L, R, Hk)
    out, reg = ops.curl_layer_forward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev))
    assert max_err(N(out), ref.numpy()) <= 1e-5
    np.testing.assert_allclose(N(reg), rreg.numpy(), rtol=2e-6)
    for name in ("rgb2lab", "lab2rgb", "rgb2hsv", "hsv2rgb"):
        assert max_err(N(getattr(ops, name)(img.to(dev))), getattr(O, name)(img).numpy()) <= 2e-6, name
    o2, _ = ops.adjust_hsv(img.to(dev), Hk.to(dev))
    assert max_err(N(o2), O.adjust_hsv(img, Hk)[0].numpy()) <= 2e-6
    # exact-order mode = the summation order of torch's VECTOR body.  torch reduces the last (H*W mod 32)
    # pixels of an image with interleaved accumulators, so the reference is only bit-reproducible across
    # shapes where H*W is a multiple of 32 (measured in tests/test_twin_math.py); elsewhere 1-ulp noise.
    o3, _ = ops.apply_curve(img.to(dev), torch.exp(L[:, :16]).to(dev), None, 2, 0)
    r3 = O.apply_curve(img, torch.exp(L[:, :16]), torch.zeros(B), 2, 0)[0].numpy()
    if (H * W) % 32 == 0:
        assert np.array_equal(N(o3), r3)
    else:
        assert max_err(N(o3), r3) <= 1e-6


def test_misaligned_base_pointer(ops, dev):
    """A tensor whose storage starts 4 bytes off a 16-byte boundary falls back to the scalar kernels."""
    import curl_oracle as O
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 8, 16
    x = torch.rand(B, 3, H, W, generator=g)
    buf = torch.empty(B * 3 * H * W + 1, device=dev)
    view = buf[1:].view(B, 3, H, W)
    view.copy_(x)
    assert view.data_ptr() % 16 == 4 and view.is_contiguous()
    assert max_err(N(ops.rgb2lab(view)), O.rgb2lab(x).numpy()) <= 2e-6
    R = torch.randn(B, 48, generator=g) * 0.1
    out, _ = ops.adjust_rgb(view, R.to(dev))
    assert max_err(N(out), O.adjust_rgb(x, R)[0].numpy()) <= 2e-6


@pytest.mark.parametrize("Kl,Kr,Kh", [(16, 16, 16), (8, 24, 5), (2, 3, 4), (33, 17, 64)])
def test_knot_counts(ops, dev, Kl, Kr, Kh):
    """K != 16 per space (CURLLayer(num_lab_points=3*Kl, ...)); K = 2 is the degenerate constant curve."""
    import curl_oracle as O
    from curl_amd import model
    g = torch.Generator().manual_seed(Kl * 100 + Kr)
    B, H, W = 2, 12, 20
    img = torch.rand(B, 3, H, W, generator=g)
    mask = torch.rand(B, 1, H, W, generator=g) > 0.2
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (3 * Kl, 3 * Kr, 4 * Kh))
    ref, rreg = O.curl_layer(img, mask.float(), L, R, Hk, 3 * Kl, 3 * Kr, 4 * Kh)
    layer = model.CURLLayer(3 * Kl, 3 * Kr, 4 * Kh).to(dev)
    # hand the layer WIDER parameter tensors, as GCURLNet's head might: it slices [:, :num_points] (model.py:153)
    pad = lambda t: torch.cat([t, torch.full((B, 5), 9.0)], 1).to(dev)  # noqa: E731
    with torch.no_grad():
        out, reg = layer(img.to(dev), mask.to(dev), pad(L), pad(R), pad(Hk))
    assert max_err(N(out), ref.numpy()) <= 1e-5
    np.testing.assert_allclose(N(reg), rreg.numpy(), rtol=3e-6, atol=1e-9)


@pytest.mark.parametrize("K", [17, 18, 19, 40, 256])
def test_exact_order_large_k_bitexact(ops, dev, K):
    """K-2 > 16 terms: the kernel follows ATen's cascade summation (flush every 16 terms)."""
    import curl_oracle as O
    g = torch.Generator().manual_seed(K)
    x = torch.rand(2, 3, 8, 16, generator=g) * 2 - 0.5
    C = torch.exp(torch.randn(2, K, generator=g) * 0.3)
    ref, rreg = O.apply_curve(x, C, torch.zeros(2), 1, 2)
    reg = torch.zeros(2, device=dev)
    out, reg = ops.apply_curve(x.to(dev), C.to(dev), reg, 1, 2)
    assert np.array_equal(N(out), ref.numpy())
    np.testing.assert_allclose(N(reg), rreg.numpy(), rtol=3e-6)
    raw = torch.randn(2, 3 * K, generator=g) * 0.2
    o2, r2 = ops.adjust_rgb(x.to(dev), raw.to(dev), flags=1)
    ref2, rr2 = O.adjust_rgb(x, raw)
    assert max_err(N(o2), ref2.numpy()) <= 2e-6
    np.testing.assert_allclose(N(r2), rr2.numpy(), rtol=3e-6)


def test_pwl_option_is_the_paper_curve(ops, dev):
    g = torch.Generator().manual_seed(3)
    K = 16
    x = torch.rand(1, 3, 4, 32, generator=g)
    C = torch.exp(torch.randn(1, K, generator=g) * 0.3)
    out, _ = ops.apply_curve(x.to(dev), C.to(dev), None, 0, 0, flags=2)
    xs = x[0, 0].numpy().astype(np.float64)
    scale = np.interp(xs * (K - 1), np.arange(K), C[0].numpy().astype(np.float64))
    assert np.abs(N(out)[0, 0] - np.clip(xs * scale, 0, 1)).max() < 1e-5


def test_in_place_and_streams(ops, dev, golden):
    """out may alias img (C ABI contract); work goes to torch's CURRENT stream."""
    from curl_amd import _lib
    c = golden("chain")
    img = T(c["img"], dev)
    L, R, H = (T(c["s01" + k], dev) for k in ("_L", "_R", "_H"))
    want, _ = ops.curl_layer_forward(img, None, L, R, H)
    x = img.clone()
    got, _ = ops.curl_layer_forward(x, None, L, R, H, out=x)
    assert got.data_ptr() == x.data_ptr() and torch.equal(got, want)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        on_side, _ = ops.curl_layer_forward(img, None, L, R, H)
    s.synchronize()
    assert torch.equal(on_side, want)
    assert _lib.load().curl_last_error() == b""


def test_layout_edges_golden(ops, dev, golden):
    g = golden("layout")
    out = ops.f32chw_to_u8hwc(T(g["egress_in"], dev))
    assert np.array_equal(N(out), g["egress_out"])
    inn = ops.u8hwc_to_f32chw(T(g["ingest_rgba"], dev))
    assert np.array_equal(N(inn), g["ingest_out"])
    rgb = ops.u8hwc_to_f32chw(T(np.ascontiguousarray(g["ingest_rgba"][..., :3]), dev))
    assert torch.equal(rgb, inn)
    # round trip on the 8-bit grid is the identity up to truncation of k/255*255
    u8 = torch.arange(0, 256, dtype=torch.uint8, device=dev).view(1, 16, 16, 1).expand(1, 16, 16, 3).contiguous()
    back = ops.f32chw_to_u8hwc(ops.u8hwc_to_f32chw(u8))
    assert (back.int() - u8.int()).abs().max().item() <= 1


def test_errors_on_device(ops, dev):
    x = torch.rand(1, 3, 4, 4, device=dev)
    with pytest.raises(TypeError):
        ops.rgb2lab(x.double())
    with pytest.raises(ValueError):
        ops.rgb2lab(x[:, :2])
    with pytest.raises(ValueError):
        ops.apply_curve(x, torch.ones(1, 1, device=dev), None, 0, 0)
    with pytest.raises(ValueError):
        ops.apply_curve(x, torch.ones(1, 16, device=dev), None, 3, 0)
    with pytest.raises(ValueError):
        ops.adjust_rgb(x, torch.zeros(1, 4, device=dev))  # torch.chunk(4, 3): two chunks -- the reference's unpacking fails
    with pytest.raises(ValueError):
        ops.curl_layer_forward(x, torch.ones(1, 1, 5, 4, device=dev), *(torch.zeros(1, n, device=dev) for n in (48, 48, 64)))
    with pytest.raises(ValueError):
        ops.apply_curve(x, torch.ones(1, 16, device=dev), None, 0, 0, flags=3)


# ------------------------------------------------------------------ BASELINE sizes: size-independent properties
@pytest.fixture(scope="module")
def big(dev):
    g = torch.Generator().manual_seed(2024)
    B, H, W = 4, 1000, 1500
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    mask = ((((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2) < 0.9)[None, None].expand(B, 1, H, W)
    return img, mask.contiguous().to(dev), L, R, Hk


def test_fullsize_batch_independence_and_permutation(ops, big):
    """1500x1000 frames: every image of a batch equals the same image processed alone (per-image knots reach
    the right pixels), and the op commutes with any permutation of the pixels (it is pointwise)."""
    img, mask, L, R, Hk = big
    out, reg = ops.curl_layer_forward(img, mask, L, R, Hk)
    for b in (0, 3):
        o1, r1 = ops.curl_layer_forward(img[b:b + 1], mask[b:b + 1], L[b:b + 1], R[b:b + 1], Hk[b:b + 1])
        assert torch.equal(o1[0], out[b]) and torch.equal(r1[0], reg[b])
    B, _, H, W = img.shape
    perm = torch.randperm(H * W, device=img.device)
    pimg = img.view(B, 3, -1)[:, :, perm].view(B, 3, H, W).contiguous()
    pmask = mask.view(B, 1, -1)[:, :, perm].view(B, 1, H, W).contiguous()
    pout, _ = ops.curl_layer_forward(pimg, pmask, L, R, Hk)
    assert torch.equal(pout.view(B, 3, -1), out.view(B, 3, -1)[:, :, perm])
    assert (out * (~mask) == 0).all() and out.min() >= 0 and out.max() <= 1


def test_fullsize_identity_curves(ops, big):
    """raw knots = 0 -> every knot = 1 -> scale == 1: adjust_* reduces to clamp(img, 0, 1) exactly."""
    img = big[0]
    z48 = torch.zeros(img.shape[0], 48, device=img.device)
    out, reg = ops.adjust_rgb(img * 1.5 - 0.25, z48)
    assert torch.equal(out, (img * 1.5 - 0.25).clamp(0, 1)) and (reg == 0).all()
    out, _ = ops.adjust_rgb(img * 1.5 - 0.25, z48, flags=1)
    assert torch.equal(out, (img * 1.5 - 0.25).clamp(0, 1))


def test_fullsize_vs_oracle_frames(ops, big):
    """1.5 Mpix frames against the oracle.  Whether every pixel can agree to 1e-5 depends on the knots: for
    some curve sets the reference's OWN float32 result is > 1e-5 from its float64 evaluation on ~1e-3 of the
    pixels (ill-conditioned dark / near-grey pixels; DESIGN.md 'Parity').  So the bar is anchored on float64
    truth: the HIP result may be no further from it than the reference's float32 result is (small margin),
    and its disagreement with the reference stays within the reference's own noise.  Plus > 115 dB PSNR."""
    import curl_oracle as O
    dev = big[0].device
    for b in (0, 1):
        img, mask, L, R, Hk = (t[b:b + 1].cpu() for t in big)
        mf = mask.float()
        ref, rreg = O.curl_layer(img, mf, L, R, Hk)
        r64, _ = O.curl_layer(img.double(), mf.double(), L.double(), R.double(), Hk.double())
        out, reg = ops.curl_layer_forward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev))
        out = out.cpu().double()
        ref_noise = (ref.double() - r64).abs()
        our_noise = (out - r64).abs()
        d = (out - ref.double()).abs()
        frac = lambda t, thr: float((t > thr).double().mean())  # noqa: E731
        assert float(our_noise.max()) <= 1.5 * float(ref_noise.max()) + 2e-6, b
        assert frac(our_noise, 1e-5) <= 1.25 * frac(ref_noise, 1e-5) + 1e-5, b
        assert frac(d, 1e-5) <= 1.25 * frac(ref_noise, 1e-5) + 1e-5, b
        assert float(d.max()) <= 2.0 * float(ref_noise.max()) + 2e-6, b
        mse = float((d ** 2).sum() / (3 * mf.sum()))
        assert 10 * np.log10(1.0 / mse) > 115, b
        np.testing.assert_allclose(N(reg), rreg.numpy(), rtol=2e-6)
        ls_ref, _ = O.lab_stage(img, mf, L)
        ls64, _ = O.lab_stage(img.double(), mf.double(), L.double())
        ls, _ = ops.lab_stage(img.to(dev), mask.to(dev), L.to(dev))
        ls = ls.cpu().double()
        assert float((ls - ls64).abs().max()) <= 1.5 * float((ls_ref.double() - ls64).abs().max()) + 2e-6, b
        assert frac((ls - ls_ref.double()).abs(), 1e-5) <= 1.25 * frac((ls_ref.double() - ls64).abs(), 1e-5) + 1e-5, b


def test_bs32_fullsize_round_trips_and_batch_independence(ops, dev):
    """BASELINE configs[2] at its real size, 32 x 1500x1000 (48 Mpix, 576 MB per tensor): colour-space round trips
    return the input, and the first / last image of the batch equal the same image processed alone."""
    torch.manual_seed(32)
    B, H, W = 32, 1000, 1500
    img = torch.rand(B, 3, H, W, device=dev)
    back = ops.lab2rgb(ops.rgb2lab(img))
    # the reference's forward matrix (OpenCV, colors.py:10-12) and inverse matrix (Lindbloom, colors.py:71-73) are
    # not exact inverses of each other: its own round trip is off by up to ~4e-5
    assert float((back - img).abs().max()) <= 1e-4
    del back
    hsv_back = ops.hsv2rgb(ops.rgb2hsv(img))
    # exact channel ties are not round-trip points of the reference: its hue terms ADD there (colors.py:221-224,
    # r == g > b comes out as 120 degrees); among 48M random pixels a handful tie
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    distinct = ((r != g) & (g != b) & (r != b)).unsqueeze(1)
    assert float(((hsv_back - img.clamp(1e-9, 1)).abs() * distinct).max()) <= 2e-6
    assert int((~distinct).sum()) < 100
    del hsv_back, distinct
    L, R, Hk = (torch.randn(B, n, device=dev) * 0.1 for n in (48, 48, 64))
    mask = torch.ones(B, 1, H, W, dtype=torch.bool, device=dev)
    out, reg = ops.curl_layer_forward(img, mask, L, R, Hk)
    assert out.min() >= 0 and out.max() <= 1 and torch.isfinite(out).all()
    for b in (0, 31):
        o1, r1 = ops.curl_layer_forward(img[b:b + 1], mask[b:b + 1], L[b:b + 1], R[b:b + 1], Hk[b:b + 1])
        assert torch.equal(o1[0], out[b]) and torch.equal(r1[0], reg[b])


def test_fullsize_trispace_properties(ops, dev):
    """1500x1000 frames through the row-tiled polynomial kernel: zero coefficients give sigmoid(0) = 0.5 everywhere,
    so the residual is one constant colour (the oracle's value for a single pixel); constant-only coefficients make
    the result independent of position; images of a batch equal the same images processed alone."""
    import curl_oracle as O
    torch.manual_seed(7)
    B, H, W = 3, 1000, 1500
    img = torch.rand(B, 3, H, W, device=dev)
    zero = torch.zeros(B, 3, 3, 126, device=dev)
    res = ops.trispace_forward(img, zero, residual_only=True)
    one = O.trispace_residual(torch.rand(1, 3, 1, 1), *(torch.zeros(1, 3, 126),) * 3)[0, :, 0, 0]
    assert float((res - one.to(dev).view(1, 3, 1, 1)).abs().max()) <= 2e-6
    c = torch.zeros(B, 3, 3, 126, device=dev)
    c[..., 0] = torch.randn(B, 3, 3, device=dev)          # constant terms only
    c[:, 0, :, 1:4] = torch.randn(B, 3, 3, device=dev) * 0.3  # RGB space: linear in the colour channels
    out = ops.trispace_forward(img, c)
    perm = torch.randperm(H * W, device=dev)
    pimg = img.view(B, 3, -1)[:, :, perm].view(B, 3, H, W).contiguous()
    pout = ops.trispace_forward(pimg, c)
    assert torch.equal(pout.view(B, 3, -1), out.view(B, 3, -1)[:, :, perm])  # no coordinate terms: pointwise
    full = torch.randn(B, 3, 3, 126, device=dev) * 0.2
    out = ops.trispace_forward(img, full)
    for b in (0, 2):
        assert torch.equal(ops.trispace_forward(img[b:b + 1], full[b:b + 1])[0], out[b])


def test_gcurlnet_forward(dev):
    """Encoder (stock PyTorch-ROCm) -> 160 knots -> fused HIP layer, against the oracle on the same knots."""
    import curl_oracle as O
    from curl_amd import model
    torch.manual_seed(0)
    net = model.GCURLNet(backbone=model.CurveEncoder(160, width=0.25, num_features=128), encoder_size=64).to(dev).eval()
    img = torch.rand(2, 3, 120, 160, device=dev)
    mask = torch.rand(2, 1, 120, 160, device=dev) > 0.1
    with torch.no_grad():
        out, reg = net(img, mask, None, None, None)
        knots = net.predict_knots(img)
    L, R, H = O.split_knots(knots.cpu())
    ref, rreg = O.curl_layer(img.cpu(), mask.cpu().float(), L, R, H)
    # north_star's 1e-5, except where the chain itself amplifies rounding noise: |err| <= max(1e-5, 2e-6 * S) per pixel
    # with S the float64 chain's input sensitivity (the bound of test_fullsize_exception_set_is_pinned_by_conditioning)
    S = O.input_sensitivity(img.cpu(), mask.cpu().float(), L, R, H)
    d = (out.cpu().double() - ref.double()).abs().amax(1)
    bound = torch.clamp(2e-6 * S, min=1e-5)
    assert int((d > bound).sum()) == 0, (float((d / bound).max()), float(d.max()))
    np.testing.assert_allclose(N(reg), rreg.numpy(), rtol=1e-5)


def test_compose_white_background(ops, dev):
    """infer.py:46-47 fused: x*mask + (1-mask), *255, truncate, CHW->HWC."""
    import curl_oracle as O
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 9, 13, generator=g)
    for mask in (torch.rand(2, 1, 9, 13, generator=g) > 0.4, torch.rand(2, 1, 9, 13, generator=g)):
        want = np.stack([O.f32chw_to_u8hwc(O.white_background(x[b], mask[b].float())) for b in range(2)])
        got = ops.compose_white_u8hwc(x.to(dev), mask.to(dev))
        np.testing.assert_array_equal(N(got), want)  # x*m rounded, + (1-m) rounded: the reference's two eager ops


def test_infer_cli_end_to_end(dev, tmp_path):
    """The reference's CLI flags (infer.py:14-17) on a synthetic RGBA file + mask: runs, writes a PNG whose
    background is white where the mask is 0 and whose foreground equals the layer applied at full resolution."""
    from PIL import Image
    from curl_amd import infer as cli
    g = np.random.default_rng(0)
    H, W = 200, 333  # not multiples of anything
    rgba = g.integers(0, 256, (H, W, 4), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    mask[40:160, 60:300] = 255
    Image.fromarray(rgba, "RGBA").save(tmp_path / "in.png")
    Image.fromarray(mask, "L").save(tmp_path / "mask.png")
    torch.manual_seed(0)
    for arch in ("trispace", "curl"):
        cli.infer(["--img_path", str(tmp_path / "in.png"), "--mask_path", str(tmp_path / "mask.png"),
                   "--model_file", "random", "--out_path", str(tmp_path / "out.png"), "--arch", arch])
        out = np.asarray(Image.open(tmp_path / "out.png"))
        assert out.shape == (H, W, 3) and out.dtype == np.uint8
        assert (out[mask == 0] == 255).all()
        assert out[mask == 255].std() > 0


# ------------------------------------------------------------------ polynomial path
@pytest.mark.parametrize("s,tol", [("s02", 1e-5), ("s1", 2e-5)])
def test_trispace_golden(ops, dev, golden, s, tol):
    """TriSpaceRegNet.generate_residual / generate_image (model.py:499-520) vs outputs of the reference's classes."""
    g = golden("poly")
    c = T(g[s + "_coeffs"], dev)
    for nm in ("img", "img8"):
        x = T(g[nm], dev)
        assert max_err(N(ops.trispace_forward(x, c, residual_only=True)), g[f"{s}_{nm}_residual"]) <= tol, nm
        assert max_err(N(ops.trispace_forward(x, c)), g[f"{s}_{nm}_image"]) <= tol, nm
    c35 = T(g[s + "_coeffs35"], dev)
    assert max_err(N(ops.trispace_forward(T(g["img"], dev), c35, residual_only=True)),
                   g[f"{s}_img_residual_nonspatial"]) <= tol


def test_poly_layer_golden(ops, dev, golden):
    g = golden("poly")
    assert max_err(N(ops.poly_layer(T(g["x5"], dev), T(g["c5"], dev))), g["mobile_poly"]) <= 3e-6
    assert max_err(N(ops.poly_layer(T(g["x5"], dev), T(g["c5"], dev))), g["channel_poly_d4v5"]) <= 3e-6
    assert max_err(N(ops.poly_layer(T(g["x3"], dev), T(g["c3"], dev))), g["channel_poly_d4v3"]) <= 3e-6


@pytest.mark.parametrize("shape", [(1, 7, 9), (2, 30, 50), (1, 64, 100), (1, 5, 1500), (2, 3, 1028), (1, 2, 4100),
                                   (1, 300, 1)])
def test_trispace_shapes_vs_oracle(ops, dev, shape):
    """Coordinates (x/W, y/H) must follow the pixel through scalar / float4 kernels, W % 4 != 0 included; rows wider
    than one block (1500 = 2 blocks of 192 lanes, 1028 = 257 float4 groups, 4100 = 5 blocks), one-pixel rows."""
    import curl_oracle as O
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    img = torch.rand(B, 3, H, W, generator=g)
    c = torch.randn(B, 3, 3, 126, generator=g) * 0.3
    c[:, :, :, 4] = 3.0   # make the x/W coefficient matter: a coordinate mix-up shows as a ramp error
    c[:, :, :, 5] = -2.0  # y/H
    ref = O.trispace_residual(img, c[:, 0], c[:, 1], c[:, 2])
    out = ops.trispace_forward(img.to(dev), c.to(dev), residual_only=True)
    assert max_err(N(out), ref.numpy()) <= 1e-5
    full = ops.trispace_forward(img.to(dev), c.to(dev))
    # the residual spans +-6 here (three spaces x +-2, large coordinate coefficients): same relative error, and
    # the clamped image inherits its ABSOLUTE size
    assert max_err(N(full), O.generate_image(img, ref).numpy()) <= 3e-5


def test_trispace_regnet_module(dev):
    """The mirror of the fork's live model: encoder -> [B,3,3,126] -> fused polynomial kernel, vs the oracle."""
    import curl_oracle as O
    from curl_amd import model
    torch.manual_seed(1)
    net = model.TriSpaceRegNet(spatial=True, is_train=True, polylayer=model.Deg4MobilePolyLayer(),
                               backbone=model.CurveEncoder(1, width=0.25, num_features=64), feature_width=64).to(dev).eval()
    img = torch.rand(2, 3, 48, 80, device=dev)
    mask = (torch.rand(2, 1, 48, 80, device=dev) > 0.2).float()
    big = torch.rand(2, 3, 100, 140, device=dev)
    with torch.no_grad():
        out = net(img, mask, big)
        R, L, H = net.generate_coefficients(img, mask)
        res = net.generate_residual(big, R, L, H)
    ref = O.trispace_residual(big.cpu(), R.cpu(), L.cpu(), H.cpu())
    assert max_err(N(res), ref.numpy()) <= 2e-5
    assert max_err(N(out), O.generate_image(big.cpu(), ref).numpy()) <= 2e-5
    assert set(k for k in net.state_dict() if k.startswith("backbone.classifier")) >= {
        "backbone.classifier.0.weight", "backbone.classifier.3.bias"}
    lay = model.Deg4MobilePolyLayer().to(dev)
    x5 = torch.rand(2, 5, 16, 16, device=dev)
    c = torch.randn(2, 3, 126, device=dev) * 0.3
    assert max_err(N(lay(x5, c)), O.deg4_mobile_poly_layer(x5.cpu(), c.cpu()).numpy()) <= 3e-6


def test_psnr_golden_and_oracle(dev, golden):
    """Masked PSNR (metric.py:35-68) vs the value the reference's PSNRMetric produced, and per image vs the oracle."""
    import curl_oracle as O
    from curl_amd import metric, ops as _ops
    c = golden("chain")
    a, b, m = T(c["psnr_a"], dev), T(c["psnr_b"], dev), T(c["mask_disk"], dev)
    for mm in (m, m.float()):
        v = metric.PSNRMetric()(a, b, mm)
        assert abs(float(v) - float(c["psnr_val"])) < 1e-4
    per = _ops.psnr_per_image(a, b, m)
    for i in range(a.shape[0]):
        want = O.psnr(torch.from_numpy(c["psnr_a"][i:i + 1]), torch.from_numpy(c["psnr_b"][i:i + 1]),
                      torch.from_numpy(c["mask_disk"][i:i + 1].astype(np.float32)))
        assert abs(float(per[i]) - float(want)) < 1e-4
    empty = torch.zeros_like(m)
    assert metric.PSNRMetric()(a, b, empty) is None  # 0/0 -> NaN for every image -> None (metric.py:67-68)


def test_psnr_special_cases_vs_oracle(dev):
    """What an evaluation run meets and random floats do not (metric.py:35-68): identical images (zero error: +inf, which
    nanmean keeps), an image whose mask is empty beside live ones (0/0 = NaN, which nanmean skips), a float mask with
    fractional values (the reference multiplies by it and divides by 3 * its sum), inputs outside [0, 1] (clamped first)."""
    import math
    import curl_oracle as O
    from curl_amd import metric, ops as _ops
    g = torch.Generator().manual_seed(31)
    a = torch.rand(3, 3, 33, 47, generator=g) * 1.4 - 0.2
    b = torch.rand(3, 3, 33, 47, generator=g) * 1.4 - 0.2
    b[1] = a[1]                                   # image 1: identical
    ones = torch.ones(3, 1, 33, 47)
    soft = torch.rand(3, 1, 33, 47, generator=g) * 0.9 + 0.05
    hole = ones.clone()
    hole[2] = 0.0                                 # image 2: nothing unmasked
    for m in (ones, ones.bool(), soft, hole, hole.bool()):
        per = _ops.psnr_per_image(a.to(dev), b.to(dev), m.to(dev)).cpu()
        for i in range(3):
            want = O.psnr(a[i:i + 1], b[i:i + 1], m[i:i + 1].float())
            if want is None:
                assert math.isnan(float(per[i])), (i, float(per[i]))
            elif math.isinf(float(want)):
                assert float(per[i]) == float(want)
            else:
                assert abs(float(per[i]) - float(want)) < 1e-4, (i, float(per[i]), float(want))
        got, ref = metric.PSNRMetric()(a.to(dev), b.to(dev), m.to(dev)), O.psnr(a, b, m.float())
        assert (got is None) == (ref is None) and (ref is None or float(got) == float(ref) or abs(float(got) - float(ref)) < 1e-4)


def test_hip_graph_capture_and_replay(ops, dev, golden):
    """The C ABI never allocates or synchronises, so a whole step (knot prep + fused layer) captures into a
    hipGraph and replays on new data in the same buffers."""
    c = golden("chain")
    L, R, H = (T(c["s01" + k], dev) for k in ("_L", "_R", "_H"))
    static_in = T(c["img"], dev).clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):  # warm-up outside capture (allocator pools)
        ops.curl_layer_forward(static_in, None, L, R, H)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out, static_reg = ops.curl_layer_forward(static_in, None, L, R, H)
    for name in ("img", "img8"):
        static_in.copy_(T(c[name], dev))
        graph.replay()
        torch.cuda.synchronize()
        assert max_err(N(static_out), c[f"s01_{name}_ones_out"]) <= 1e-5
        np.testing.assert_allclose(N(static_reg), c[f"s01_{name}_ones_reg"], rtol=2e-6)


@pytest.mark.parametrize("shape", [(2, 36, 52), (1, 33, 65), (2, 7, 9)])
@pytest.mark.parametrize("nc", [126, 35])
def test_trispace_u8hwc_fused_file_edge(ops, dev, shape, nc):
    """infer.py:35-47 in one launch on interleaved bytes == the same steps through the f32 entry points
    (ingest, trispace, white background, truncating egress), byte for byte; and within one grey level of the
    oracle's chain (truncation edges) on < 1 % of the bytes.  Shapes cover the dword path and the byte path."""
    import curl_oracle as O
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W + nc)
    img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    white = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, generator=g)
    white[:, : H // 3] = 0
    white[:, -H // 3:] = 255
    coeffs = torch.randn(B, 3, 3, nc, generator=g) * 0.2
    for wm in (None, white):
        got = ops.trispace_forward_u8hwc(img.to(dev), coeffs.to(dev), None if wm is None else wm.to(dev))
        x = ops.u8hwc_to_f32chw(img.to(dev))
        y = ops.trispace_forward(x, coeffs.to(dev))
        if wm is not None:
            m = (wm.float() / 255).unsqueeze(1).to(dev)  # divided on the CPU: torch's GPU div-by-scalar multiplies by 1/255
            y = y * m + (1 - m)
        steps = ops.f32chw_to_u8hwc(y)
        assert torch.equal(got, steps)
        xo = torch.stack([O.u8hwc_to_f32chw(img[b].numpy()) for b in range(B)])
        ro = O.trispace_residual(xo, coeffs[:, 0], coeffs[:, 1], coeffs[:, 2], spatial=(nc == 126))
        yo = O.generate_image(xo, ro)
        if wm is not None:
            yo = O.white_background(yo, (wm.float() / 255).unsqueeze(1))
        want = np.stack([O.f32chw_to_u8hwc(yo[b]) for b in range(B)])
        d = np.abs(N(got).astype(int) - want.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01


@pytest.mark.parametrize("shape,mask_kind", [((2, 36, 52), "bool"), ((1, 33, 65), "f32"), ((2, 7, 9), None),
                                             ((1, 40, 64), None)])
def test_curl_layer_u8hwc_fused_file_edge(ops, dev, shape, mask_kind):
    import curl_oracle as O
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    white = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, generator=g)
    L, R = torch.randn(B, 48, generator=g) * 0.1, torch.randn(B, 48, generator=g) * 0.1
    Hk = torch.randn(B, 64, generator=g) * 0.1
    mask = None
    if mask_kind == "bool":
        mask = torch.rand(B, 1, H, W, generator=g) > 0.3
    elif mask_kind == "f32":
        mask = torch.rand(B, 1, H, W, generator=g)
    md = None if mask is None else mask.to(dev)
    for wm in (None, white):
        got, reg = ops.curl_layer_forward_u8hwc(img.to(dev), md, L.to(dev), R.to(dev), Hk.to(dev),
                                                None if wm is None else wm.to(dev))
        x = ops.u8hwc_to_f32chw(img.to(dev))
        y, reg2 = ops.curl_layer_forward(x, md, L.to(dev), R.to(dev), Hk.to(dev))
        if wm is not None:
            m = (wm.float() / 255).unsqueeze(1).to(dev)
            y = y * m + (1 - m)
        assert torch.equal(got, ops.f32chw_to_u8hwc(y))
        assert torch.equal(reg, reg2)
        xo = torch.stack([O.u8hwc_to_f32chw(img[b].numpy()) for b in range(B)])
        mo = torch.ones(B, 1, H, W) if mask is None else mask.float()
        yo, _ = O.curl_layer(xo, mo, L, R, Hk)
        if wm is not None:
            yo = O.white_background(yo, (wm.float() / 255).unsqueeze(1))
        want = np.stack([O.f32chw_to_u8hwc(yo[b]) for b in range(B)])
        d = np.abs(N(got).astype(int) - want.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_u8hwc_entry_points_reject_bad_arguments(ops, dev):
    img = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device=dev)
    c = torch.zeros(1, 3, 3, 126, device=dev)
    with pytest.raises(ValueError):
        ops.trispace_forward_u8hwc(img.float(), c)
    with pytest.raises(ValueError):
        ops.trispace_forward_u8hwc(img, c, torch.zeros(1, 8, 9, dtype=torch.uint8, device=dev))
    with pytest.raises(RuntimeError):
        ops.trispace_forward_u8hwc(img.cpu(), c)


def test_layer_randomised_stress_vs_oracle(ops, dev):
    """Forty random cases -- shapes (scalar and float4 paths, ragged tiles), knot scales, mask kinds and densities,
    in-range / out-of-range / 8-bit-grid inputs (exact ties, exact zeros) -- against the oracle, with the bar of
    DESIGN.md 4: float32 noise of the reference's own size around the float64 evaluation.  On a few thousand pixels
    the maximum is one ill-conditioned pixel (a channel exactly 0, a near-grey), so the factor is looser here than
    in the 1.5 Mpix tests (3x instead of 1.5x); a wrong branch, mask or coordinate shows up as 1e-2, not 1e-5."""
    import curl_oracle as O
    rng = np.random.default_rng(2025)
    for case in range(40):
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 90))
        if case % 3 == 0:
            W = 4 * max(1, W // 4)
        g = torch.Generator().manual_seed(1000 + case)
        kind = case % 4
        img = torch.rand(B, 3, H, W, generator=g)
        if kind == 1:
            img = img * 1.3 - 0.15
        elif kind == 2:
            img = torch.round(img * 255) / 255
        elif kind == 3:
            img = torch.round(img * 7) / 7
        sigma = float(rng.choice([0.05, 0.1, 0.3]))
        L, R, Hk = (torch.randn(B, n, generator=g) * sigma for n in (48, 48, 64))
        density = float(rng.choice([0.0, 0.3, 0.9, 1.0]))
        mb = torch.rand(B, 1, H, W, generator=g) < density
        mask = [None, mb, mb.to(torch.uint8) * 255, torch.rand(B, 1, H, W, generator=g)][case % 4 if case % 8 < 4 else (case + 1) % 4]
        mref = torch.ones(B, 1, H, W) if mask is None else (mask != 0).float() if mask.dtype != torch.float32 else mask
        ref, rreg = O.curl_layer(img, mref, L, R, Hk)
        r64, _ = O.curl_layer(img.double(), mref.double(), L.double(), R.double(), Hk.double())
        out, reg = ops.curl_layer_forward(img.to(dev), None if mask is None else mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev))
        out = out.cpu()
        noise = float((ref.double() - r64).abs().max())
        assert float((out.double() - r64).abs().max()) <= 3.0 * noise + 5e-6, (case, B, H, W, kind, sigma)
        assert float((out - ref).abs().max()) <= 4.0 * noise + 5e-6, (case, B, H, W, kind, sigma)
        np.testing.assert_allclose(N(reg), rreg.numpy(), rtol=3e-6)


@pytest.mark.parametrize("K,mask_kind", [(16, None), (16, "bool"), (5, "f32"), (40, "bool")])
def test_lab_stage_pwl_fused_equals_the_separate_steps(ops, dev, K, mask_kind):
    """CURL_F_PWL on the fused Lab stage (knots in LDS, direct interval index, lerp) == RGB->Lab, adjust_lab with the
    same flag, x mask, Lab->RGB through the separate entry points; and it is NOT the affine form (the option is real)."""
    from curl_amd import _lib
    g = torch.Generator().manual_seed(K)
    B, H, W = 2, 37, 52
    img = (torch.rand(B, 3, H, W, generator=g) * 1.1 - 0.05).to(dev)
    L = (torch.randn(B, 3 * K, generator=g) * 0.3).to(dev)
    mask = None
    if mask_kind == "bool":
        mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
    elif mask_kind == "f32":
        mask = torch.rand(B, 1, H, W, generator=g).to(dev)
    fused, reg = ops.lab_stage(img, mask, L, flags=_lib.F_PWL)
    lab, reg2 = ops.adjust_lab(ops.rgb2lab(img), L, flags=_lib.F_PWL)
    if mask is not None:
        lab = lab * mask
    steps = ops.lab2rgb(lab.contiguous())
    assert float((fused - steps).abs().max()) <= 2e-6
    assert torch.equal(reg, reg2)
    affine, _ = ops.lab_stage(img, mask, L)
    assert float((fused - affine).abs().max()) > 1e-3


@pytest.mark.parametrize("Ks,mask_kind", [((16, 16, 16), None), ((16, 16, 16), "bool"), ((5, 9, 7), "f32")])
def test_layer_pwl_fused_equals_the_separate_steps(ops, dev, Ks, mask_kind):
    """CURL_F_PWL on the fused layer == the chain of model.py:150-170 run through the separate entry points with the
    same flag on every adjust_* (ten curves from LDS, three different knot counts)."""
    from curl_amd import _lib
    Kl, Kr, Kh = Ks
    g = torch.Generator().manual_seed(sum(Ks))
    B, H, W = 2, 33, 44
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.3).to(dev) for n in (3 * Kl, 3 * Kr, 4 * Kh))
    mask = None
    if mask_kind == "bool":
        mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
    elif mask_kind == "f32":
        mask = torch.rand(B, 1, H, W, generator=g).to(dev)
    m = 1.0 if mask is None else mask.float()
    fused, reg = ops.curl_layer_forward(img, mask, L, R, Hk, flags=_lib.F_PWL)
    x, r_lab = ops.adjust_lab(ops.rgb2lab(img), L, flags=_lib.F_PWL)
    x = ops.lab2rgb((x * m).contiguous())
    x, r_rgb = ops.adjust_rgb(x, R, flags=_lib.F_PWL)
    x = ops.rgb2hsv((x * m).contiguous())
    x, r_hsv = ops.adjust_hsv(x, Hk, flags=_lib.F_PWL)
    res = ops.hsv2rgb((x * m).contiguous())
    steps = (img + res).clamp(0, 1) * m
    assert float((fused - steps).abs().max()) <= 3e-6
    np.testing.assert_allclose(N(reg), N((r_rgb + r_lab) + r_hsv), rtol=1e-6)
    affine, _ = ops.curl_layer_forward(img, mask, L, R, Hk)
    assert float((fused - affine).abs().max()) > 1e-3


def test_empty_inputs_behave_like_the_eager_reference(ops, dev):
    """torch's eager ops take empty tensors in their stride: an empty batch or a zero-pixel image comes back empty,
    and the regulariser (a function of the knots alone) is still returned."""
    L, R, Hk = (torch.randn(2, n, device=dev) * 0.1 for n in (48, 48, 64))
    for shape in ((0, 3, 8, 8), (2, 3, 0, 5)):
        img = torch.empty(*shape, device=dev)
        B = shape[0]
        for conv in (ops.rgb2lab, ops.lab2rgb, ops.rgb2hsv, ops.hsv2rgb):
            assert conv(img).shape == img.shape
        out, reg = ops.curl_layer_forward(img, None, L[:B], R[:B], Hk[:B])
        assert out.shape == img.shape and reg.shape == (B,)
        out, reg_l = ops.lab_stage(img, torch.ones(B, 1, *shape[2:], dtype=torch.bool, device=dev), L[:B])
        assert out.shape == img.shape and reg_l.shape == (B,)
        out, reg_r = ops.adjust_rgb(img, R[:B])
        assert out.shape == img.shape
        if B:
            full, reg_full = ops.adjust_rgb(torch.rand(B, 3, 4, 4, device=dev), R[:B])
            assert torch.equal(reg_r, reg_full)  # the same number as with pixels
        c = torch.zeros(B, 3, 3, 126, device=dev)
        assert ops.trispace_forward(img, c).shape == img.shape


def test_polyregnet_module(dev):
    """model.py:418-436: sigmoid(degree-4 polynomial in the 3 colour channels) * mask, against the oracle's layer."""
    import curl_oracle as O
    from curl_amd import model
    torch.manual_seed(4)
    net = model.PolyRegNet(backbone=model.CurveEncoder(1, width=0.25, num_features=128), feature_width=128).to(dev).eval()
    img = torch.rand(2, 3, 40, 56, device=dev)
    mask = (torch.rand(2, 1, 40, 56, device=dev) > 0.2).float()
    with torch.no_grad():
        out = net(img, mask)
        coeffs = net.backbone(img).reshape(2, 3, 35)
    ref = torch.sigmoid(O.channel_poly_layer(img.cpu(), coeffs.cpu(), degree=4)) * mask.cpu()
    assert max_err(N(out), ref.numpy()) <= 1e-5


def test_new_entry_points_reject_bad_arguments(ops, dev):
    """Argument errors of the later entry points surface as ValueError before any launch."""
    from curl_amd import _lib
    a = torch.rand(1, 1, 64, 64, device=dev)
    with pytest.raises(ValueError):
        ops.msssim_stats(a, a, window_size=13)          # window > 11
    with pytest.raises(ValueError):
        ops.msssim_stats(a, a, window_size=4)           # even window
    with pytest.raises(ValueError):
        ops.msssim_stats(a[:, :, :20, :20].contiguous(), a[:, :, :20, :20].contiguous())  # < 32 px: five levels impossible
    with pytest.raises(RuntimeError):
        ops.msssim_stats(a, a[:, :, :32].contiguous())  # shape mismatch, like metric.py:181-183
    img = torch.rand(1, 3, 8, 8, device=dev)
    c = torch.zeros(1, 3, 3, 126, device=dev)
    with pytest.raises(ValueError):
        ops.trispace_backward(img, c[:, :, :, :100].contiguous(), img)    # neither 126 nor 35 coefficients
    u8 = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device=dev)
    L, R, H = (torch.zeros(1, n, device=dev) for n in (48, 48, 64))
    lib = _lib.load()
    ws, nbytes = ops._workspace(1, 160, dev)
    out, reg = torch.empty_like(u8), torch.empty(1, device=dev)
    rc = lib.curl_layer_fwd_u8hwc(u8.data_ptr(), 0, 0, L.data_ptr(), R.data_ptr(), H.data_ptr(), 0, out.data_ptr(),
                                  reg.data_ptr(), ws.data_ptr(), nbytes, 1, 8, 8, 16, 16, 16, _lib.F_PWL, ops._stream(u8))
    assert rc < 0 and b"flag" in lib.curl_last_error()   # the byte entry points take no flags
    # and the fused PWL forms need every curve to fit the LDS table
    with pytest.raises(ValueError):
        ops.lab_stage(img, None, torch.zeros(1, 3 * 300, device=dev), flags=_lib.F_PWL)


# ------------------------------------------------------------------ round 2: row slabs, exception set, configs[0]
@pytest.mark.parametrize("shape,rows", [((2, 16, 24), (4, 12)), ((1, 7, 9), (2, 5)), ((2, 10, 6), (0, 5)), ((2, 10, 6), (5, 10)),
                                        ((1, 33, 64), (32, 33)), ((3, 12, 20), (0, 12))])
@pytest.mark.parametrize("mask_kind", [None, "bool", "f32"])
def test_layer_rows_equal_the_same_rows_of_the_whole_image(ops, dev, shape, rows, mask_kind):
    """curl_layer_fwd_slab_f32 (split-pixels layout): rows [r0, r1) of the full tensors, in place, bit-identical to
    those rows of the whole-image call; nothing outside the slab is written.  Vector and scalar kernels."""
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
    mask = None
    if mask_kind == "bool":
        mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
    elif mask_kind == "f32":
        mask = torch.rand(B, 1, H, W, generator=g).to(dev)
    full, reg = ops.curl_layer_forward(img, mask, L, R, Hk)
    out = torch.full_like(img, -7.0)
    got, reg2 = ops.curl_layer_forward_rows(img, mask, L, R, Hk, rows, out)
    r0, r1 = rows
    assert got is out and torch.equal(out[:, :, r0:r1], full[:, :, r0:r1]) and torch.equal(reg2, reg)
    untouched = torch.ones(H, dtype=torch.bool)
    untouched[r0:r1] = False
    assert (out[:, :, untouched.to(dev)] == -7.0).all()
    # in place on the image itself
    work = img.clone()
    ops.curl_layer_forward_rows(work, mask, L, R, Hk, rows, work)
    assert torch.equal(work[:, :, r0:r1], full[:, :, r0:r1]) and torch.equal(work[:, :, untouched.to(dev)], img[:, :, untouched.to(dev)])


@pytest.mark.parametrize("nc", [126, 35])
@pytest.mark.parametrize("shape,rows", [((2, 30, 52), (7, 19)), ((1, 9, 7), (3, 9)), ((2, 16, 1500), (8, 16))])
def test_trispace_rows_keep_the_full_image_coordinates(ops, dev, nc, shape, rows):
    """The polynomial's y = row / H must be the FULL image's (model.py:487-497): the slab entry equals the rows of the
    whole-image call bit for bit, while slicing the tensor first (what apply_row_slab did in round 1) does not."""
    B, H, W = shape
    g = torch.Generator().manual_seed(nc + H)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    coeffs = (torch.randn(B, 3, 3, nc, generator=g) * 0.2).to(dev)
    full = ops.trispace_forward(img, coeffs)
    out = torch.full_like(img, -7.0)
    ops.trispace_forward_rows(img, coeffs, rows, out)
    r0, r1 = rows
    assert torch.equal(out[:, :, r0:r1], full[:, :, r0:r1])
    assert (out[:, :, :r0] == -7.0).all() and (out[:, :, r1:] == -7.0).all()
    if nc == 126 and r0 > 0:
        sliced = ops.trispace_forward(img[:, :, r0:r1].contiguous(), coeffs)
        assert not torch.equal(sliced, full[:, :, r0:r1])  # the renumbered rows of the round-1 route
    from curl_amd import shard
    both = torch.zeros_like(img)
    for rank in range(3):  # three "ranks" tile the image
        o, (a, b) = shard.apply_row_slab_trispace(img, coeffs, rank, 3)
        both[:, :, a:b] = o[:, :, a:b]
    assert torch.equal(both, full)


def test_row_slab_arguments_are_checked(ops, dev):
    img = torch.rand(1, 3, 8, 8, device=dev)
    L, R, Hk = (torch.zeros(1, n, device=dev) for n in (48, 48, 64))
    out = torch.empty_like(img)
    for rows in ((0, 0), (-1, 4), (4, 9), (5, 3)):
        with pytest.raises(ValueError):
            ops.curl_layer_forward_rows(img, None, L, R, Hk, rows, out)
    with pytest.raises(ValueError):
        ops.curl_layer_forward_rows(img, None, L, R, Hk, (0, 4), torch.empty(1, 3, 4, 8, device=dev))
    from curl_amd import _lib
    lib = _lib.load()
    assert lib.curl_layer_fwd_slab_f32(img.data_ptr(), 0, 0, L.data_ptr(), R.data_ptr(), Hk.data_ptr(), out.data_ptr(), 0,
                                       0, 0, 1, 8, 8, 6, 4, 16, 16, 16, 0, 0) == -2  # rows leave the image
    assert b"slab" in lib.curl_last_error()


def _input_sensitivity(O, img, mf, L, R, Hk, r64, h=1e-6):
    return O.input_sensitivity(img, mf, L, R, Hk, r64=r64, h=h)


def test_fullsize_exception_set_is_pinned_by_conditioning(ops, big):
    """WHICH pixels may differ from the reference by more than 1e-5, and by how much.

    The chain amplifies rounding noise where it is ill-conditioned (out-of-gamut Lab->RGB values met unclamped by
    the R curve, gamma slope 12.92 at black, hue ~ 1/(max-min)): there the reference's own float32 result moves by
    more than 1e-5 when torch merely vectorises differently (a pixel evaluated alone vs inside the frame), so its
    realised float32 noise at a pixel is not a stable yardstick.  The stable one is the chain's input sensitivity
    S = max |d out / d in| of the float64 evaluation.  Pinned here, per pixel:
        |HIP - ref32| <= max(1e-5, 2e-6 * S)      and      |HIP - ref64| <= max(1e-5, 2e-6 * S)
    i.e. a pixel may exceed 1e-5 only if S > 5 (2 % of frame 0, none of frame 1), and then by no more than 2e-6 per
    unit of amplification -- the bound the reference's float32 evaluation itself needs (measured 1.8e-6 * S).  On the
    well-conditioned frame 1 the strict 1e-5 holds on every pixel."""
    import curl_oracle as O
    dev = big[0].device
    for b, strict in ((1, True), (0, False)):
        img, mask, L, R, Hk = (t[b:b + 1].cpu() for t in big)
        mf = mask.float()
        ref, _ = O.curl_layer(img, mf, L, R, Hk)
        r64, _ = O.curl_layer(img.double(), mf.double(), L.double(), R.double(), Hk.double())
        out, _ = ops.curl_layer_forward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev))
        out = out.cpu().double()
        d = (out - ref.double()).abs().amax(1)
        ours = (out - r64).abs().amax(1)
        if strict:
            assert float(d.max()) <= 1e-5 and float(ours.max()) <= 1e-5
            continue
        S = _input_sensitivity(O, img, mf, L, R, Hk, r64)
        bound = torch.clamp(2e-6 * S, min=1e-5)
        assert int((d > bound).sum()) == 0, (int((d > bound).sum()), float((d / bound).max()))
        assert int((ours > bound).sum()) == 0, (int((ours > bound).sum()), float((ours / bound).max()))
        over = d > 1e-5
        assert 0 < int(over.sum()) < 2e-3 * over.numel()      # the set exists on this frame, and is small
        assert float(S[over].min()) > 5.0                       # ... and consists of ill-conditioned pixels only
        assert float(((S > 5.0) & (mask[:, 0].cpu())).double().mean()) < 0.08   # which are few
        # the reference's own float32 evaluation needs the same allowance on this frame
        noise = (ref.double() - r64).abs().amax(1)
        assert float((noise > 1e-5).double().mean()) > 0.5 * float(over.double().mean())


def test_wider_knots_sigma03_conditioning_bound(ops, dev):
    """the 2e-6 * S constant of the pinned bound is a statement about the benchmark's knot spread (sigma 0.1).
    At sigma 0.3 (the worst case of tools/parity_stress.py: seed 0, 512x768, no mask) the reference's OWN float32 result
    needs 2.5e-6 * S against its float64 evaluation and the kernel 2.8e-6 * S against the reference: the constant that
    holds for both is 3e-6, and the kernel stays within 25 % of the reference's own noise measured the same way."""
    import curl_oracle as O
    sigma, seed, H, W = 0.3, 0, 512, 768
    g = torch.Generator().manual_seed(1000 * seed + int(sigma * 100))
    img = torch.rand(1, 3, H, W, generator=g)
    L, R, Hk = (torch.randn(1, n, generator=g) * sigma for n in (48, 48, 64))
    mask = torch.ones(1, 1, H, W)
    ref, _ = O.curl_layer(img, mask, L, R, Hk)
    r64, _ = O.curl_layer(img.double(), mask.double(), L.double(), R.double(), Hk.double())
    S = O.input_sensitivity(img, mask, L, R, Hk, r64=r64)
    out, _ = ops.curl_layer_forward(img.to(dev), None, L.to(dev), R.to(dev), Hk.to(dev))
    d = (out.cpu().double() - ref.double()).abs().amax(1)
    noise = (ref.double() - r64).abs().amax(1)
    bound = torch.clamp(3e-6 * S, min=1e-5)
    assert int((d > bound).sum()) == 0, float((d / bound).max())
    assert float((d / bound).max()) <= 1.25 * float((noise / bound).max()) + 0.05


def test_config1_real_image_pixels_on_the_hip_path(ops, dev, golden):
    """BASELINE configs[0]: pixels of the reference's example image (adobe5k_dpe/curl_example_test_input, 16x16 corner
    of the 256x256 centre crop) with the golden knots, HIP layer vs the reference-generated output; and the byte
    output through the fused uint8 entry point."""
    c = golden("config1")
    x = T(c["in_corner"], dev)[None]
    L, R, Hk = T(c["L"], dev), T(c["R"], dev), T(c["H"], dev)
    out, reg = ops.curl_layer_forward(x, torch.ones(1, 1, 16, 16, dtype=torch.bool, device=dev), L, R, Hk)
    assert max_err(N(out)[0], c["out_corner"]) <= 1e-5
    np.testing.assert_allclose(N(reg), c["reg"], rtol=2e-6)
    u8 = ops.f32chw_to_u8hwc(out)[0]
    assert (N(u8).astype(int) - c["out_u8_corner"].astype(int)).__abs__().max() <= 1
    out_none, _ = ops.curl_layer_forward(x, None, L, R, Hk)
    assert torch.equal(out_none, out)


@pytest.mark.parametrize("shape", [(2, 40, 60), (1, 33, 65), (2, 7, 9)])
def test_tuning_flags_do_not_change_results(ops, dev, shape):
    """CURL_F_TUNE_*: float4 groups per lane (1, 2, 4), threads per workgroup (256, 128, 64) and plain instead of
    non-temporal accesses are scheduling choices -- every combination must give bit-identical images."""
    from curl_amd import _lib
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
    ref, reg = ops.curl_layer_forward(img, mask, L, R, Hk)
    ls0, _ = ops.lab_stage(img, mask, L)
    rgb0, _ = ops.adjust_rgb(img, R)
    for u in (1, 2, 4):
        for b in (0, 1, 2):
            for nt in (0, _lib.F_TUNE_NO_NT):
                fl = (u << _lib.F_TUNE_UNROLL_SHIFT) | (b << _lib.F_TUNE_BLOCK_SHIFT) | nt
                out, r2 = ops.curl_layer_forward(img, mask, L, R, Hk, flags=fl)
                assert torch.equal(out, ref) and torch.equal(r2, reg), hex(fl)
                assert torch.equal(ops.lab_stage(img, mask, L, flags=fl)[0], ls0), hex(fl)
                assert torch.equal(ops.adjust_rgb(img, R, flags=fl)[0], rgb0), hex(fl)
    with pytest.raises(ValueError):
        ops.curl_layer_forward(img, mask, L, R, Hk, flags=3 << _lib.F_TUNE_BLOCK_SHIFT)


def _misaligned(t):
    """The same values in a tensor whose storage starts one element past an allocation boundary (4 bytes for float32, 1 byte
    for uint8 / bool): contiguous, but not 16-byte aligned -> the kernels take their one-pixel-per-lane path."""
    flat = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    view = flat[1:].view(t.shape)
    view.copy_(t)
    assert view.data_ptr() % (16 if t.dtype == torch.float32 else 4) != 0 and view.is_contiguous()
    return view


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 36, 40), (3, 37, 41), (1, 64, 256)])
def test_vector_and_scalar_paths_of_the_f3_and_edge_kernels_agree(ops, dev, shape):
    """PSNR, the CURLLoss terms (forward, backward) and the byte edges run 4 pixels per lane when the plane size is a multiple
    of 4 and every base is aligned, one pixel per lane otherwise: both paths against each other (per-pixel outputs bit for
    bit; sums to float32 rounding -- their blocks cover different pixel sets) and against the oracle / torch."""
    import curl_oracle as O
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    a = torch.rand(B, 3, H, W, generator=g).to(dev)
    b = (a + 0.1 * torch.randn(B, 3, H, W, generator=g).to(dev)).clamp(0, 1)
    m = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
    w4 = torch.tensor([0.7, 1.3, 0.9, 1.1], device=dev)
    gL = torch.rand(B, 1, H, W, generator=g).to(dev)
    aligned = (H * W) % 4 == 0
    variants = [(a, b, m, gL)]
    if aligned:  # the same call through the other code path
        variants.append((_misaligned(a), _misaligned(b), _misaligned(m), _misaligned(gL)))
    res = []
    for (x, y, mk, gl) in variants:
        for mask in (mk, mk.float()):
            sums, Lp, Lt = ops.loss_term_sums(x, y, mask)
            res.append(dict(psnr=ops.psnr_per_image(x, y, mask), sums=sums, Lp=Lp, Lt=Lt,
                            gb=ops.loss_terms_backward(x, y, mask, w4, gl), u8=ops.f32chw_to_u8hwc(x),
                            white=ops.compose_white_u8hwc(x, mask)))
    # against the oracle / torch
    r0 = res[0]
    for i in range(B):
        want = O.psnr(a[i:i + 1].cpu(), b[i:i + 1].cpu(), m[i:i + 1].float().cpu())
        assert abs(float(r0["psnr"][i]) - float(want)) < 1e-4
    terms = O.curl_loss_terms(a.cpu(), b.cpu(), m.cpu())
    got = r0["sums"].sum(0).cpu()
    n = 3.0 * float(got[4])  # model.py:92: the sums are normalised by channels x unmasked pixels
    assert abs(float(got[4]) - float(m.sum())) == 0.0
    for k in (0, 2, 3):
        assert abs(float(got[k]) / n - float(terms[k])) <= 3e-6, k
    assert float((r0["Lp"].cpu() - terms[4]).abs().max()) <= 1e-6 and float((r0["Lt"].cpu() - terms[5]).abs().max()) <= 1e-6
    assert torch.equal(r0["u8"].cpu(), (a.cpu() * 255).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1))
    back = ops.u8hwc_to_f32chw(r0["u8"])
    assert torch.equal(back.cpu(), r0["u8"].cpu().permute(0, 3, 1, 2).float() / 255)
    if aligned:
        assert torch.equal(ops.u8hwc_to_f32chw(_misaligned(r0["u8"])), back)
    # the paths against each other
    for r in res[1:]:
        for k in ("Lp", "Lt", "u8", "white"):
            assert torch.equal(r[k], r0[k]), k
        # the gradient chains are compiled separately per path (other fma contractions): equal to float32 rounding
        assert float((r["gb"] - r0["gb"]).abs().max() / r0["gb"].abs().max()) <= 2e-6
        assert float((r["sums"] - r0["sums"]).abs().max() / r0["sums"].abs().max()) <= 1e-6
        assert float((r["psnr"] - r0["psnr"]).abs().max()) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("nvar,shape", [(5, (2, 36, 40)), (5, (1, 37, 41)), (3, (2, 20, 64)), (3, (1, 1, 1)), (5, (1, 3, 1030))])
def test_poly_layer_paths_vs_twin(ops, dev, twin, nvar, shape):
    """The stand-alone polynomial layer: 4 scalar Horner chains per lane when planes are 4-pixel multiples and aligned, one
    otherwise -- both against the host twin of the same arithmetic (bit for bit between the two GPU paths)."""
    B, H, W = shape
    g = torch.Generator().manual_seed(nvar * 100 + H + W)
    x = torch.rand(B, nvar, H, W, generator=g)
    c = torch.randn(B, 3, 126 if nvar == 5 else 35, generator=g) * 0.3
    want = twin.poly_layer(x.numpy(), c.numpy())
    got = ops.poly_layer(x.to(dev), c.to(dev))
    assert max_err(N(got), want) <= 3e-6
    if (H * W) % 4 == 0:
        assert torch.equal(ops.poly_layer(_misaligned(x.to(dev)), c.to(dev)), got)


# ------------------------------------------------------------------ fused HSV stage (model.py:163-169)
@pytest.mark.parametrize("sig,tol", [("s01", 3e-6), ("s05", 1e-5)])
def test_hsv_stage_golden(ops, dev, golden, sig, tol):
    """curl_hsv_stage_f32 -- RGB -> HSV -> adjust_hsv -> *mask -> RGB in one pass -- against the reference's own
    RGB2HSV / apply_curve x4 / HSV2RGB chained as model.py:163-169 chains them (make_golden_hsv_stage.py): unit-range,
    8-bit-grid (exact channel ties, grey, black, white, primaries) and out-of-range inputs; no / bool / float masks."""
    c = golden("hsv_stage")
    H = T(c[sig + "_H"], dev)
    for inn in ("img", "img8", "wide"):
        for mk in ("ones", "holes", "disk", "soft"):
            key = f"{sig}_{inn}_{mk}"
            for m in _mask_variants(c, mk, dev):
                out, reg = ops.hsv_stage(T(c[inn], dev), m, H)
                assert max_err(N(out), c[key + "_out"]) <= tol, (key, None if m is None else m.dtype)
                np.testing.assert_allclose(N(reg), c[key + "_reg"], rtol=2e-6)


def test_hsv_stage_is_the_layers_third_stage(ops, dev, golden):
    """The reference's chain fixture holds the HSV stage's INPUT (after_rgb_stage) for the all-ones mask and the layer's
    output: clamp(img + hsv_stage(after_rgb_stage)) must be the layer (model.py:169-170), to 2e-6; shapes that take the
    scalar path, in-place output and a masked-out wave (exactly 0 = hsv2rgb(0,0,0)) included."""
    import curl_oracle as O
    c = golden("chain")
    H = T(c["s01_H"], dev)
    res, reg = ops.hsv_stage(T(c["s01_img_ones_rgb_stage"], dev), None, H)
    np.testing.assert_allclose(N(reg), c["s01_img_ones_reg_hsv"], rtol=2e-6)
    assert max_err(N((T(c["img"], dev) + res).clamp(0, 1)), c["s01_img_ones_out"]) <= 2e-6
    g = torch.Generator().manual_seed(5)
    for B, Hh, W in ((1, 7, 9), (2, 33, 65), (1, 64, 1024)):
        img = torch.rand(B, 3, Hh, W, generator=g) * 1.4 - 0.2
        Hk = torch.randn(B, 64, generator=g) * 0.1
        mask = torch.rand(B, 1, Hh, W, generator=g) > 0.3
        mask[:, :, : Hh // 2] = False  # whole wavefronts masked out
        want, wreg = O.hsv_stage(img, mask.float(), Hk)
        for m in (mask.to(dev), mask.float().to(dev)):
            x = img.to(dev)
            got, reg = ops.hsv_stage(x, m, Hk.to(dev))
            assert max_err(N(got), want.numpy()) <= 3e-6, (B, Hh, W, m.dtype)
            assert (got[:, :, : Hh // 2] == 0).all()
            ops.hsv_stage(x, m, Hk.to(dev), out=x)  # in place: every pixel is read before it is written
            assert torch.equal(x, got)
        np.testing.assert_allclose(N(reg), wreg.numpy(), rtol=2e-6)


# ------------------------------------------------------------------ validation mode of the fused stages
def test_fused_stages_exact_order_validation_mode(ops, dev, golden, big):
    """CURL_F_EXACT_ORDER on curl_lab_stage_f32 / curl_layer_fwd_f32 (affine form as the default AND a
    validation mode): every curve as the reference's term-by-term float32 sum (curves.py:31-32, ATen's cascade order),
    every `* mask` executed, floors and refined reciprocals kept.  It must hold the golden bar like the default, agree
    with the default to the size of the reference's own summation noise, and -- being the reference's arithmetic but for
    the three hardware transcendentals -- may not be FURTHER from the reference than the default on a full-size frame."""
    import curl_oracle as O
    from curl_amd._lib import F_EXACT_ORDER
    c = golden("chain")
    L, R, H = (T(c["s01" + k], dev) for k in ("_L", "_R", "_H"))
    for mk in ("ones", "holes", "soft"):
        key = f"s01_img_{mk}"
        for m in _mask_variants(c, mk, dev):
            out, reg = ops.curl_layer_forward(T(c["img"], dev), m, L, R, H, flags=F_EXACT_ORDER)
            assert max_err(N(out), c[key + "_out"]) <= 1e-5, key
            np.testing.assert_allclose(N(reg), c[key + "_reg"], rtol=2e-6)
            ls, _ = ops.lab_stage(T(c["img"], dev), m, L, flags=F_EXACT_ORDER)
            assert max_err(N(ls), c[key + "_lab_stage"]) <= 1e-5, key
            dflt, _ = ops.curl_layer_forward(T(c["img"], dev), m, L, R, H)
            assert max_err(N(out), N(dflt)) <= 1e-5
    with pytest.raises(ValueError):
        ops.curl_layer_forward(T(c["img"], dev), None, L, R, H, flags=F_EXACT_ORDER | 2)  # exclusive with CURL_F_PWL
    # full size: frame 1 of the conditioning test (well conditioned: strict 1e-5) and frame 0 (exception set)
    img, mask, Lb, Rb, Hb = (t[:2] for t in big)
    ex, _ = ops.curl_layer_forward(img, mask, Lb, Rb, Hb, flags=F_EXACT_ORDER)
    df, _ = ops.curl_layer_forward(img, mask, Lb, Rb, Hb)
    for b in (0, 1):
        ref, _ = O.curl_layer(img[b:b + 1].cpu(), mask[b:b + 1].cpu().float(), Lb[b:b + 1].cpu(), Rb[b:b + 1].cpu(), Hb[b:b + 1].cpu())
        d_ex = (ex[b:b + 1].cpu().double() - ref.double()).abs()
        d_df = (df[b:b + 1].cpu().double() - ref.double()).abs()
        n_ex, n_df = int((d_ex > 1e-5).sum()), int((d_df > 1e-5).sum())
        print(f"frame {b}: px*ch over 1e-5 -- exact-order {n_ex} (max {float(d_ex.max()):.2e}), default {n_df} (max {float(d_df.max()):.2e})")
        if b == 1:
            assert n_ex == 0 and n_df == 0
        else:
            assert n_ex <= 1.5 * n_df + 50


# ------------------------------------------------------------------ torch.chunk's uneven split (curves.py:53,105,152)
def test_uneven_torch_chunk_split(ops, dev):
    """A parameter count that does not divide by the number of curves: torch.chunk hands the first curves ceil(N/n) knots
    and the last one the remainder (47 -> 16, 16, 15; 62 -> 16, 16, 16, 14).  The affine forms and the backward follow
    it (CURL_K_UNEVEN); counts torch.chunk would split into FEWER chunks raise, as the reference's unpacking does."""
    import curl_oracle as O
    g = torch.Generator().manual_seed(47)
    B, H, W = 2, 20, 36
    img = torch.rand(B, 3, H, W, generator=g)
    mask = torch.rand(B, 1, H, W, generator=g) > 0.2
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (47, 46, 62))
    for name, fn, k in (("rgb", "adjust_rgb", R), ("lab", "adjust_lab", L), ("hsv", "adjust_hsv", Hk)):
        out, reg = getattr(ops, fn)(img.to(dev), k.to(dev))
        want, wreg = getattr(O, fn)(img, k)
        assert max_err(N(out), want.numpy()) <= 2e-6, name
        np.testing.assert_allclose(N(reg), wreg.numpy(), rtol=2e-6)
        with pytest.raises(ValueError):
            getattr(ops, fn)(img.to(dev), k.to(dev), flags=1)  # exact-order mode: equal curves only
    out, reg = ops.curl_layer_forward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev))
    want, wreg = O.curl_layer(img, mask.float(), L, R, Hk, 47, 46, 62)
    assert max_err(N(out), want.numpy()) <= 1e-5
    np.testing.assert_allclose(N(reg), wreg.numpy(), rtol=2e-6)
    ls, _ = ops.lab_stage(img.to(dev), mask.to(dev), L.to(dev))
    assert max_err(N(ls), O.lab_stage(img, mask.float(), L, 47)[0].numpy()) <= 1e-5
    hs, _ = ops.hsv_stage(img.to(dev), mask.to(dev), Hk.to(dev))
    assert max_err(N(hs), O.hsv_stage(img, mask.float(), Hk, 62)[0].numpy()) <= 3e-6
    # backward: knot gradients land on the right knots of the shorter last curve
    w = torch.randn(B, 3, H, W, generator=g)
    wr = torch.rand(B, generator=g)
    x = img.clone().requires_grad_(True)
    Lg, Rg, Hg = (t.clone().requires_grad_(True) for t in (L, R, Hk))
    o, r = O.curl_layer(x, mask.float(), Lg, Rg, Hg, 47, 46, 62)
    ((o * w).sum() + (r * wr).sum()).backward()
    gi, gL, gR, gH = ops.curl_layer_backward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev), w.to(dev), wr.to(dev))
    for a, b in ((gL, Lg.grad), (gR, Rg.grad), (gH, Hg.grad)):
        assert float((a.cpu() - b).abs().max() / b.abs().max()) <= 1e-3
    with pytest.raises(ValueError):
        ops.adjust_rgb(img.to(dev), torch.zeros(B, 4, device=dev))   # torch.chunk(4, 3) gives two chunks of 2
    with pytest.raises(ValueError):
        ops.adjust_hsv(img.to(dev), torch.zeros(B, 13, device=dev))  # 4, 4, 4, 1: a one-knot curve


def test_xcd_contiguous_tile_mapping_is_the_same_result(ops, dev):
    """CURL_F_TUNE_XCD = 2 (each XCD walks one contiguous eighth of an image's tiles; a lost experiment kept as a tuning
    bit) only renumbers which workgroup takes which tile: identical bits, ragged tile counts and the
    padding workgroups of the last eighth included."""
    XCD = 2 << 13
    g = torch.Generator().manual_seed(8)
    for B, H, W in ((2, 300, 500), (1, 257, 1021), (3, 512, 512)):   # 147 / 257 (scalar path) / 256 tiles per image
        img = torch.rand(B, 3, H, W, generator=g).to(dev)
        mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
        L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
        a, ra = ops.curl_layer_forward(img, mask, L, R, Hk)
        b, rb = ops.curl_layer_forward(img, mask, L, R, Hk, flags=XCD)
        assert torch.equal(a, b) and torch.equal(ra, rb), (B, H, W)
        assert torch.equal(ops.lab_stage(img, mask, L)[0], ops.lab_stage(img, mask, L, flags=XCD)[0])
        assert torch.equal(ops.rgb2lab(img), ops.rgb2lab(img, flags=XCD))
    with pytest.raises(ValueError):
        ops.rgb2lab(img, flags=3 << 13)


def test_resident_workgroup_cap_is_the_same_result(ops, dev):
    """Large launches of the light operators run one float4 group per lane at Op::kResident workgroups per CU (unused LDS
    reserved at launch); CURL_F_TUNE_OCC = 1 asks for the uncapped library tile shape, 2..7 for k.  Only
    the schedule changes: identical bits above and below the size where the default switches (22 tiles per CU)."""
    OFF, K2, K7 = 1 << 19, 2 << 19, 7 << 19
    g = torch.Generator().manual_seed(19)
    for B, H, W in ((6, 1000, 1500), (1, 512, 768)):   # 8 790 tiles: capped by default / 384: not
        img = torch.rand(B, 3, H, W, generator=g).to(dev)
        mask = (torch.rand(B, 1, H, W, generator=g) > 0.3).to(dev)
        L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
        for f in (OFF, K2, K7, K2 | (2 << 8)):
            assert torch.equal(ops.rgb2lab(img), ops.rgb2lab(img, flags=f)), (B, f)
            assert torch.equal(ops.hsv2rgb(img), ops.hsv2rgb(img, flags=f)), (B, f)
            a, ra = ops.adjust_rgb(img, R)
            b, rb = ops.adjust_rgb(img, R, flags=f)
            assert torch.equal(a, b) and torch.equal(ra, rb), (B, f)
            assert torch.equal(ops.lab_stage(img, mask, L)[0], ops.lab_stage(img, mask, L, flags=f)[0]), (B, f)
            assert torch.equal(ops.hsv_stage(img, mask, Hk)[0], ops.hsv_stage(img, mask, Hk, flags=f)[0]), (B, f)
            assert torch.equal(ops.curl_layer_forward(img, mask, L, R, Hk)[0], ops.curl_layer_forward(img, mask, L, R, Hk, flags=f)[0])
            C = torch.exp(L[:, :16])
            assert torch.equal(ops.apply_curve(img, C, None, 2, 0, flags=0)[0], ops.apply_curve(img, C, None, 2, 0, flags=f)[0]), (B, f)


def test_layer_forward_and_backward_replay_from_a_hip_graph(ops, dev):
    """The library only enqueues on the stream it is handed (no synchronisation, no allocation of its own), so its calls
    can be captured into a HIP graph and replayed: the launch-bound shapes (a training crop batch is 22 us forward + 26 us
    backward over five kernels) are the ones that gain.  Captured through torch.cuda.CUDAGraph; replay on new data equals
    the eager calls bit for bit."""
    g = torch.Generator().manual_seed(21)
    B, H, W = 4, 64, 96
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.2).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
    gout = torch.rand(B, 3, H, W, generator=g).to(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):  # warm-up off the default stream, as graph capture asks
        for _ in range(2):
            o, r, ws = ops.curl_layer_forward(img, mask, L, R, Hk, return_workspace=True)
            ops.curl_layer_backward(img, mask, L, R, Hk, gout, None, workspace=ws)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o, r, ws = ops.curl_layer_forward(img, mask, L, R, Hk, return_workspace=True)
        gi, gL, gR, gH = ops.curl_layer_backward(img, mask, L, R, Hk, gout, None, workspace=ws)
    for seed in (1, 2):
        g2 = torch.Generator().manual_seed(seed)
        img.copy_(torch.rand(B, 3, H, W, generator=g2))
        L.copy_(torch.randn(B, 48, generator=g2) * 0.1)
        graph.replay()
        torch.cuda.synchronize(dev)
        eo, er = ops.curl_layer_forward(img, mask, L, R, Hk)
        egi, egL, egR, egH = ops.curl_layer_backward(img, mask, L, R, Hk, gout, None)
        assert torch.equal(o, eo) and torch.equal(r, er)
        assert torch.equal(gi, egi) and torch.equal(gL, egL) and torch.equal(gR, egR) and torch.equal(gH, egH)


def test_capped_and_mask_first_launches_replay_from_a_hip_graph(ops, dev):
    """A large launch carries launch parameters of its own -- the unused-LDS reservation that holds the resident workgroups
    down and the mask-first kernel variant -- and they are part of the captured kernel node: replay on
    new data equals the eager calls."""
    g = torch.Generator().manual_seed(22)
    B, H, W = 6, 1000, 1500   # 8 790 tiles: above the threshold where the caps apply
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.2).to(dev)
    mask[:, :, 300:600] = False
    R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 64))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            ops.adjust_rgb(img, R)
            ops.hsv_stage(img, mask, Hk, flags=ops.F_MASK_FIRST)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        a, ra = ops.adjust_rgb(img, R)
        h, rh = ops.hsv_stage(img, mask, Hk, flags=ops.F_MASK_FIRST)
    img.copy_(torch.rand(B, 3, H, W, generator=g))
    graph.replay()
    torch.cuda.synchronize(dev)
    ea, era = ops.adjust_rgb(img, R)
    eh, erh = ops.hsv_stage(img, mask, Hk)
    assert torch.equal(a, ea) and torch.equal(ra, era) and torch.equal(h, eh) and torch.equal(rh, erh)
    assert not h[:, :, 300:600].any()
