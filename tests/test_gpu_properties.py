"""The property tests of tests/test_properties.py against the COMPILED kernels (v_log_f32 / v_exp_f32 / v_rcp_f32 instead
of the host twin's libm), through the C ABI: hypothesis searches ties, 8-bit grid values, threshold neighbours and
out-of-range inputs, and shrinks a failure to one pixel."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import curl_oracle as O
from test_properties import channel, layer_backward_lone_flips, pixels, unit_channel, _near

import os

pytestmark = pytest.mark.gpu
SCALE = float(os.environ.get("CURL_HYP_SCALE", 1))  # CURL_HYP_SCALE=10: the soak run (profiles/r0*/hypothesis_soak.log)
# The suite's own run is DERANDOMISED (the same examples every time): the reference has isolated points -- an exact tie of two
# float32 intermediates, DESIGN.md 4 item 5, one colour in 16.7 M -- that a random search meets once in ~1e3 runs and that the
# conditioned bound does not cover; the soak run (SCALE > 1) keeps the random search and is read by a person.
COMMON = dict(deadline=None, derandomize=SCALE == 1, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from curl_amd import _lib, ops as _ops
    _lib.load()
    return _ops


def _sens(fn, t, h=1e-6):
    r64 = fn(t.double())
    S = torch.zeros(1, 1, t.shape[3], dtype=torch.float64)
    for k in range(3):
        for sgn in (h, -h):
            p = t.double().clone()
            p[:, k] += sgn
            S = torch.maximum(S, (fn(p) - r64).abs().amax(1) / h)
    return S


@pytest.mark.parametrize("op,tol", [("rgb2lab", 2e-6), ("rgb2hsv", 1e-6), ("hsv2rgb", 1e-6)])
@settings(max_examples=int(120 * SCALE), **COMMON)
@given(x=pixels())
def test_converter_kernels_on_edge_values(ops, op, tol, x):
    t = torch.from_numpy(x)
    ref = getattr(O, op)(t).numpy()
    got = getattr(ops, op)(t.cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    assert float(np.abs(got.astype(np.float64) - ref).max()) <= tol * max(1.0, float(np.abs(ref).max()))


@settings(max_examples=int(120 * SCALE), **COMMON)
@given(x=pixels(elem=st.one_of(st.floats(-0.25, 1.25, width=32), st.sampled_from([v for t in (6 / 29, 0.0, 1.0) for v in _near(t)]))))
def test_lab2rgb_kernel_on_edge_values(ops, x):
    t = torch.from_numpy(x)
    ref = O.lab2rgb(t).numpy()
    got = ops.lab2rgb(t.cuda()).cpu().numpy()
    bound = torch.clamp(8e-7 * _sens(O.lab2rgb, t), min=2e-6 * max(1.0, float(np.abs(ref).max())))
    d = torch.from_numpy(np.abs(got.astype(np.float64) - ref)).amax(1)
    assert bool((d <= bound).all()), (float(d.max()), float(bound.min()))


@settings(max_examples=int(60 * SCALE), **COMMON)
@given(x=pixels(elem=unit_channel, max_px=8), seed=st.integers(0, 2 ** 16), kind=st.sampled_from(["none", "bool", "f32"]))
def test_fused_layer_kernel_error_is_bounded_by_the_chain_conditioning(ops, x, seed, kind):
    """|HIP - reference (float32)| <= max(1e-5, 2e-6 * S) for every pixel, every mask kind (the all-ones masks: the bound
    is about the arithmetic; masked-out pixels are covered by the golden and shape tests)."""
    g = torch.Generator().manual_seed(seed)
    L, R, H = (torch.randn(1, n, generator=g) * 0.1 for n in (48, 48, 64))
    img = torch.from_numpy(x)
    n = img.shape[3]
    ones = torch.ones(1, 1, 1, n)
    ref, _ = O.curl_layer(img, ones, L, R, H)

    def chain64(p):
        return O.curl_layer(p, ones.double(), L.double(), R.double(), H.double())[0]
    S = _sens(chain64, img)
    mask = None if kind == "none" else (ones.bool().cuda() if kind == "bool" else ones.cuda())
    got, _ = ops.curl_layer_forward(img.cuda(), mask, L.cuda(), R.cuda(), H.cuda())
    d = (got.cpu().double() - ref.double()).abs().amax(1)
    bound = torch.clamp(2e-6 * S, min=1e-5)
    assert bool((d <= bound).all()), (float(d.max()), float(S.max()))


@settings(max_examples=int(60 * SCALE), **COMMON)
@given(x=pixels(elem=channel, max_px=8), seed=st.integers(0, 2 ** 16), kind=st.sampled_from(["none", "bool", "f32"]))
def test_fused_hsv_stage_kernel_error_is_bounded_by_its_conditioning(ops, x, seed, kind):
    """curl_hsv_stage_f32 (model.py:163-169) on hypothesis' edge values -- exact and near channel ties, 8-bit grid values,
    out-of-range inputs: |HIP - reference (float32)| <= max(3e-6, 2e-6 * S), S the float64 stage's input sensitivity (the
    hue is discontinuous where g crosses b under a red maximum: S is what says which pixels sit there)."""
    g = torch.Generator().manual_seed(seed)
    H = torch.randn(1, 64, generator=g) * 0.1
    img = torch.from_numpy(x)
    ones = torch.ones(1, 1, 1, img.shape[3])
    ref, _ = O.hsv_stage(img, ones, H)
    S = _sens(lambda p: O.hsv_stage(p, ones.double(), H.double())[0], img)
    mask = None if kind == "none" else (ones.bool().cuda() if kind == "bool" else ones.cuda())
    got, _ = ops.hsv_stage(img.cuda(), mask, H.cuda())
    assert bool(torch.isfinite(got).all())
    d = (got.cpu().double() - ref.double()).abs().amax(1)
    bound = torch.clamp(2e-6 * S, min=3e-6)
    assert bool((d <= bound).all()), (float(d.max()), float(S.max()))


def _pad(t, n=256):
    """[1,C,1,P] -> [1,C,1,n] by repeating the pixels (a launch of a few pixels and one of a wavefront's worth run the same code;
    the float4 path needs a multiple of four)."""
    reps = -(-n // t.shape[3])
    return t.repeat(1, 1, 1, reps)[..., :n].contiguous()


@settings(max_examples=int(60 * SCALE), **COMMON)
@given(x=pixels(elem=unit_channel, max_px=8), seed=st.integers(0, 2 ** 16), kind=st.sampled_from(["none", "bool", "f32"]),
       shift=st.sampled_from([-0.7, -0.3, 0.0]))
def test_layer_backward_kernel_where_the_reference_is_unambiguous(ops, x, seed, kind, shift):
    """tests/test_properties.py's statement on the device (curl_layer_bwd_f32, every mask kind): wherever the reference's
    float32 and float64 autograd agree on a pixel's gradient, the kernel may not be off by a gate."""
    def backward(img, ones, L, R, H, w, wr):
        P = img.shape[3]
        mask = None if kind == "none" else (_pad(ones).bool().cuda() if kind == "bool" else _pad(ones).cuda())
        return ops.curl_layer_backward(_pad(img).cuda(), mask, L.cuda(), R.cuda(), H.cuda(), _pad(w).cuda(), wr.cuda())[0].cpu()[..., :P]
    bad, dmax, Gs = layer_backward_lone_flips(backward, x, seed, shift)
    assert not bad, (bad, dmax, Gs)
