"""The compiled binding of the three hot entry points (curl_amd/csrc/fastcall.cpp) against the ctypes surface it shortcuts:
same kernels, same bits; every call that is not the plain one lands in the checked path and raises what it raised before."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _case(dev, B=3, H=40, W=56, seed=5):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.25).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.1).to(dev) for n in (48, 48, 64))
    gout = torch.rand(B, 3, H, W, generator=g).to(dev)
    greg = torch.rand(B, generator=g).to(dev)
    return img, mask, L, R, Hk, gout, greg


def test_binding_is_loaded_and_bound(dev):
    from curl_amd import _lib
    fast = _lib.fast()
    assert fast is not None and fast.__file__.endswith(".so") and "curl_amd" in fast.__file__


@pytest.mark.parametrize("mask_kind", ["bool", "f32", "none", "u8"])
def test_forward_and_backward_equal_the_checked_path_bit_for_bit(dev, mask_kind):
    from curl_amd import ops
    img, mask, L, R, Hk, gout, greg = _case(dev)
    m = {"bool": mask, "f32": mask.float() * 0.7, "none": None, "u8": mask.to(torch.uint8)}[mask_kind]
    out, reg, ws = ops.curl_layer_forward(img, m, L, R, Hk, return_workspace=True)
    out2, reg2, ws2 = ops._curl_layer_forward_checked(img, m, L, R, Hk, return_workspace=True)
    assert torch.equal(out, out2) and torch.equal(reg, reg2)
    # the workspace row (curl_kernels.hip WS_*): collapsed curves + regularisers [0, 24), the stamp at 29, the exp'd knots from 32
    # (the rest of the row is never written: whatever the allocator handed over)
    a, b = ws.view(torch.int32).view(3, -1), ws2.view(torch.int32).view(3, -1)
    assert a.shape[1] == 32 + 160 and all(torch.equal(a[:, sl], b[:, sl]) for sl in (slice(0, 24), slice(29, 30), slice(32, 192)))
    assert len(ops.curl_layer_forward(img, m, L, R, Hk)) == 2
    for need in (True, False):
        for w in (None, ws):
            a = ops.curl_layer_backward(img, m, L, R, Hk, gout, grad_reg=greg, need_grad_img=need, workspace=w)
            b = ops._curl_layer_backward_checked(img, m, L, R, Hk, gout, grad_reg=greg, need_grad_img=need, workspace=w)
            assert (a[0] is None) == (not need) and (b[0] is None) == (not need)
            for x, y in zip(a, b):
                assert x is None or torch.equal(x, y)
    a = ops.curl_layer_backward(img, m, L, R, Hk, gout)
    b = ops._curl_layer_backward_checked(img, m, L, R, Hk, gout)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_trispace_bytes_equal_the_checked_path(dev):
    from curl_amd import ops
    g = torch.Generator().manual_seed(6)
    u8 = torch.randint(0, 256, (2, 33, 47, 3), generator=g, dtype=torch.uint8).to(dev)
    wm = torch.randint(0, 256, (2, 33, 47), generator=g, dtype=torch.uint8).to(dev)
    for nc in (126, 35):
        c = (torch.randn(2, 3, 3, nc, generator=g) * 0.2).to(dev)
        for w in (None, wm):
            assert torch.equal(ops.trispace_forward_u8hwc(u8, c, w), ops._trispace_forward_u8hwc_checked(u8, c, white_mask=w))


def test_calls_that_are_not_plain_take_the_checked_path(dev):
    """The binding answers None -- and the checked path computes, or raises what it always raised."""
    from curl_amd import _lib, ops
    fast = _lib.fast()
    img, mask, L, R, Hk, gout, greg = _case(dev)
    ref, ref_reg = ops._curl_layer_forward_checked(img, mask, L, R, Hk)
    # a non-contiguous image, a one-image mask broadcast over the batch, float64 knots, uneven torch.chunk knots
    wide = torch.empty(3, 3, 40, 112, device=dev)
    wide[..., ::2] = img
    assert fast.layer_fwd(wide[..., ::2], mask, L, R, Hk, 0) is None
    assert torch.equal(ops.curl_layer_forward(wide[..., ::2], mask, L, R, Hk)[0], ref)
    one = mask[:1]
    assert fast.layer_fwd(img, one, L, R, Hk, 0) is None
    assert torch.equal(ops.curl_layer_forward(img, one, L, R, Hk)[0], ops._curl_layer_forward_checked(img, one.expand(3, 1, 40, 56), L, R, Hk)[0])
    assert fast.layer_fwd(img, mask, L.double(), R, Hk, 0) is None
    assert torch.equal(ops.curl_layer_forward(img, mask, L.double(), R, Hk)[0], ref)
    assert fast.layer_fwd(img, mask, L[:, :47].contiguous(), R, Hk, 0) is None  # 47 = 16 + 16 + 15
    assert ops.curl_layer_forward(img, mask, L[:, :47].contiguous(), R, Hk)[0].shape == img.shape
    # CPU tensors, wrong shapes, an empty image, a knot count out of range: the documented errors
    assert fast.layer_fwd(img.cpu(), None, L.cpu(), R.cpu(), Hk.cpu()) is None
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.curl_layer_forward(img.cpu(), None, L, R, Hk)
    with pytest.raises(ValueError, match=r"must be \[B,3,H,W\]"):
        ops.curl_layer_forward(img[:, :2], None, L, R, Hk)
    with pytest.raises(ValueError, match="mask must be"):
        ops.curl_layer_forward(img, mask[..., :10], L, R, Hk)
    with pytest.raises(ValueError, match="knots per curve"):
        ops.curl_layer_forward(img, mask, L[:, :3].contiguous(), R, Hk)
    with pytest.raises(ValueError, match="grad_out"):
        ops.curl_layer_backward(img, mask, L, R, Hk, gout[..., :10])
    with pytest.raises(ValueError, match="workspace is not"):
        ops.curl_layer_backward(img, mask, L, R, Hk, gout, workspace=torch.zeros(4, device=dev))
    empty = ops.curl_layer_forward(img[:0], None, L[:0], R[:0], Hk[:0])
    assert empty[0].shape == (0, 3, 40, 56) and empty[1].shape == (0,)
    # a flag the library refuses comes back as the C ABI's code through the binding, and raises as through ctypes
    with pytest.raises(ValueError, match="code -6"):
        ops.curl_layer_forward(img, mask, L, R, Hk, flags=1 << 30)
    # out=: written in place by either path; a wrong one is the checked path's to refuse
    o = torch.empty_like(img)
    assert ops.curl_layer_forward(img, mask, L, R, Hk, out=o)[0] is o and torch.equal(o, ref)
    assert fast.layer_fwd(img, mask, L, R, Hk, 0, torch.empty(3, 3, 40, 28, device=dev)) is None
    with pytest.raises(ValueError, match="out must be"):
        ops.curl_layer_forward(img, mask, L, R, Hk, out=torch.empty(3, 3, 40, 28, device=dev))
    x = img.clone()
    assert torch.equal(ops.curl_layer_forward(x, mask, L, R, Hk, out=x)[0], ref)  # in place


def test_binding_follows_the_current_stream(dev):
    from curl_amd import ops
    img, mask, L, R, Hk, gout, greg = _case(dev, B=2, H=200, W=300)
    ref = ops.curl_layer_forward(img, mask, L, R, Hk)[0]
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        x = img * 1.0  # produced on s: a launch on another stream would race it
        y = ops.curl_layer_forward(x, mask, L, R, Hk)[0]
    s.synchronize()
    assert torch.equal(y, ref)
