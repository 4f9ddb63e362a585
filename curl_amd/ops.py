"""Tensor-level entry points: torch tensors in, HIP kernels underneath (via the C ABI).

PyTorch is plumbing here: it owns device memory (caching allocator) and the stream.
Every function enqueues on torch's CURRENT stream and returns without synchronising.
"""
import functools
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import F_EXACT_ORDER, F_MASK_FIRST, F_PWL, MASK_F32, MASK_NONE, MASK_U8


def _empty_ok(mask_arg=None):
    """The reference's eager ops accept empty tensors ([0,3,H,W], or H*W == 0) and return empty ones; the kernels
    are never launched on zero pixels.  Ops that also return a regulariser still owe it (it depends on the knots
    only): it is produced by the same entry point on a 1x1 stand-in image."""
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(img, *args, **kwargs):
            if not (isinstance(img, torch.Tensor) and img.dim() == 4 and img.shape[1] == 3 and img.numel() == 0):
                return fn(img, *args, **kwargs)
            _need_device(img, "img")
            out = torch.empty_like(img)
            if mask_arg is None:          # image -> image
                return out
            B = img.shape[0]
            if B == 0:
                reg = torch.zeros(0, dtype=torch.float32, device=img.device)
            else:
                args = list(args)
                if mask_arg >= 0:
                    args[mask_arg] = None  # the mask's shape belongs to the empty image
                kwargs.pop("out", None)
                res = fn(img.new_zeros((B, 3, 1, 1)), *args, **kwargs)
                reg = res[1]
            return (out, reg, None) if kwargs.get("return_workspace") else (out, reg)
        return wrapper
    return deco


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """The raw handle of torch's current stream on the tensor's device.  torch._C._cuda_getCurrentRawStream is the direct
    accessor (what torch's own compiled-kernel launchers use): 0.2 us instead of the ~2.5 us of building a torch.cuda.Stream
    object per call -- on a 12 us launch the host side is the cost (tools/small_batch.py)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_same_device(named):
    """Every tensor of one call must live on ONE device: a knot / mask / gradient tensor on cuda:1 next to an image on
    cuda:0 would hand the kernel a foreign pointer (a GPU fault) where the reference's eager ops raise cleanly."""
    devs = [(n, t.device) for n, t in named if isinstance(t, torch.Tensor)]
    cuda = [(n, d) for n, d in devs if d.type == "cuda"]
    if len({d for _, d in cuda}) > 1:
        raise ValueError("expected all tensors to be on the same device, got " +
                         ", ".join(f"{n} on {d}" for n, d in cuda))
    return cuda[0][1] if cuda and len(cuda) == len(devs) else None


def _one_device(fn):
    """Check that all tensor arguments share a device and run the call with that device current (the library
    enqueues on the stream it is handed and never calls hipSetDevice itself)."""
    import inspect
    params = list(inspect.signature(fn).parameters)

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        # fast path (the common call): every tensor on the CURRENT device -- nothing to switch, nothing to report
        dev = None
        for a in args:
            if isinstance(a, torch.Tensor):
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    dev = False
                    break
        if dev and not kwargs.keys() - _PLAIN_KWARGS and dev.type == "cuda" and dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        named = list(zip(params, args)) + list(kwargs.items())
        dev = _check_same_device(named)
        if dev is None:  # a CPU tensor (or none at all): the body's own checks raise the documented errors
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


# keyword arguments that carry no tensor, or one whose device the body checks itself against the image's (`out`: _check_out,
# `workspace`: curl_layer_backward); any other tensor passed by keyword takes the slow, fully checked path
_PLAIN_KWARGS = frozenset(("flags", "return_workspace", "residual_only", "need_grad_img", "max_intensity", "window_size",
                           "want_L", "out", "workspace"))


def _coeffs32(coeffs, pairs=True):
    """float32 and contiguous; pairs=True (the spatial 126-coefficient tables, which the kernels copy into LDS as 8-byte
    pairs): also 8-byte aligned -- a view that starts at an odd float of its storage is copied."""
    c = coeffs.to(torch.float32).contiguous()
    return c.clone() if pairs and c.data_ptr() % 8 else c


def _check_out(out, img):
    """A caller-supplied `out` is written in place by the kernel: anything but a contiguous float32 tensor of the
    image's shape on the image's device would be an out-of-bounds or foreign-device write."""
    if not isinstance(out, torch.Tensor):
        raise TypeError(f"out must be a torch.Tensor, got {type(out).__name__}")
    if out.shape != img.shape or out.dtype != img.dtype or not out.is_contiguous() or out.device != img.device:
        raise ValueError(f"out must be a contiguous {img.dtype} tensor of shape {tuple(img.shape)} on {img.device}, got "
                         f"{out.dtype} {tuple(out.shape)} on {out.device}"
                         f"{'' if out.is_contiguous() else ' (not contiguous)'}")
    return out


def _need_device(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(
            f"curl_amd: {name} is on {t.device}; this path runs on a HIP device only "
            "(no CPU fallback -- move the tensor to cuda)")


def _image(t, name="img"):
    _need_device(t, name)
    if t.dim() != 4 or t.shape[1] != 3:
        raise ValueError(f"{name} must be [B,3,H,W], got {tuple(t.shape)}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if t.numel() == 0:
        raise ValueError(f"{name} is empty: {tuple(t.shape)}")
    return t.contiguous()


def _knots(t, name, ncurves, B):
    _need_device(t, name)
    if t.dim() != 2 or t.shape[0] != B:
        raise ValueError(f"{name} must be [B={B}, {ncurves}*K], got {tuple(t.shape)}")
    N = t.shape[1]
    K = -(-N // ncurves)  # torch.chunk(P, n, dim=1): chunks of ceil(N / n) (curves.py:53,105,152) ...
    if ncurves > 1 and -(-N // K) != ncurves:
        # ... of which there may then be fewer than n: the reference's tuple unpacking fails on that
        raise ValueError(f"{name}: torch.chunk splits {N} parameters into {-(-N // K)} chunks, not {ncurves} curves")
    K_last = N - (ncurves - 1) * K  # ... and the last one holds the remainder
    if K < 2 or K > _lib.MAX_KNOTS or K_last < 2:
        raise ValueError(f"{name}: {K} knots per curve ({K_last} in the last); supported range is [2, {_lib.MAX_KNOTS}]")
    # the C ABI's packed form (include/curl_hip.h CURL_K_UNEVEN): K, and the last curve's count in the high half if it differs
    if t.dtype is not torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t, (K if K_last == K else K | (K_last << 16))


def _mask(mask, img):
    """-> (tensor or None, mask_kind).  Accepts None, bool/uint8 or floating [B|1,1,H,W].
    bool is what the reference passes (data.py:190: `mask > 0`).  A uint8 mask is read the same way -- NONZERO KEEPS the pixel --
    which equals torch's `img * mask` for bytes 0 / 1 only: a 0 / 255 mask straight from a PNG would scale by 255 in torch.
    A floating mask multiplies by its value, as torch does (pass `mask.float()` for that behaviour with other integer types)."""
    if mask is None:
        return None, MASK_NONE
    _need_device(mask, "mask")
    B, _, H, W = img.shape
    if mask.dim() == 3:
        mask = mask.unsqueeze(1)
    if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[2] != H or mask.shape[3] != W \
            or mask.shape[0] not in (1, B):
        raise ValueError(f"mask must be [B,1,H,W] matching img {tuple(img.shape)}, got {tuple(mask.shape)}")
    if mask.shape[0] != B:
        mask = mask.expand(B, 1, H, W)
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8), MASK_U8
    if mask.dtype == torch.uint8:
        return mask.contiguous(), MASK_U8
    if mask.is_floating_point():
        return mask.to(torch.float32).contiguous(), MASK_F32
    raise TypeError(f"mask dtype {mask.dtype} not supported (bool, uint8 or floating)")


def _workspace(B, n_knots, device):
    nbytes = _lib.load().curl_workspace_bytes(B, n_knots)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------ curves.py
@_one_device
@_empty_ok(mask_arg=-1)
def apply_curve(img, C, slope_sqr_diff, channel_in, channel_out, flags=F_EXACT_ORDER):
    """curves.apply_curve (curves.py:4-38).  C are the knots after exp, [B,K].
    slope_sqr_diff [B] is updated in place (curves.py:24) and returned; None skips it."""
    lib = _lib.load()
    img = _image(img)
    B, _, H, W = img.shape
    Cc, K = _knots(C, "C", 1, B)
    reg = slope_sqr_diff
    if reg is not None:
        _need_device(reg, "slope_sqr_diff")
        if reg.shape != (B,) or reg.dtype != torch.float32 or not reg.is_contiguous():
            raise ValueError("slope_sqr_diff must be a contiguous float32 [B] tensor")
    out = torch.empty_like(img)
    rc = lib.curl_apply_curve_f32(img.data_ptr(), Cc.data_ptr(), out.data_ptr(), _ptr(reg), B, H, W, K,
                                  int(channel_in), int(channel_out), flags, _stream(img))
    _lib.check(rc, "curl_apply_curve_f32")
    return out, reg


def _adjust(fn_name, ncurves, img, raw, flags):  # noqa: E302
    lib = _lib.load()
    img = _image(img)
    B, _, H, W = img.shape
    rawc, K = _knots(raw, "knots", ncurves, B)
    out = torch.empty_like(img)
    reg = torch.empty(B, dtype=torch.float32, device=img.device)
    ws, nbytes = _workspace(B, rawc.shape[1], img.device)
    rc = getattr(lib, fn_name)(img.data_ptr(), rawc.data_ptr(), out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nbytes,
                               B, H, W, K, flags, _stream(img))
    _lib.check(rc, fn_name)
    return out, reg


@_one_device
@_empty_ok(mask_arg=-1)
def adjust_rgb(img, R, flags=0):
    """curves.adjust_rgb (curves.py:90-133), regulariser seeded with zeros."""
    return _adjust("curl_adjust_rgb_f32", 3, img, R, flags)


@_one_device
@_empty_ok(mask_arg=-1)
def adjust_lab(img, L, flags=0):
    """curves.adjust_lab (curves.py:136-180)."""
    return _adjust("curl_adjust_lab_f32", 3, img, L, flags)


@_one_device
@_empty_ok(mask_arg=-1)
def adjust_hsv(img, S, flags=0):
    """curves.adjust_hsv (curves.py:41-87)."""
    return _adjust("curl_adjust_hsv_f32", 4, img, S, flags)


# ------------------------------------------------------------------ colors.py
def _convert(fn_name, img, flags=0):
    lib = _lib.load()
    img = _image(img)
    B, _, H, W = img.shape
    out = torch.empty_like(img)
    rc = getattr(lib, fn_name)(img.data_ptr(), out.data_ptr(), B, H, W, flags, _stream(img))
    _lib.check(rc, fn_name)
    return out


@_one_device
@_empty_ok()
def rgb2lab(img, flags=0):
    """colors.RGB2LAB.forward (colors.py:27-62)."""
    return _convert("curl_rgb2lab_f32", img, flags)


@_one_device
@_empty_ok()
def lab2rgb(img, flags=0):
    """colors.LAB2RGB.forward (colors.py:88-123)."""
    return _convert("curl_lab2rgb_f32", img, flags)


@_one_device
@_empty_ok()
def rgb2hsv(img, flags=0):
    """colors.RGB2HSV.forward (colors.py:195-242)."""
    return _convert("curl_rgb2hsv_f32", img, flags)


@_one_device
@_empty_ok()
def hsv2rgb(img, flags=0):
    """colors.HSV2RGB.forward (colors.py:131-177)."""
    return _convert("curl_hsv2rgb_f32", img, flags)


# ------------------------------------------------------------------ model.py: fused stages
@_one_device
@_empty_ok(mask_arg=0)
def lab_stage(img, mask, L, flags=0, out=None):
    """RGB -> Lab -> 3 curves -> *mask -> RGB in one pass (model.py:151-157). -> (rgb, reg_lab)."""
    lib = _lib.load()
    img = _image(img)
    B, _, H, W = img.shape
    Lc, Kl = _knots(L, "L", 3, B)
    m, kind = _mask(mask, img)
    out = torch.empty_like(img) if out is None else _check_out(out, img)
    reg = torch.empty(B, dtype=torch.float32, device=img.device)
    ws, nbytes = _workspace(B, Lc.shape[1], img.device)
    rc = lib.curl_lab_stage_f32(img.data_ptr(), _ptr(m), kind, Lc.data_ptr(), out.data_ptr(), reg.data_ptr(),
                                ws.data_ptr(), nbytes, B, H, W, Kl, flags, _stream(img))
    _lib.check(rc, "curl_lab_stage_f32")
    return out, reg


@_one_device
@_empty_ok(mask_arg=0)
def hsv_stage(img, mask, H, flags=0, out=None):
    """RGB -> HSV -> 4 curves -> *mask -> RGB in one pass (model.py:163-169): the layer's RGB residual. -> (rgb, reg_hsv)."""
    lib = _lib.load()
    img = _image(img)
    B, _, Hh, W = img.shape
    Hc, Kh = _knots(H, "H", 4, B)
    m, kind = _mask(mask, img)
    out = torch.empty_like(img) if out is None else _check_out(out, img)
    reg = torch.empty(B, dtype=torch.float32, device=img.device)
    ws, nbytes = _workspace(B, Hc.shape[1], img.device)
    rc = lib.curl_hsv_stage_f32(img.data_ptr(), _ptr(m), kind, Hc.data_ptr(), out.data_ptr(), reg.data_ptr(),
                                ws.data_ptr(), nbytes, B, Hh, W, Kh, flags, _stream(img))
    _lib.check(rc, "curl_hsv_stage_f32")
    return out, reg


def curl_layer_forward(img, mask, L, R, H, flags=0, out=None, return_workspace=False):
    """CURLLayer.forward (model.py:137-176) in one pass over the pixels. -> (img, reg[B]).
    L [B,3*Kl], R [B,3*Kr], H [B,4*Kh] are the already-sliced raw knots.
    return_workspace: also return the knot workspace the call filled (-> (img, reg, ws)); handed back to
    curl_layer_backward(..., workspace=ws) it saves that call its knot-prep launch.
    The plain call (every tensor on the current device, float32, contiguous, even knot counts) goes through the
    compiled binding (csrc/fastcall.cpp: one Python -> C++ transition); anything else -- and every error -- through the
    checked path below."""
    if type(img) is torch.Tensor:
        fast = _lib.fast()
        if fast is not None:
            r = fast.layer_fwd(img, mask, L, R, H, flags, out)
            if r is not None:
                if r.__class__ is int:
                    _lib.check(r, "curl_layer_fwd_f32")
                return r if return_workspace else r[:2]
    return _curl_layer_forward_checked(img, mask, L, R, H, flags=flags, out=out, return_workspace=return_workspace)


@_one_device
@_empty_ok(mask_arg=0)
def _curl_layer_forward_checked(img, mask, L, R, H, flags=0, out=None, return_workspace=False):
    lib = _lib.load()
    img = _image(img)
    B, _, Hh, W = img.shape
    Lc, Kl = _knots(L, "L", 3, B)
    Rc, Kr = _knots(R, "R", 3, B)
    Hc, Kh = _knots(H, "H", 4, B)
    m, kind = _mask(mask, img)
    out = torch.empty_like(img) if out is None else _check_out(out, img)
    reg = torch.empty(B, dtype=torch.float32, device=img.device)
    ws, nbytes = _workspace(B, Lc.shape[1] + Rc.shape[1] + Hc.shape[1], img.device)
    rc = lib.curl_layer_fwd_f32(img.data_ptr(), _ptr(m), kind, Lc.data_ptr(), Rc.data_ptr(), Hc.data_ptr(),
                                out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nbytes, B, Hh, W, Kl, Kr, Kh,
                                flags, _stream(img))
    _lib.check(rc, "curl_layer_fwd_f32")
    return (out, reg, ws) if return_workspace else (out, reg)


def _slab(rows, H):
    r0, r1 = int(rows[0]), int(rows[1])
    if not (0 <= r0 < r1 <= H):
        raise ValueError(f"rows must satisfy 0 <= r0 < r1 <= H={H}, got {rows}")
    return r0, r1 - r0


@_one_device
def curl_layer_forward_rows(img, mask, L, R, H, rows, out, flags=0):
    """CURLLayer.forward on rows [r0, r1) of every image of the FULL tensors, in place in `out` (full-size, required):
    the split-pixels layout (shard.apply_row_slab) without the .contiguous() copy of a row slice.  Rows outside the
    slab are neither read nor written.  -> (out, reg[B])."""
    lib = _lib.load()
    img = _image(img)
    B, _, Hh, W = img.shape
    r0, n = _slab(rows, Hh)
    Lc, Kl = _knots(L, "L", 3, B)
    Rc, Kr = _knots(R, "R", 3, B)
    Hc, Kh = _knots(H, "H", 4, B)
    m, kind = _mask(mask, img)
    out = _check_out(out, img)
    reg = torch.empty(B, dtype=torch.float32, device=img.device)
    ws, nbytes = _workspace(B, Lc.shape[1] + Rc.shape[1] + Hc.shape[1], img.device)
    rc = lib.curl_layer_fwd_slab_f32(img.data_ptr(), _ptr(m), kind, Lc.data_ptr(), Rc.data_ptr(), Hc.data_ptr(),
                                     out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nbytes, B, Hh, W, r0, n, Kl, Kr, Kh,
                                     flags, _stream(img))
    _lib.check(rc, "curl_layer_fwd_slab_f32")
    return out, reg


@_one_device
def trispace_forward_rows(img, coeffs, rows, out, residual_only=False):
    """trispace_forward on rows [r0, r1) of every image of the FULL tensors (pixel coordinates stay the full image's:
    the rows written equal the same rows of the whole-image call bit for bit)."""
    lib = _lib.load()
    img = _image(img)
    B, _, H, W = img.shape
    r0, n = _slab(rows, H)
    _need_device(coeffs, "coeffs")
    if coeffs.dim() != 4 or coeffs.shape[:3] != (B, 3, 3) or coeffs.shape[3] not in (126, 35):
        raise ValueError(f"coeffs must be [B={B},3,3,126|35], got {tuple(coeffs.shape)}")
    c = _coeffs32(coeffs, pairs=coeffs.shape[3] == 126)
    out = _check_out(out, img)
    rc = lib.curl_trispace_fwd_slab_f32(img.data_ptr(), c.data_ptr(), out.data_ptr(), B, H, W, r0, n, c.shape[3],
                                        _lib.F_RESIDUAL_ONLY if residual_only else 0, _stream(img))
    _lib.check(rc, "curl_trispace_fwd_slab_f32")
    return out


def curl_layer_backward(img, mask, L, R, H, grad_out, grad_reg=None, need_grad_img=True, workspace=None, flags=0):
    """Backward of curl_layer_forward (what autograd would run through model.py:137-176).
    -> (grad_img or None, grad_L, grad_R, grad_H).
    workspace: the tensor curl_layer_forward(..., return_workspace=True) returned for the SAME knots (CURL_F_WS_READY).
    (Plain calls through the compiled binding, as curl_layer_forward.)"""
    fast = _lib.fast() if type(img) is torch.Tensor else None
    if fast is not None:
        r = fast.layer_bwd(img, mask, L, R, H, grad_out, grad_reg, need_grad_img, workspace, flags & F_MASK_FIRST)
        if r is not None:
            if r.__class__ is int:
                _lib.check(r, "curl_layer_bwd_f32")
            return r
    return _curl_layer_backward_checked(img, mask, L, R, H, grad_out, grad_reg=grad_reg, need_grad_img=need_grad_img,
                                        workspace=workspace, flags=flags)


@_one_device
def _curl_layer_backward_checked(img, mask, L, R, H, grad_out, grad_reg=None, need_grad_img=True, workspace=None, flags=0):
    lib = _lib.load()
    img = _image(img)
    grad_out = _image(grad_out, "grad_out")
    if grad_out.shape != img.shape:
        raise ValueError(f"grad_out {tuple(grad_out.shape)} does not match img {tuple(img.shape)}")
    B, _, Hh, W = img.shape
    Lc, Kl = _knots(L, "L", 3, B)
    Rc, Kr = _knots(R, "R", 3, B)
    Hc, Kh = _knots(H, "H", 4, B)
    m, kind = _mask(mask, img)
    if grad_reg is not None:
        _need_device(grad_reg, "grad_reg")
        grad_reg = grad_reg.to(torch.float32).contiguous()
        if grad_reg.shape != (B,):
            raise ValueError("grad_reg must be [B]")
    g_img = torch.empty_like(img) if need_grad_img else None
    gL, gR, gH = torch.empty_like(Lc), torch.empty_like(Rc), torch.empty_like(Hc)
    ws, nbytes = _workspace(B, Lc.shape[1] + Rc.shape[1] + Hc.shape[1], img.device)
    bflags = flags & F_MASK_FIRST  # the one forward flag that means something here
    if workspace is not None:
        if workspace.device != img.device or workspace.dtype != torch.float32 or workspace.numel() * 4 < nbytes:
            raise ValueError("workspace is not the tensor curl_layer_forward returned for this batch")
        ws, bflags = workspace, bflags | _lib.F_WS_READY
    sbytes = lib.curl_layer_bwd_scratch_bytes(B, Hh, W)
    scratch = torch.empty(sbytes // 4, dtype=torch.float32, device=img.device)
    rc = lib.curl_layer_bwd_f32(img.data_ptr(), _ptr(m), kind, Lc.data_ptr(), Rc.data_ptr(), Hc.data_ptr(),
                                grad_out.data_ptr(), _ptr(grad_reg), _ptr(g_img), gL.data_ptr(), gR.data_ptr(),
                                gH.data_ptr(), ws.data_ptr(), nbytes, scratch.data_ptr(), sbytes, B, Hh, W, Kl, Kr, Kh,
                                bflags, _stream(img))
    _lib.check(rc, "curl_layer_bwd_f32")
    return g_img, gL, gR, gH


# ------------------------------------------------------------------ polynomial path (model.py:206-520)
@_one_device
@_empty_ok()
def trispace_forward(img, coeffs, residual_only=False, flags=0):
    """TriSpaceRegNet.generate_residual (+ generate_image unless residual_only), model.py:499-520, in one pass.
    coeffs [B,3,3,NC] with NC = 126 (spatial) or 35; [:,0]=R, [:,1]=L, [:,2]=H (model.py:526)."""
    lib = _lib.load()
    img = _image(img)
    B, _, H, W = img.shape
    _need_device(coeffs, "coeffs")
    if coeffs.dim() != 4 or coeffs.shape[:3] != (B, 3, 3) or coeffs.shape[3] not in (126, 35):
        raise ValueError(f"coeffs must be [B={B},3,3,126|35], got {tuple(coeffs.shape)}")
    c = _coeffs32(coeffs, pairs=coeffs.shape[3] == 126)
    out = torch.empty_like(img)
    rc = lib.curl_trispace_fwd_f32(img.data_ptr(), c.data_ptr(), out.data_ptr(), B, H, W, c.shape[3],
                                   flags | (_lib.F_RESIDUAL_ONLY if residual_only else 0), _stream(img))
    _lib.check(rc, "curl_trispace_fwd_f32")
    return out


@_one_device
def trispace_backward(img, coeffs, grad_out, residual_only=False):
    """d loss / d coeffs [B,3,3,NC] of trispace_forward, given grad_out = d loss / d out."""
    lib = _lib.load()
    img, grad_out = _image(img), _image(grad_out, "grad_out")
    B, _, H, W = img.shape
    c = _coeffs32(coeffs, pairs=coeffs.shape[3] == 126)
    nc = c.shape[3]
    g = torch.empty_like(c)
    nbytes = lib.curl_trispace_bwd_scratch_bytes(B, H, W, nc)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=img.device)
    rc = lib.curl_trispace_bwd_f32(img.data_ptr(), c.data_ptr(), grad_out.data_ptr(), g.data_ptr(), scratch.data_ptr(),
                                   nbytes, B, H, W, nc, _lib.F_RESIDUAL_ONLY if residual_only else 0, _stream(img))
    _lib.check(rc, "curl_trispace_bwd_f32")
    return g


@_one_device
def poly_layer(img, coeffs):
    """ChannelPolyLayer(degree=4) / Deg4MobilePolyLayer forward (model.py:295-333, 399-415):
    img [B,V,H,W] with V = 5 or 3, coeffs [B,3,126|35] -> [B,3,H,W]."""
    lib = _lib.load()
    _need_device(img, "img")
    _need_device(coeffs, "coeffs")
    if img.dim() != 4 or img.shape[1] not in (3, 5) or img.dtype != torch.float32:
        raise ValueError(f"img must be float32 [B,3|5,H,W], got {tuple(img.shape)} {img.dtype}")
    B, V, H, W = img.shape
    nc = 126 if V == 5 else 35
    if tuple(coeffs.shape) != (B, 3, nc):
        raise ValueError(f"coeffs must be [B={B},3,{nc}], got {tuple(coeffs.shape)}")
    img, c = img.contiguous(), _coeffs32(coeffs, pairs=False)  # poly_layer_kernel reads scalars
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=img.device)
    _lib.check(lib.curl_poly_layer_f32(img.data_ptr(), c.data_ptr(), out.data_ptr(), B, H, W, V, _stream(img)),
               "curl_poly_layer_f32")
    return out


# ------------------------------------------------------------------ layout edges
@_one_device
def u8hwc_to_f32chw(x):
    """uint8 [B,H,W,3|4] (or [H,W,C]) -> float32 [B,3,H,W] = value/255 (infer.py:35-40, transpose.py:19-31)."""
    lib = _lib.load()
    _need_device(x, "x")
    if x.dtype != torch.uint8:
        raise TypeError(f"expected uint8, got {x.dtype}")
    squeeze = x.dim() == 3
    if squeeze:
        x = x.unsqueeze(0)
    if x.dim() != 4 or x.shape[3] not in (3, 4):
        raise ValueError(f"expected [B,H,W,3|4], got {tuple(x.shape)}")
    x = x.contiguous()
    B, H, W, C = x.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
    _lib.check(lib.curl_u8hwc_to_f32chw(x.data_ptr(), out.data_ptr(), B, H, W, C, _stream(x)), "curl_u8hwc_to_f32chw")
    return out[0] if squeeze else out


@_one_device
def f32chw_to_u8hwc(x):
    """float32 [B,3,H,W] (or [3,H,W]) -> uint8 [B,H,W,3], (x*255) TRUNCATED (evaluate.py:64-66)."""
    lib = _lib.load()
    squeeze = isinstance(x, torch.Tensor) and x.dim() == 3
    if squeeze:
        x = x.unsqueeze(0)
    x = _image(x, "x")
    B, _, H, W = x.shape
    out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=x.device)
    _lib.check(lib.curl_f32chw_to_u8hwc(x.data_ptr(), out.data_ptr(), B, H, W, _stream(x)), "curl_f32chw_to_u8hwc")
    return out[0] if squeeze else out


@_one_device
def compose_white_u8hwc(x, mask):
    """infer.py:46-47 in one pass: x*mask + (1-mask), then (.*255) truncated to uint8, CHW -> HWC."""
    lib = _lib.load()
    squeeze = isinstance(x, torch.Tensor) and x.dim() == 3
    if squeeze:
        x = x.unsqueeze(0)
        mask = mask.unsqueeze(0) if mask.dim() == 3 else mask
    x = _image(x, "x")
    m, kind = _mask(mask, x)
    if m is None:
        raise ValueError("compose_white_u8hwc needs a mask")
    B, _, H, W = x.shape
    out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=x.device)
    _lib.check(lib.curl_compose_white_u8hwc(x.data_ptr(), m.data_ptr(), kind, out.data_ptr(), B, H, W, _stream(x)),
               "curl_compose_white_u8hwc")
    return out[0] if squeeze else out


def _bytes_image(x, name="img"):
    _need_device(x, name)
    if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
        raise ValueError(f"{name} must be uint8 [B,H,W,3] (PIL's RGB layout), got {x.dtype} {tuple(x.shape)}")
    return x.contiguous()


def _white(white_mask, img_u8):
    if white_mask is None:
        return None
    _need_device(white_mask, "white_mask")
    B, H, W, _ = img_u8.shape
    if white_mask.dtype != torch.uint8 or tuple(white_mask.shape) != (B, H, W):
        raise ValueError(f"white_mask must be uint8 [B,H,W] ('L' image), got {white_mask.dtype} {tuple(white_mask.shape)}")
    return white_mask.contiguous()


def trispace_forward_u8hwc(img_u8, coeffs, white_mask=None):
    """infer.py:35-47 on the file's own bytes, one launch: byte/255 -> generate_residual + generate_image ->
    [out*m + (1-m), m = white_mask/255] -> truncating *255.  img_u8 [B,H,W,3] uint8 -> [B,H,W,3] uint8.
    (Plain calls through the compiled binding, as curl_layer_forward.)"""
    fast = _lib.fast() if type(img_u8) is torch.Tensor else None
    if fast is not None:
        r = fast.trispace_fwd_u8hwc(img_u8, coeffs, white_mask)
        if r is not None:
            if r.__class__ is int:
                _lib.check(r, "curl_trispace_fwd_u8hwc")
            return r
    return _trispace_forward_u8hwc_checked(img_u8, coeffs, white_mask=white_mask)


@_one_device
def _trispace_forward_u8hwc_checked(img_u8, coeffs, white_mask=None):
    lib = _lib.load()
    x = _bytes_image(img_u8)
    B, H, W, _ = x.shape
    _need_device(coeffs, "coeffs")
    if coeffs.dim() != 4 or coeffs.shape[:3] != (B, 3, 3) or coeffs.shape[3] not in (126, 35):
        raise ValueError(f"coeffs must be [B={B},3,3,126|35], got {tuple(coeffs.shape)}")
    c = _coeffs32(coeffs, pairs=coeffs.shape[3] == 126)
    wm = _white(white_mask, x)
    out = torch.empty_like(x)
    rc = lib.curl_trispace_fwd_u8hwc(x.data_ptr(), c.data_ptr(), _ptr(wm), out.data_ptr(), B, H, W, c.shape[3], 0,
                                     _stream(x))
    _lib.check(rc, "curl_trispace_fwd_u8hwc")
    return out


@_one_device
def curl_layer_forward_u8hwc(img_u8, mask, L, R, H, white_mask=None):
    """CURLLayer.forward between byte images: byte/255 -> the layer (mask [B,1,H,W] as in curl_layer_forward) ->
    [white background] -> truncating *255.  -> (uint8 [B,H,W,3], reg [B])."""
    lib = _lib.load()
    x = _bytes_image(img_u8)
    B, Hh, W, _ = x.shape
    Lc, Kl = _knots(L, "L", 3, B)
    Rc, Kr = _knots(R, "R", 3, B)
    Hc, Kh = _knots(H, "H", 4, B)
    m, kind = _mask(mask, SimpleNamespace(shape=(B, 3, Hh, W)))
    wm = _white(white_mask, x)
    out = torch.empty_like(x)
    reg = torch.empty(B, dtype=torch.float32, device=x.device)
    ws, nbytes = _workspace(B, Lc.shape[1] + Rc.shape[1] + Hc.shape[1], x.device)
    rc = lib.curl_layer_fwd_u8hwc(x.data_ptr(), _ptr(m), kind, Lc.data_ptr(), Rc.data_ptr(), Hc.data_ptr(), _ptr(wm),
                                  out.data_ptr(), reg.data_ptr(), ws.data_ptr(), nbytes, B, Hh, W, Kl, Kr, Kh, 0,
                                  _stream(x))
    _lib.check(rc, "curl_layer_fwd_u8hwc")
    return out, reg


def _planes(t, name):
    _need_device(t, name)
    if t.dim() != 4 or t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32 [B,C,H,W], got {t.dtype} {tuple(t.shape)}")
    return t.contiguous()


@_one_device
def msssim_stats(a, b, window_size=11):
    """MSSSIMMetric.compute_ssim over the five pyramid levels (metric.py:120-166,185-192) in five launches.
    a, b [B,C,H,W] -> (ssims [B,5], mcs [B,5]): per level, the per-image means of the SSIM and contrast-structure maps."""
    lib = _lib.load()
    a, b = _planes(a, "img1"), _planes(b, "img2")
    if a.shape != b.shape:
        raise RuntimeError(f"Input images must have the same shape ({tuple(a.shape)} vs. {tuple(b.shape)}).")
    B, C, H, W = a.shape
    nbytes = lib.curl_msssim_scratch_bytes(B, C, H, W)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
    ssims = torch.empty(B, 5, dtype=torch.float32, device=a.device)
    mcs = torch.empty_like(ssims)
    rc = lib.curl_msssim_fwd_f32(a.data_ptr(), b.data_ptr(), ssims.data_ptr(), mcs.data_ptr(), scratch.data_ptr(), nbytes,
                                 B, C, H, W, window_size, _stream(a))
    _lib.check(rc, "curl_msssim_fwd_f32")
    return ssims, mcs


@_one_device
def msssim_stats_backward(a, b, g_ssims, g_mcs, window_size=11):
    """d loss / d a of msssim_stats, given d loss / d ssims and d loss / d mcs ([B,5] each)."""
    lib = _lib.load()
    a, b = _planes(a, "img1"), _planes(b, "img2")
    B, C, H, W = a.shape
    gs = g_ssims.to(torch.float32).contiguous()
    gc = g_mcs.to(torch.float32).contiguous()
    nbytes = lib.curl_msssim_scratch_bytes(B, C, H, W)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
    grad = torch.empty_like(a)
    rc = lib.curl_msssim_bwd_f32(a.data_ptr(), b.data_ptr(), gs.data_ptr(), gc.data_ptr(), grad.data_ptr(),
                                 scratch.data_ptr(), nbytes, B, C, H, W, window_size, _stream(a))
    _lib.check(rc, "curl_msssim_bwd_f32")
    return grad


@_one_device
def psnr_per_image(a, b, mask=None, max_intensity=1.0):
    """metric.py:35-62 per image: masked PSNR [B] (NaN where an image has no unmasked pixel or zero error -> inf)."""
    lib = _lib.load()
    a, b = _image(a, "a"), _image(b, "b")
    if a.shape != b.shape:
        raise ValueError("a and b must have the same shape")
    m, kind = _mask(mask, a)
    B, _, H, W = a.shape
    out = torch.empty(B, dtype=torch.float32, device=a.device)
    nbytes = lib.curl_psnr_scratch_bytes(B, H, W)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
    rc = lib.curl_psnr_f32(a.data_ptr(), b.data_ptr(), _ptr(m), kind, out.data_ptr(), scratch.data_ptr(), nbytes,
                           B, H, W, float(max_intensity), _stream(a))
    _lib.check(rc, "curl_psnr_f32")
    return out


@_one_device
def loss_term_sums(pred, target, mask, want_L=True):
    """Per-image sums of the CURLLoss pointwise terms (model.py:89-109): [B,5] float64 =
    (sum|p-t|, sum cos_sim, sum|lab|, sum|hsv cone|, sum mask), plus the clamped L planes for MS-SSIM."""
    lib = _lib.load()
    pred, target = _image(pred, "pred"), _image(target, "target")
    if pred.shape != target.shape:
        raise ValueError("pred and target must have the same shape")
    m, kind = _mask(mask, pred)
    B, _, H, W = pred.shape
    sums = torch.empty(B, 5, dtype=torch.float64, device=pred.device)
    Lp = torch.empty(B, 1, H, W, dtype=torch.float32, device=pred.device) if want_L else None
    Lt = torch.empty_like(Lp) if want_L else None
    nbytes = lib.curl_loss_terms_scratch_bytes(B, H, W)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=pred.device)
    rc = lib.curl_loss_terms_f32(pred.data_ptr(), target.data_ptr(), _ptr(m), kind, sums.data_ptr(), _ptr(Lp), _ptr(Lt),
                                 scratch.data_ptr(), nbytes, B, H, W, _stream(pred))
    _lib.check(rc, "curl_loss_terms_f32")
    return sums, Lp, Lt


@_one_device
def layer_loss_forward(img, mask, L, R, H, target, want_L=True):
    """The train step's forward in one pass (main.py:283-285): CURLLayer.forward (model.py:137-176) and CURLLoss' pointwise
    terms (model.py:89-109) on the prediction while it is in registers -- curl_layer_loss_fwd_f32.
    -> (out, reg[B], sums [B,5] float64, L_pred, L_target, workspace): what curl_layer_forward(..., return_workspace=True) and
    loss_term_sums(out, target, mask) return, the same bits, with one mask for both."""
    lib = _lib.load()
    img, target = _image(img), _image(target, "target")
    if img.shape != target.shape:
        raise ValueError("img and target must have the same shape")
    B, _, Hh, W = img.shape
    Lc, Kl = _knots(L, "L", 3, B)
    Rc, Kr = _knots(R, "R", 3, B)
    Hc, Kh = _knots(H, "H", 4, B)
    m, kind = _mask(mask, img)
    out = torch.empty_like(img)
    reg = torch.empty(B, dtype=torch.float32, device=img.device)
    sums = torch.empty(B, 5, dtype=torch.float64, device=img.device)
    Lp = torch.empty(B, 1, Hh, W, dtype=torch.float32, device=img.device) if want_L else None
    Lt = torch.empty_like(Lp) if want_L else None
    ws, nbytes = _workspace(B, Lc.shape[1] + Rc.shape[1] + Hc.shape[1], img.device)
    sbytes = lib.curl_loss_terms_scratch_bytes(B, Hh, W)
    scratch = torch.empty(sbytes // 4, dtype=torch.float32, device=img.device)
    rc = lib.curl_layer_loss_fwd_f32(img.data_ptr(), _ptr(m), kind, Lc.data_ptr(), Rc.data_ptr(), Hc.data_ptr(), target.data_ptr(),
                                     out.data_ptr(), reg.data_ptr(), sums.data_ptr(), _ptr(Lp), _ptr(Lt), ws.data_ptr(), nbytes,
                                     scratch.data_ptr(), sbytes, B, Hh, W, Kl, Kr, Kh, 0, _stream(img))
    _lib.check(rc, "curl_layer_loss_fwd_f32")
    return out, reg, sums, Lp, Lt, ws


@_one_device
def loss_terms_backward(pred, target, mask, weights, grad_L_pred=None):
    """d(sum_k weights[k] * sum_k-th pointwise sum + <grad_L_pred, L_pred>) / d pred.  weights: device float32 [4]."""
    lib = _lib.load()
    pred, target = _image(pred, "pred"), _image(target, "target")
    m, kind = _mask(mask, pred)
    B, _, H, W = pred.shape
    w = weights.to(torch.float32).contiguous()
    g = None if grad_L_pred is None else grad_L_pred.to(torch.float32).contiguous()
    out = torch.empty_like(pred)
    rc = lib.curl_loss_terms_bwd_f32(pred.data_ptr(), target.data_ptr(), _ptr(m), kind, w.data_ptr(), _ptr(g),
                                     out.data_ptr(), B, H, W, _stream(pred))
    _lib.check(rc, "curl_loss_terms_bwd_f32")
    return out
