"""CPU oracle for the CURL colour-curve hot path.  TEST INFRASTRUCTURE ONLY.

This file is a torch-eager CPU restatement of the arithmetic of the reference
(danielbulhosa/CURL) for the path SURVEY.md section 8 scopes: curve application
(curves.py), the four colour-space converters (colors.py), the layout swaps
(transpose.py), the CURLLayer stage order (model.py:137-176) and the masked
PSNR (metric.py:35-68).  It exists to CHECK the HIP kernels.  Only tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py may import it;
nothing under curl_amd/ does, and the product fails loudly without its HIP
library instead of falling back to this code.

Parity pin: tests/test_oracle_vs_reference.py compares every function here
bit-for-bit with the reference's own modules imported from /root/reference
(possible only in the build container), and tests/golden/*.npz holds outputs
of the reference itself (generator: tests/golden/make_golden.py) that this
file and the HIP path are both checked against wherever the reference cannot
travel.

Semantics where the reference is broken as written:
  * adjust_rgb/adjust_lab/adjust_hsv seed the regulariser with zeros(B)
    (curves.py:56,111,155 seed None, which raises TypeError);
  * curl_layer skips the three dead `feat` lines (model.py:152,158,164).

Every function works in the dtype of its input (float32 = the reference's
arithmetic; float64 gives a high-precision yardstick for judging rounding
noise at the reference's own discontinuities).
"""
import math

import numpy as np
import torch

# ---------------------------------------------------------------------------
# constants (colors.py:8-25, 69-86)
# ---------------------------------------------------------------------------
_RGB_FROM_ROWS_TO_XYZ = [  # rows = R,G,B ; columns = X,Y,Z   (colors.py:10-12)
    [0.412453, 0.212671, 0.019334],
    [0.357580, 0.715160, 0.119193],
    [0.180423, 0.072169, 0.950227],
]
_F_ROWS_TO_LAB = [  # rows = fx,fy,fz ; columns = L,a,b   (colors.py:18-20)
    [0.0, 500.0, 0.0],
    [116.0, -500.0, 200.0],
    [0.0, 0.0, -200.0],
]
_XYZ_ROWS_TO_RGB = [  # rows = X,Y,Z ; columns = R,G,B   (colors.py:71-73)
    [3.2404542, -0.9692660, 0.0556434],
    [-1.5371385, 1.8760108, -0.2040259],
    [-0.4985314, 0.0415560, 1.0572252],
]
_LAB_ROWS_TO_F = [  # rows = L+16,a,b ; columns = fx,fy,fz   (colors.py:79-81)
    [1 / 116.0, 1 / 116.0, 1 / 116.0],
    [1 / 500.0, 0, 0],
    [0, 0, -1 / 200.0],
]
_D65 = [0.950456, 1.0, 1.088754]  # colors.py:24,85
_LAB_OFFSET = [16.0, 0.0, 0.0]  # colors.py:25,86
_EPS = 6 / 29  # colors.py:43,108


def _param(rows, like, transpose=True):
    """A constant as the reference holds it: built in float32 (colors.py:13,21,74,82),
    then cast to the working dtype, laid out (1,1,3,3) after a transpose."""
    t = torch.tensor(rows, dtype=torch.float)
    if transpose:
        t = t.transpose(1, 0)[None, None]
    else:
        t = t.reshape(1, 3, 1, 1)
    return t.to(like.dtype)


def _chan_mix(img, mat):
    # colors.py:40,50,104,117 -- same einsum so the reduction order is the reference's.
    return torch.einsum('bcyx,bykc->bkyx', img, mat)


# ---------------------------------------------------------------------------
# curves.py
# ---------------------------------------------------------------------------
def curve_regulariser(C):
    """Sum of squared second differences of the knots, per image (curves.py:19,24)."""
    seg = C[:, 1:] - C[:, :-1]
    return ((seg[:, 1:] - seg[:, :-1]) ** 2).sum(1)


def curve_scale(x, C):
    """Per-pixel multiplier of curves.py:29-32.  x: [B,H,W], C: [B,K] (already exp'd).

    scale = C0 + sum_{j=0}^{K-3} slope_j * (S*x - j),  S = K-1, no clamp of (S*x - j):
    the function is affine in x and the last knot never touches a pixel.
    """
    n_seg = C.shape[1] - 1
    seg = C[:, 1:] - C[:, :-1]
    used = seg[:, :-1]
    idx = torch.arange(0, seg.shape[1] - 1)
    lifted = n_seg * x.unsqueeze(1) - idx.reshape(1, -1, 1, 1)
    return C[:, 0].reshape(-1, 1, 1) + (used.reshape(used.shape[0], used.shape[1], 1, 1) * lifted).sum(1)


def apply_curve(img, C, reg, channel_in, channel_out):
    """curves.py:4-38.  Returns (new image, reg); reg is updated IN PLACE like the
    reference does with `slope_sqr_diff +=` (curves.py:24)."""
    reg += curve_regulariser(C)
    scale = curve_scale(img[:, channel_in], C)
    out = img.clone()
    out[:, channel_out] = img[:, channel_out] * scale
    out = torch.clamp(out, 0.0, 1.0)  # the WHOLE image, all channels (curves.py:36)
    return out, reg


def _adjust(img, raw, pairs):
    """Shared body of curves.py:41-87, 90-133, 136-180 with the regulariser seeded at 0."""
    img = img.contiguous()
    knots = [torch.exp(p) for p in torch.chunk(raw, len(pairs), dim=1)]
    reg = torch.zeros(img.shape[0], dtype=img.dtype)
    for C, (cin, cout) in zip(knots, pairs):
        img, reg = apply_curve(img, C, reg, cin, cout)
    return img.contiguous(), reg


def adjust_rgb(img, R):
    """curves.py:90-133: R->R, G->G, B->B."""
    return _adjust(img, R, [(0, 0), (1, 1), (2, 2)])


def adjust_lab(img, L):
    """curves.py:136-180: L->L, a->a, b->b (on Lab normalised to [0,1])."""
    return _adjust(img, L, [(0, 0), (1, 1), (2, 2)])


def adjust_hsv(img, S):
    """curves.py:41-87: H->H, H->S (adjusted hue), S->S, V->V."""
    return _adjust(img, S, [(0, 0), (0, 1), (1, 1), (2, 2)])


# ---------------------------------------------------------------------------
# colors.py
# ---------------------------------------------------------------------------
def rgb2lab(img):
    """colors.py:27-62."""
    img = img.contiguous()
    lo = img.le(0.04045).to(img.dtype)
    hi = img.gt(0.04045).to(img.dtype)
    img = (img / 12.92) * lo + (((torch.clamp(img, min=0.0001) + 0.055) / 1.055) ** 2.4) * hi
    img = _chan_mix(img, _param(_RGB_FROM_ROWS_TO_XYZ, img))
    img = torch.mul(img, 1 / _param(_D65, img, transpose=False))
    lo = img.le(_EPS ** 3).to(img.dtype)
    hi = img.gt(_EPS ** 3).to(img.dtype)
    img = ((img / (3.0 * _EPS ** 2) + 4.0 / 29.0) * lo) + (torch.clamp(img, min=0.0001) ** (1.0 / 3.0) * hi)
    img = _chan_mix(img, _param(_F_ROWS_TO_LAB, img)) - _param(_LAB_OFFSET, img, transpose=False)
    img[:, 0] = img[:, 0] / 100
    img[:, 1] = (img[:, 1] / 110 + 1) / 2
    img[:, 2] = (img[:, 2] / 110 + 1) / 2
    return img.contiguous()


def lab2rgb(img):
    """colors.py:88-123.  Output is NOT clamped."""
    img = img.contiguous()
    c0 = (img[:, 0] * 100).unsqueeze(1)
    c1 = (((img[:, 1] * 2) - 1) * 110).unsqueeze(1)
    c2 = (((img[:, 2] * 2) - 1) * 110).unsqueeze(1)
    img = torch.cat([c0, c1, c2], dim=1)
    img = _chan_mix(img + _param(_LAB_OFFSET, img, transpose=False), _param(_LAB_ROWS_TO_F, img))
    lo = img.le(_EPS).to(img.dtype)
    hi = img.gt(_EPS).to(img.dtype)
    img = ((3.0 * _EPS ** 2 * (img - 4.0 / 29.0)) * lo) + ((torch.clamp(img, min=0.0001) ** 3.0) * hi)
    img = torch.mul(img, _param(_D65, img, transpose=False))
    img = _chan_mix(img, _param(_XYZ_ROWS_TO_RGB, img))
    lo = img.le(0.0031308).to(img.dtype)
    hi = img.gt(0.0031308).to(img.dtype)
    img = (img * 12.92 * lo) + ((torch.clamp(img, min=0.0001) ** (1 / 2.4) * 1.055) - 0.055) * hi
    return img.contiguous()


def _inv_or_zero(t):
    """colors.py:186-193: 1/t where t != 0, else 0."""
    out = 0.0 * t
    nz = t != 0
    out[nz] = 1 / (t[nz])
    return out


def rgb2hsv(img):
    """colors.py:195-242.  Hue terms ADD on channel ties; lower clamp is 1e-9."""
    img = torch.clamp(img, 10 ** (-9), 1.0).contiguous()
    zero = torch.tensor(0.0, dtype=img.dtype)
    mx = torch.max(img, 1)[0]
    mn = torch.min(img, 1)[0]
    df = torch.add(mx, -1.0 * mn)
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    df_inv = _inv_or_zero(df)
    sextant = ((g - b) * df_inv) * r.eq(mx).to(img.dtype) \
        + (2.0 + (b - r) * df_inv) * g.eq(mx).to(img.dtype) \
        + (4.0 + (r - g) * df_inv) * b.eq(mx).to(img.dtype)
    img[:, 0] = torch.where(df == 0.0, zero, sextant)  # overwrites r: g, b views stay valid
    img[:, 0] = img[:, 0] * 60.0
    img[:, 0] = img[:, 0].lt(0.0).to(img.dtype) * (img[:, 0] + 360) + img[:, 0].ge(0.0).to(img.dtype) * (img[:, 0])
    img[:, 0] = img[:, 0] / 360
    mx_inv = _inv_or_zero(mx)
    img[:, 1] = torch.where(mx == 0.0, zero,
                            mx.ne(0.0).to(img.dtype) * (df * mx_inv) + mx.eq(0.0).to(img.dtype) * (0.0))
    img[:, 2] = mx
    return torch.clamp(img, 10 ** (-9), 1.0)


def hsv2rgb(img):
    """colors.py:131-177."""
    img = torch.clamp(img, 0.0, 1.0)
    h360 = img[:, 0] * 360
    s = img[:, 1]
    v = img[:, 2]

    def ramp(start, width):
        return torch.clamp(h360 - start, 0.0, width)

    # colors.py:143-150 (the identically-zero m1,m3,m5 terms add +0 and are kept for the
    # evaluation order of the sum)
    m2 = (v * (1 - s) - v) / 60
    m4 = -1 * m2
    r = v + ramp(0, 60.0) * 0 + ramp(60, 60.0) * m2 + ramp(120, 120.0) * 0 + ramp(240, 60.0) * m4 + ramp(300, 60.0) * 0
    # colors.py:153-159
    m1 = (v - v * (1 - s)) / 60
    m3 = -1 * m1
    g = v * (1 - s) + ramp(0, 60.0) * m1 + ramp(60, 120.0) * 0 + ramp(180, 60.0) * m3 + ramp(240, 120.0) * 0
    # colors.py:162-168
    m2 = (v - v * (1 - s)) / 60
    m4 = -1 * m2
    b = v * (1 - s) + ramp(0, 120.0) * 0 + ramp(120, 60.0) * m2 + ramp(180, 120.0) * 0 + ramp(300, 60.0) * m4
    out = torch.stack((r, g, b), 1).contiguous()
    return torch.clamp(out, 0.0, 1.0)


# ---------------------------------------------------------------------------
# model.py: CURLLayer stage order
# ---------------------------------------------------------------------------
def lab_stage(img, mask, L, num_lab_points=48):
    """First stage of CURLLayer.forward, closed back to RGB (model.py:151-157):
    RGB -> Lab -> 3 curves -> * mask -> RGB.  Returns (rgb, reg_lab)."""
    lab = rgb2lab(img)
    lab, reg = adjust_lab(lab, L[:, :num_lab_points])
    lab = lab * mask
    return lab2rgb(lab), reg


def hsv_stage(img, mask, H, num_hsv_points=64):
    """Third stage of CURLLayer.forward from RGB to its RGB residual (model.py:163-169):
    RGB -> HSV -> 4 curves -> * mask -> RGB.  Returns (residual_rgb, reg_hsv)."""
    hsv = rgb2hsv(img)
    hsv, reg = adjust_hsv(hsv, H[:, :num_hsv_points])
    hsv = hsv * mask
    return hsv2rgb(hsv), reg


def curl_layer(img, mask, L, R, H, num_lab_points=48, num_rgb_points=48, num_hsv_points=64):
    """CURLLayer.forward (model.py:137-176) minus the dead `feat` lines."""
    rgb, reg_lab = lab_stage(img, mask, L, num_lab_points)
    rgb, reg_rgb = adjust_rgb(rgb, R[:, :num_rgb_points])
    rgb = rgb * mask
    hsv = rgb2hsv(rgb)
    hsv, reg_hsv = adjust_hsv(hsv, H[:, :num_hsv_points])
    hsv = hsv * mask
    residual = hsv2rgb(hsv)
    out = torch.clamp(img + residual, 0.0, 1.0) * mask
    return out, (reg_rgb + reg_lab + reg_hsv)


def split_knots(flat, num_lab_points=48, num_rgb_points=48):
    """GCURLNet.forward's split of the encoder output (model.py:186-187,197-199)."""
    b1 = num_lab_points
    b2 = num_lab_points + num_rgb_points
    return flat[:, :b1], flat[:, b1:b2], flat[:, b2:]


# ---------------------------------------------------------------------------
# metric.py: masked PSNR
# ---------------------------------------------------------------------------
def psnr(a, b, mask, max_intensity=1.0):
    """metric.py:35-68: per-image masked PSNR, nan-mean over the batch."""
    a = torch.clamp(a, 0.0, 1.0) * mask
    b = torch.clamp(b, 0.0, 1.0) * mask
    n = a.shape[1] * torch.squeeze(mask, dim=1).sum(dim=(1, 2))
    mse = ((a - b) ** 2).sum(dim=(1, 2, 3)) / n
    val = (10 * torch.log10(max_intensity ** 2 / mse)).nanmean()
    return None if val.isnan() else val


# ---------------------------------------------------------------------------
# transpose.py / evaluate.py:64 / infer.py:37-47 : layout edges
# ---------------------------------------------------------------------------
def chw_to_hwc(arr):
    """transpose.py:4-16 (3-D and 4-D numpy arrays; returns a view)."""
    if arr.ndim == 3:
        return np.transpose(arr, (1, 2, 0))
    if arr.ndim == 4:
        return np.transpose(arr, (0, 2, 3, 1))
    return None


def hwc_to_chw(arr):
    """transpose.py:19-31."""
    if arr.ndim == 3:
        return np.transpose(arr, (2, 0, 1))
    if arr.ndim == 4:
        return np.transpose(arr, (0, 3, 1, 2))
    return None


def u8hwc_to_f32chw(arr_u8):
    """What PIL + TF.to_tensor do at the file edge (infer.py:37, data.py:133-158):
    uint8 HWC (3 or 4 channels, alpha dropped) -> float32 CHW, value/255."""
    a = np.asarray(arr_u8)[..., :3]
    t = torch.from_numpy(np.ascontiguousarray(hwc_to_chw(a)))
    return t.to(torch.float32).div(255)


def f32chw_to_u8hwc(t):
    """evaluate.py:64-66: (x*255).astype('uint8') -- TRUNCATION, then CHW->HWC.
    Values are expected in [0,1] (the layer clamps)."""
    a = (t.numpy() * 255).astype('uint8')
    return np.ascontiguousarray(chw_to_hwc(a))


def white_background(img, mask):
    """infer.py:46: out*mask + (1-mask)."""
    return img * mask + (1 - mask)


def affine_coefficients(C):
    """Exact-arithmetic collapse of curve_scale: scale(x) = a + b*x with
    a = C0 - sum_j j*slope_j, b = S*sum_j slope_j (j = 0..K-3), evaluated in float64
    from the float32 slopes.  Used by tests to bound |affine - in-order fp32| noise;
    it is the form the fused HIP kernels evaluate."""
    C32 = C.to(torch.float32)
    seg = (C32[:, 1:] - C32[:, :-1]).to(torch.float64)[:, :-1]
    j = torch.arange(seg.shape[1], dtype=torch.float64)
    S = float(C.shape[1] - 1)
    a = C32[:, 0].to(torch.float64) - (seg * j).sum(1)
    b = S * seg.sum(1)
    return a, b


# ---------------------------------------------------------------------------
# model.py:206-415, 499-520 : the polynomial layers and TriSpaceRegNet's per-pixel residual
# ---------------------------------------------------------------------------
def poly_powers(degree, num_variables):
    """Exponent table in the order of ChannelPolyLayer.generate_powers (model.py:222-246): graded, and inside
    one degree lexicographic with variable 0 most significant.  [num_coeffs, num_variables]."""
    import itertools
    rows = []
    for total in range(degree + 1):
        ts = [t for t in itertools.product(range(total + 1), repeat=num_variables) if sum(t) == total]
        ts.sort(reverse=True)
        rows.extend(ts)
    return torch.tensor(rows, dtype=torch.float)


def channel_poly_layer(img, coeffs, degree):
    """ChannelPolyLayer.forward (model.py:295-333): img [B,V,H,W], coeffs [B,num_out,num_coeffs]."""
    V = img.shape[1]
    pw = poly_powers(degree, V).to(img.dtype)
    n = pw.shape[0]
    img_us = torch.unsqueeze(img, dim=0)
    terms = torch.permute(torch.pow(img_us, pw.reshape(n, 1, V, 1, 1)), [1, 2, 3, 4, 0]).prod(dim=1)
    return (coeffs.reshape(img.shape[0], coeffs.shape[1], 1, 1, n) * torch.unsqueeze(terms, dim=1)).sum(dim=-1)


def deg4_mobile_poly_terms(img):
    """Deg4MobilePolyLayer.poly_terms (model.py:346-397) for img [B,5,H,W]: the 126 monomials built from
    explicit products / `**k` (torch evaluates x**2 as x*x, x**3 as x*x*x, x**4 through pow), last dim = term."""
    x = torch.unsqueeze(img, dim=-1)
    v = [x[:, i] for i in range(5)]
    cols = []
    for row in poly_powers(4, 5).to(torch.int64).tolist():
        factors = [v[i] if p == 1 else v[i] ** p for i, p in enumerate(row) if p > 0]
        if not factors:
            cols.append(1.0 + v[0] * 0.0)
            continue
        t = factors[0]
        for f in factors[1:]:
            t = t * f
        cols.append(t)
    return torch.cat(cols, dim=-1)


def deg4_mobile_poly_layer(img, coeffs):
    """Deg4MobilePolyLayer.forward (model.py:399-415): 5 variables, degree 4, 3 outputs."""
    terms = deg4_mobile_poly_terms(img)
    return (coeffs.reshape(img.shape[0], 3, 1, 1, 126) * torch.unsqueeze(terms, dim=1)).sum(dim=-1)


def cat_coords(img, spatial=True, rows=None):
    """TriSpaceRegNet.cat_coords (model.py:487-497): append x/width and y/height planes.
    rows=(row0, H_full) (tests only): `img` is the band of rows [row0, row0 + img.shape[2]) of an image H_full rows
    high -- the y plane then carries the FULL image's row / height, as the reference would compute for those rows."""
    if not spatial:
        return img
    B, _, H, W = img.shape
    row0, H_full = (0, H) if rows is None else rows
    zeros = img[:, 0:1] * 0.0
    x = zeros + torch.arange(0, W).reshape(1, 1, 1, W) / W
    y = zeros + torch.arange(row0, row0 + H).reshape(1, 1, H, 1) / H_full
    return torch.cat([img, x, y], dim=1)


def trispace_residual(img, R, L, H, spatial=True, mobile=True, rows=None):
    """TriSpaceRegNet.generate_residual (model.py:499-515).  R, L, H: [B,3,num_coeffs].  rows: see cat_coords."""
    def poly(x, c):
        if spatial and mobile:
            return deg4_mobile_poly_layer(x, c)
        return channel_poly_layer(x, c, 4)
    rgb_res = torch.sigmoid(poly(cat_coords(img, spatial, rows), R))
    lab_res = lab2rgb(torch.sigmoid(poly(cat_coords(rgb2lab(img), spatial, rows), L)))
    hsv_res = hsv2rgb(torch.sigmoid(poly(cat_coords(rgb2hsv(img), spatial, rows), H)))
    rgb_res = 2 * (rgb_res - 0.5)
    lab_res = 2 * (lab_res - 0.5)
    hsv_res = 2 * (hsv_res - 0.5)
    return rgb_res + lab_res + hsv_res


def generate_image(img, residual):
    """TriSpaceRegNet.generate_image (model.py:517-520)."""
    return torch.clamp(img + residual, 0.0, 1.0)


# ---------------------------------------------------------------------------
# model.py:78-116 : the per-pixel terms of CURLLoss
# ---------------------------------------------------------------------------
def _hsv_cone(img):
    """CURLLoss.batch_hsv_convert (model.py:62-76)."""
    hsv = torch.clamp(rgb2hsv(img), 0.0, 1.0)
    hue = 2 * math.pi * hsv[:, 0]
    val, sat = hsv[:, 2], hsv[:, 1]
    return torch.stack((val * sat * torch.cos(hue), val * sat * torch.sin(hue), val), 1)


def curl_loss_terms(pred, target, mask):
    """CURLLoss.forward (model.py:89-109) without the MS-SSIM term: returns
    (rgb_loss, cosine_rgb_loss, lab_l1_loss, hsv_loss, L_pred, L_target)."""
    import torch.nn.functional as F
    unmasked = pred.shape[1] * mask.sum()
    p, t = pred * mask, target * mask
    rgb = F.l1_loss(p, t, reduction='sum') / unmasked
    base = F.cosine_similarity(p, t, dim=1)
    cosine = (1.0 - (base + torch.logical_not(mask)).mean(dim=(1, 2))).mean()
    lab_t = torch.clamp(rgb2lab(t), 0.0, 1.0)
    lab_p = torch.clamp(rgb2lab(p), 0.0, 1.0)
    lab = F.l1_loss(lab_p, lab_t, reduction='sum') / unmasked
    hsv = F.l1_loss(_hsv_cone(p), _hsv_cone(t), reduction='sum') / unmasked
    return rgb, cosine, lab, hsv, lab_p[:, 0:1], lab_t[:, 0:1]


def curl_loss(pred, target, mask, ssim_loss_value):
    """model.py:111-116 given the MS-SSIM loss term."""
    rgb, cosine, lab, hsv, _, _ = curl_loss_terms(pred, target, mask)
    return (rgb + cosine + lab + hsv + 10 * ssim_loss_value) / 5


# ---------------------------------------------------------------- MS-SSIM (metric.py:75-211)
def msssim_window(window_size, num_channel, sigma=1.5):
    """MSSSIMMetric.create_window / gaussian (metric.py:90-118): normalised Gaussian, outer product,
    expanded to [C,1,ws,ws]."""
    from math import exp
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(num_channel, 1, window_size, window_size).contiguous()


def ssim_and_cs(img1, img2, window):
    """MSSSIMMetric.compute_ssim (metric.py:120-166): per-image mean SSIM and contrast-structure term."""
    import torch.nn.functional as F
    C, ws = window.shape[0], window.shape[-1]
    window = window.type_as(img1)
    conv = lambda x: F.conv2d(x, window, padding=ws // 2, groups=C)  # noqa: E731
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(img1 * img1) - mu1_sq
    sigma2_sq = conv(img2 * img2) - mu2_sq
    sigma12 = conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    cs = torch.mean((2.0 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2), dim=(1, 2, 3))
    return ssim_map.mean(dim=(1, 2, 3)), cs


def msssim(img1, img2, window_size=11, num_channel=3):
    """MSSSIMMetric.compute_msssim (metric.py:168-208): 5 levels, 2x2 average pooling between them,
    (x+1)/2 normalisation, prod(mcs[:-1]^w * ssim[-1]^w[-1])."""
    import torch.nn.functional as F
    weights = torch.tensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333], dtype=torch.float32)
    if img1.shape[2] != img2.shape[2]:
        img1 = img1.transpose(2, 3)
    window = msssim_window(window_size, num_channel)
    ssims, mcs = [], []
    for _ in range(weights.numel()):
        s, c = ssim_and_cs(img1, img2, window)
        ssims.append(s)
        mcs.append(c)
        img1, img2 = F.avg_pool2d(img1, (2, 2)), F.avg_pool2d(img2, (2, 2))
    ssims = (torch.stack(ssims, dim=1) + 1) / 2
    mcs = (torch.stack(mcs, dim=1) + 1) / 2
    w = weights.reshape(1, -1).to(img1.dtype)
    pow1, pow2 = mcs ** w, ssims ** w
    return torch.prod(pow1[:, :-1] * pow2[:, -1].reshape(-1, 1), dim=1)


def input_sensitivity(img, mask, L, R, H, r64=None, h=1e-6):
    """Conditioning of the chain at every pixel: max over the three input channels and both signs of
    |d out / d in|, by finite differences of curl_layer evaluated in float64 -- how much model.py:137-176 amplifies a
    rounding-sized perturbation there.  -> [B,H,W] float64.  The parity bound |HIP - ref| <= max(1e-5, 2e-6 * S) is
    stated on it (tests/test_gpu_parity.py, __graft_entry__.smoke, bench.py accuracy)."""
    img, mask, L, R, H = (t.double() for t in (img, mask, L, R, H))
    if r64 is None:
        r64, _ = curl_layer(img, mask, L, R, H)
    S = torch.zeros(r64.shape[0], r64.shape[2], r64.shape[3], dtype=torch.float64)
    for k in range(3):
        for sgn in (1.0, -1.0):
            p = img.clone()
            p[:, k] += sgn * h
            o, _ = curl_layer(p, mask, L, R, H)
            S = torch.maximum(S, (o - r64).abs().amax(1) / h)
    return S


def layer_gradients(img, mask, L, R, H, w, wr=None, dtype=torch.float64):
    """Autograd of curl_layer (the reference's arithmetic, model.py:137-176) for the loss sum(out * w) + sum(reg * wr),
    evaluated in `dtype`.  -> (d img, d L, d R, d H).  The float64 evaluation is the backward's parity yardstick."""
    x = img.detach().to(dtype).clone().requires_grad_(True)
    Lk, Rk, Hk = (t.detach().to(dtype).clone().requires_grad_(True) for t in (L, R, H))
    out, reg = curl_layer(x, mask.to(dtype), Lk, Rk, Hk)
    loss = (out * w.to(dtype)).sum()
    if wr is not None:
        loss = loss + (reg * wr.to(dtype)).sum()
    loss.backward()
    return x.grad, Lk.grad, Rk.grad, Hk.grad


def gradient_curvature(img, mask, L, R, H, w, g64=None, h=1e-6):
    """How fast the image gradient itself changes with the input, per pixel: max over the three channels and both signs of
    |d img(x +- h e_k) - d img(x)| / h, float64 -- the backward's counterpart of input_sensitivity.  A float32 evaluation of
    the chain sees its intermediates perturbed by roundings worth ~1e-6 of input; its gradient can be off by that times this
    curvature.  A pixel whose value h * curvature is a sizeable fraction of the gradient scale sits within h of a clamp gate, a
    threshold or a channel tie -- a DISCONTINUITY of the gradient: the exception set of tests/test_gpu_backward.py.
    -> [B,H,W] float64."""
    if g64 is None:
        g64 = layer_gradients(img, mask, L, R, H, w)[0]
    C = torch.zeros(g64.shape[0], g64.shape[2], g64.shape[3], dtype=torch.float64)
    for k in range(3):
        for sgn in (1.0, -1.0):
            p = img.double().clone()
            p[:, k] += sgn * h
            gp = layer_gradients(p, mask, L, R, H, w)[0]
            C = torch.maximum(C, (gp - g64).abs().amax(1) / h)
    return C
