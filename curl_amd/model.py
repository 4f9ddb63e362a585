"""Drop-in for the curve classes of the reference's model.py: CURLLayer (model.py:121-176) and
GCURLNet (model.py:179-203).

CURLLayer.forward is ONE fused HIP kernel over the pixels (plus a per-image knot-prep kernel);
the encoder of GCURLNet is stock PyTorch-ROCm convolutions, as BASELINE.json's north_star asks.
"""
import torch
import torch.nn as nn

from . import colors, metric, ops


class _CurlLayerFn(torch.autograd.Function):
    """Autograd node around the fused forward/backward kernels."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)  # under autocast: float32 in, autocast off
    def forward(ctx, img, mask, L, R, H):
        out, reg = ops.curl_layer_forward(img, mask, L, R, H)
        ctx.save_for_backward(img, L.contiguous(), R.contiguous(), H.contiguous())
        ctx.mask = mask
        return out, reg

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out, grad_reg):
        img, L, R, H = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0]
        g_img, gL, gR, gH = ops.curl_layer_backward(img, ctx.mask, L, R, H, grad_out.contiguous(), grad_reg, need_img)
        return g_img, None, gL, gR, gH


class CURLLayer(nn.Module):
    """model.py:121-176.  Same constructor arguments, same forward signature and returns.
    `paper_pwl=True` (not in the reference) evaluates the curves as the paper's clamped piecewise-linear
    interpolation of the knots (CURL_F_PWL: knots in LDS) instead of the reference's affine form; inference only."""

    def __init__(self, num_lab_points=48, num_rgb_points=48, num_hsv_points=64, paper_pwl=False):
        super().__init__()
        self.paper_pwl = paper_pwl
        self.num_lab_points = num_lab_points
        self.num_rgb_points = num_rgb_points
        self.num_hsv_points = num_hsv_points
        # kept so reference checkpoints load key for key (model.py:130-133); the fused kernel
        # bakes the same constants and never reads these parameters
        self.rgb2lab = colors.RGB2LAB()
        self.lab2rgb = colors.LAB2RGB()
        self.rgb2hsv = colors.RGB2HSV()
        self.hsv2rgb = colors.HSV2RGB()

    def forward(self, img, mask, L, R, H):
        """img [B,3,H,W] in [0,1]; mask [B,1,H,W] (bool or float) or None; L, R, H raw knots.
        Returns (img, gradient_regulariser[B])  (model.py:176).  The dead `feat` concatenations of
        model.py:152,158,164 (a NameError in the reference) are not part of the semantics."""
        L = L[:, :self.num_lab_points]  # model.py:153
        R = R[:, :self.num_rgb_points]  # model.py:159
        H = H[:, :self.num_hsv_points]  # model.py:165
        needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (img, L, R, H))
        if self.paper_pwl:
            if needs_grad:
                raise NotImplementedError("curl_amd: paper_pwl has no backward (the reference's curves are the affine form)")
            return ops.curl_layer_forward(img, mask, L, R, H, flags=ops.F_PWL)
        if needs_grad:
            return _CurlLayerFn.apply(img, mask, L, R, H)
        return ops.curl_layer_forward(img, mask, L, R, H)


def _conv_bn_act(cin, cout, k, stride, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout), nn.SiLU(inplace=True))


class _FusedMBConv(nn.Module):
    def __init__(self, cin, cout, stride, expand):
        super().__init__()
        mid = cin * expand
        self.use_res = stride == 1 and cin == cout
        self.body = nn.Sequential(_conv_bn_act(cin, mid, 3, stride),
                                  nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.body(x)
        return x + y if self.use_res else y


class _MBConv(nn.Module):
    def __init__(self, cin, cout, stride, expand):
        super().__init__()
        mid = cin * expand
        self.use_res = stride == 1 and cin == cout
        self.expand = _conv_bn_act(cin, mid, 1, 1)
        self.dw = _conv_bn_act(mid, mid, 3, stride, groups=mid)
        se = max(8, cin // 4)
        self.se = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(mid, se, 1), nn.SiLU(inplace=True),
                                nn.Conv2d(se, mid, 1), nn.Sigmoid())
        self.project = nn.Sequential(nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.dw(self.expand(x))
        y = self.project(y * self.se(y))
        return x + y if self.use_res else y


class CurveEncoder(nn.Module):
    """EfficientNetV2-style CNN with a `num_features`-wide pooled output and a `classifier` head,
    standing in for timm's `efficientnetv2_rw_s` (model.py:189; timm is not installed here and its
    pretrained weights need a download).  Stock PyTorch-ROCm ops only (MIOpen convolutions)."""

    def __init__(self, num_outputs=160, width=1.0, num_features=1792):
        super().__init__()

        def c(v):
            return max(8, int(v * width + 4) // 8 * 8)

        cfg = [  # (block, repeats, out, stride, expand)   ~ efficientnetv2_rw_s stages
            (_FusedMBConv, 2, c(24), 1, 1), (_FusedMBConv, 4, c(48), 2, 4), (_FusedMBConv, 4, c(64), 2, 4),
            (_MBConv, 6, c(128), 2, 4), (_MBConv, 9, c(160), 1, 6), (_MBConv, 15, c(272), 2, 6)]
        layers = [_conv_bn_act(3, c(24), 3, 2)]
        cin = c(24)
        for block, n, cout, stride, expand in cfg:
            for i in range(n):
                layers.append(block(cin, cout, stride if i == 0 else 1, expand))
                cin = cout
        layers.append(_conv_bn_act(cin, num_features, 1, 1))
        self.features = nn.Sequential(*layers)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.num_features = num_features
        self.classifier = nn.Linear(num_features, num_outputs)

    def forward(self, x):
        return self.classifier(self.pool(self.features(x)).flatten(1))


class GCURLNet(nn.Module):
    """model.py:179-203: encoder -> [B,160] raw knots -> split at 48 / 96 -> CURLLayer.

    The reference's constructor is broken (undefined self.num_spaces..., model.py:191) and needs a
    network download (model.py:189); here the head width is what the layer consumes
    (num_lab_points + num_rgb_points + num_hsv_points) and the backbone is injectable.
    `encoder_size` (optional): the encoder sees the image resized to this square while the curves are
    applied at full resolution -- the low-res-encode / full-res-apply shape of infer.py:32-44."""

    def __init__(self, num_lab_points=48, num_rgb_points=48, num_hsv_points=64, backbone=None, encoder_size=None):
        super().__init__()
        self.num_lab_points = num_lab_points
        self.num_rgb_points = num_rgb_points
        self.num_hsv_points = num_hsv_points
        self.curve_break_1 = num_lab_points
        self.curve_break_2 = num_lab_points + num_rgb_points
        n_out = num_lab_points + num_rgb_points + num_hsv_points
        if backbone is None:
            backbone = CurveEncoder(num_outputs=n_out)
        elif hasattr(backbone, "classifier") and isinstance(backbone.classifier, nn.Linear) \
                and backbone.classifier.out_features != n_out:
            backbone.classifier = nn.Sequential(nn.Linear(backbone.classifier.in_features, n_out))  # model.py:190-192
        self.backbone = backbone
        self.encoder_size = encoder_size
        self.curllayer = CURLLayer(num_lab_points, num_rgb_points, num_hsv_points)

    def predict_knots(self, img):
        x = img
        if self.encoder_size is not None and tuple(img.shape[-2:]) != (self.encoder_size, self.encoder_size):
            x = nn.functional.interpolate(img, size=(self.encoder_size, self.encoder_size), mode="bilinear",
                                          align_corners=False, antialias=True)
        return self.backbone(x)  # model.py:196

    def forward(self, img, mask, L=None, R=None, H=None):
        """L, R, H are accepted and ignored, exactly as in the reference (model.py:195-199 overwrites them)."""
        curves = self.predict_knots(img)
        L, R, H = curves[:, :self.curve_break_1], \
            curves[:, self.curve_break_1:self.curve_break_2], \
            curves[:, self.curve_break_2:]
        img, gradient_regulariser = self.curllayer(img, mask, L, R, H)
        return img, gradient_regulariser


# ---------------------------------------------------------------------------------------------------------
# The polynomial model of the fork: model.py:206-535.  Per-pixel work = ONE fused HIP kernel.
# ---------------------------------------------------------------------------------------------------------
def _ncr(n, r):
    import math
    return math.comb(n, r)


def _powers(degree, num_variables):
    import itertools
    rows = []
    for total in range(degree + 1):
        ts = [t for t in itertools.product(range(total + 1), repeat=num_variables) if sum(t) == total]
        ts.sort(reverse=True)
        rows.extend(ts)
    return rows


class _TriSpaceFn(torch.autograd.Function):
    """Autograd node around the fused polynomial kernels: gradient w.r.t. the coefficients only."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, img, coeffs, residual_only):
        ctx.save_for_backward(img, coeffs)
        ctx.residual_only = residual_only
        return ops.trispace_forward(img, coeffs, residual_only=residual_only)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        img, coeffs = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("curl_amd: the polynomial path differentiates w.r.t. the coefficients only "
                                      "(the image is data in main.py's train step)")
        return None, ops.trispace_backward(img, coeffs, grad_out.contiguous(), ctx.residual_only), None


def _no_grad_path(coeffs, who):
    """The stand-alone polynomial layers are forward-only kernels; gradients flow through TriSpaceRegNet's fused path."""
    if torch.is_grad_enabled() and coeffs.requires_grad:
        raise NotImplementedError(f"curl_amd: {who} alone is forward-only; TriSpaceRegNet.forward / generate_residual "
                                  "carry the backward (ops.trispace_backward). Use torch.no_grad() here.")


class ChannelPolyLayer(nn.Module):
    """model.py:206-333.  forward(img [B,V,H,W], coeffs [B,num_out,num_coeffs]) -> [B,num_out,H,W].
    The HIP kernel covers what the fork uses: degree 4, V = 5 or 3, num_out = 3."""

    def __init__(self, degree=3, num_variables=3, num_out=None):
        assert degree >= 0 and type(degree) == int, "`degree` must be non-negative integer"
        assert num_variables >= 0 and type(num_variables) == int, "`num_variables` must be non-negative integer"
        super().__init__()
        self.degree = degree
        self.num_variables = num_variables
        self.num_out = self.num_variables if num_out is None else num_out
        self.num_coeffs = _ncr(num_variables + degree, degree)
        self.powers = nn.Parameter(torch.Tensor(_powers(degree, num_variables)), requires_grad=False)  # state-dict key

    @staticmethod
    def generate_powers(order, n_variables):
        """Same sequence as the reference's generator (model.py:222-246)."""
        yield from _powers(order, n_variables)

    def forward(self, img, coeffs):
        assert img.shape[1] == self.num_variables, "There should be a polynomial variable per channel"
        assert len(coeffs.shape) == 3 and coeffs.shape[2] == self.num_coeffs, \
            f"coeffs must be [B, num_out, {self.num_coeffs}]"
        if self.degree != 4 or self.num_variables not in (3, 5) or self.num_out != 3:
            raise NotImplementedError("the HIP polynomial kernel is built for degree 4, 3 or 5 variables, 3 outputs "
                                      "(the configurations model.py:426,450 use)")
        _no_grad_path(coeffs, "ChannelPolyLayer")
        return ops.poly_layer(img, coeffs)


class Deg4MobilePolyLayer(nn.Module):
    """model.py:336-415: ChannelPolyLayer(degree=4, num_variables=5, num_out=3) written out for CoreML."""

    def __init__(self):
        super().__init__()
        self.num_coeffs = 126
        self.powers = nn.Parameter(torch.Tensor(_powers(4, 5)), requires_grad=False)

    def forward(self, img, coeffs):
        _no_grad_path(coeffs, "Deg4MobilePolyLayer")
        return ops.poly_layer(img, coeffs.reshape(img.shape[0], 3, self.num_coeffs))


class PolyRegNet(nn.Module):
    """model.py:418-436: encoder -> [B,3,35] coefficients -> sigmoid(ChannelPolyLayer(degree 4, 3 variables)(img)) * mask.
    The polynomial layer is the HIP kernel (ops.poly_layer; forward only, like the reference's use of this class in
    inference); the backbone is injectable as in TriSpaceRegNet (the reference downloads timm's efficientnetv2_rw_s)."""

    def __init__(self, num_channels=3, polynomial_order=4, backbone=None, feature_width=1792):
        super().__init__()
        self.num_channels = num_channels
        self.order = polynomial_order
        self.polylayer = ChannelPolyLayer(degree=self.order, num_variables=self.num_channels)
        self.num_coeffs = self.polylayer.num_coeffs
        if backbone is None:
            backbone = CurveEncoder(num_outputs=1, num_features=feature_width)
        backbone.classifier = nn.Linear(in_features=feature_width, out_features=self.num_channels * self.num_coeffs)
        self.backbone = backbone
        self.sigmoid = nn.Sigmoid()

    def forward(self, img, mask):
        coeffs = self.backbone(img).reshape(img.shape[0], self.num_channels, self.num_coeffs)
        if torch.is_grad_enabled() and coeffs.requires_grad:
            raise NotImplementedError("curl_amd: PolyRegNet is forward-only (the trainable polynomial model of this "
                                      "path is TriSpaceRegNet); wrap the call in torch.no_grad()")
        return self.sigmoid(self.polylayer(img, coeffs)) * mask


class TriSpaceRegNet(nn.Module):
    """model.py:439-535: encoder -> [B,3,3,num_coeffs] -> per-pixel degree-4 polynomials in RGB, Lab and HSV.
    generate_residual + generate_image run as one fused kernel (ops.trispace_forward).
    The reference's backbone is timm `efficientnetv2_rw_t` (not installed here, needs a download); any module
    with a `.classifier` whose pooled feature width is `feature_width` can be injected."""

    def __init__(self, polynomial_order=4, spatial=False, max_resolution=10000, is_train=True, use_sync_bn=False,
                 polylayer=None, backbone=None, feature_width=1024):
        super().__init__()
        self.num_channels = 3
        self.num_spaces = 3
        self.num_in = self.num_channels + 2 * spatial
        self.order = polynomial_order
        self.is_train = is_train
        self.max_resolution = max_resolution
        self.polylayer = polylayer if polylayer is not None else ChannelPolyLayer(
            degree=self.order, num_variables=self.num_in, num_out=self.num_channels)
        self.num_coeffs = self.polylayer.num_coeffs
        if self.order != 4 or self.num_coeffs not in (126, 35):
            raise NotImplementedError("fused kernel: polynomial_order 4 with spatial=True (126) or False (35)")
        if backbone is None:
            backbone = CurveEncoder(num_outputs=1, num_features=feature_width)
        if use_sync_bn:
            backbone = nn.SyncBatchNorm.convert_sync_batchnorm(backbone)  # model.py:457-458
        backbone.classifier = nn.Sequential(  # model.py:459-463
            nn.Linear(feature_width, 1024), nn.Linear(1024, 512), nn.Linear(512, 512),
            nn.Linear(512, self.num_spaces * self.num_channels * self.num_coeffs))
        self.backbone = backbone
        self.rgb2lab, self.lab2rgb = colors.RGB2LAB(), colors.LAB2RGB()
        self.rgb2hsv, self.hsv2rgb = colors.RGB2HSV(), colors.HSV2RGB()
        self.sigmoid = nn.Sigmoid()

    def generate_coefficients(self, img, mask):
        """model.py:522-527."""
        coeffs = self.backbone(img * mask).reshape(img.shape[0], self.num_spaces, self.num_channels, self.num_coeffs)
        return coeffs[:, 0], coeffs[:, 1], coeffs[:, 2]

    def generate_residual(self, img, R, L, H):
        """model.py:499-515, one kernel (differentiable w.r.t. R, L, H)."""
        coeffs = torch.stack((R, L, H), 1)
        if torch.is_grad_enabled() and coeffs.requires_grad:
            return _TriSpaceFn.apply(img, coeffs, True)
        return ops.trispace_forward(img, coeffs, residual_only=True)

    @staticmethod
    def generate_image(img, residual):
        """model.py:517-520."""
        return torch.clamp(img + residual, 0.0, 1.0)

    def forward(self, img, mask, target_img=None):
        """model.py:529-535: coefficients from (img*mask); residual on `target_img` (full resolution) if given.
        is_train=True returns the image clamp(input + residual), else the residual."""
        coeffs = self.backbone(img * mask).reshape(img.shape[0], self.num_spaces, self.num_channels, self.num_coeffs)
        input_img = img if target_img is None else target_img
        if torch.is_grad_enabled() and coeffs.requires_grad:
            return _TriSpaceFn.apply(input_img, coeffs, not self.is_train)
        return ops.trispace_forward(input_img, coeffs, residual_only=not self.is_train)


# ---------------------------------------------------------------------------------------------------------
# CURLLoss (model.py:35-118): the four per-pixel terms in one fused HIP pass (+ backward); MS-SSIM is injected.
# ---------------------------------------------------------------------------------------------------------
class _LossTermsFn(torch.autograd.Function):
    """(pred, target, mask) -> (rgb_l1, cosine, lab_l1, hsv_l1, L_pred, L_target) as in model.py:89-109."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, pred, target, mask):
        sums, Lp, Lt = ops.loss_term_sums(pred, target, mask)
        s = sums.sum(0)
        n = float(pred.shape[0] * pred.shape[2] * pred.shape[3])
        unmasked = 3.0 * s[4]
        rgb, lab, hsv = s[0] / unmasked, s[2] / unmasked, s[3] / unmasked
        cosine = 1.0 - s[1] / n - (n - s[4]) / n  # model.py:98: mean over the broadcast [B,B,H,W]
        ctx.save_for_backward(pred, target, unmasked)
        ctx.mask, ctx.n = mask, n
        ctx.mark_non_differentiable(Lt)
        f = torch.float32
        return rgb.to(f), cosine.to(f), lab.to(f), hsv.to(f), Lp, Lt

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_rgb, g_cos, g_lab, g_hsv, g_Lp, _g_Lt):
        pred, target, unmasked = ctx.saved_tensors
        w = torch.stack((g_rgb.double() / unmasked, -g_cos.double() / ctx.n, g_lab.double() / unmasked,
                         g_hsv.double() / unmasked)).to(torch.float32)
        return ops.loss_terms_backward(pred, target, ctx.mask, w, g_Lp), None, None


class CURLLoss(nn.Module):
    """model.py:35-118: the four pointwise terms on the fused HIP kernels, the MS-SSIM term of
    model.py:103-105 through `metric.MSSSIMMetric(num_channel=num_channel)` exactly as the reference builds it
    (model.py:48: default window 11; `ssim_window_size` is stored and, as there, not used)."""

    def __init__(self, ssim_window_size=5, num_channel=1, msssim_layer="reference"):
        super().__init__()
        self.ssim_window_size = ssim_window_size
        self.num_channel = num_channel
        # None switches the term off (kernel-only timing); anything callable replaces it
        self.msssim_layer = metric.MSSSIMMetric(num_channel=num_channel) if msssim_layer == "reference" else msssim_layer
        self.rgb2lab = colors.RGB2LAB()
        self.rgb2hsv = colors.RGB2HSV()

    def forward(self, predicted_img_batch, target_img_batch, mask):
        rgb, cosine, lab, hsv, Lp, Lt = _LossTermsFn.apply(predicted_img_batch, target_img_batch, mask)
        ssim = (1.0 - self.msssim_layer(Lp, Lt)).mean() if self.msssim_layer is not None else 0.0
        return (rgb + cosine + lab + hsv + 10 * ssim) / 5  # model.py:111-116
