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
    def forward(ctx, img, mask, L, R, H, flags=0):
        out, reg, ws = ops.curl_layer_forward(img, mask, L, R, H, flags=flags, return_workspace=True)
        # the knot workspace (exp'd knots, collapsed curves: 768 B per image) rides along: the backward skips its prep launch
        ctx.save_for_backward(img, L.contiguous(), R.contiguous(), H.contiguous(), ws)
        ctx.mask = mask
        ctx.flags = flags
        return out, reg

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out, grad_reg):
        img, L, R, H, ws = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0]
        g_img, gL, gR, gH = ops.curl_layer_backward(img, ctx.mask, L, R, H, grad_out.contiguous(), grad_reg, need_img,
                                                    workspace=ws, flags=ctx.flags)
        return g_img, None, gL, gR, gH, None


class CURLLayer(nn.Module):
    """model.py:121-176.  Same constructor arguments, same forward signature and returns.
    `paper_pwl=True` (not in the reference) evaluates the curves as the paper's clamped piecewise-linear
    interpolation of the knots (CURL_F_PWL: knots in LDS) instead of the reference's affine form; inference only.
    `foreground_masks=True` (not in the reference either) tells the kernel that the bool / uint8 masks it will see have
    sizeable empty regions (data.py:186-190: segmentation masks): wavefronts test their mask bytes before asking for their
    pixels and never read fully masked-out tiles (CURL_F_MASK_FIRST: -7 % at 70 % coverage, -25 % at 40 %, +0.9 % on an
    all-ones mask; the same results bit for bit)."""

    def __init__(self, num_lab_points=48, num_rgb_points=48, num_hsv_points=64, paper_pwl=False, foreground_masks=False):
        super().__init__()
        self.paper_pwl = paper_pwl
        self.flags = ops.F_MASK_FIRST if foreground_masks else 0
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
            return ops.curl_layer_forward(img, mask, L, R, H, flags=ops.F_PWL | self.flags)
        if needs_grad:
            return _CurlLayerFn.apply(img, mask, L, R, H, self.flags)
        return ops.curl_layer_forward(img, mask, L, R, H, flags=self.flags)


# ---------------------------------------------------------------------------------------------------------
# Encoder: EfficientNetV2 (timm's `efficientnetv2_rw_t` / `efficientnetv2_rw_s`, model.py:189,427,456) with timm's
# module names, so that a checkpoint saved by the reference (main.py:332-338) loads key for key.  timm itself is
# not installed here (and the reference asks it for a weight download); the architecture is restated from its
# published definition (timm 0.5.4, efficientnet.py `_gen_efficientnetv2_s`, efficientnet_blocks.py).
# Stock PyTorch-ROCm ops only (MIOpen convolutions): the encoder is not part of the hand-written path.
# ---------------------------------------------------------------------------------------------------------
def _make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def _conv(cin, cout, k, stride=1, groups=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride, ((stride - 1) + (k - 1)) // 2, groups=groups, bias=bias)


class _ConvBnAct(nn.Module):
    """timm ConvBnAct ('cn'): conv, bn1, act1 (+ skip)."""

    def __init__(self, cin, cout, k, stride, skip):
        super().__init__()
        self.has_residual = skip and stride == 1 and cin == cout
        self.conv = _conv(cin, cout, k, stride)
        self.bn1 = nn.BatchNorm2d(cout)
        self.act1 = nn.SiLU(inplace=True)

    def forward(self, x):
        y = self.act1(self.bn1(self.conv(x)))
        return y + x if self.has_residual else y


class _SqueezeExcite(nn.Module):
    """timm SqueezeExcite: conv_reduce, act1, conv_expand, gate."""

    def __init__(self, chs, rd_channels):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, rd_channels, 1, bias=True)
        self.act1 = nn.SiLU(inplace=True)
        self.conv_expand = nn.Conv2d(rd_channels, chs, 1, bias=True)
        self.gate = nn.Sigmoid()

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * self.gate(self.conv_expand(self.act1(self.conv_reduce(s))))


class _EdgeResidual(nn.Module):
    """timm EdgeResidual ('er', FusedMBConv): conv_exp (k x k), bn1, act1, se, conv_pwl (1 x 1), bn2."""

    def __init__(self, cin, cout, k, stride, exp_ratio):
        super().__init__()
        mid = _make_divisible(cin * exp_ratio)
        self.has_residual = stride == 1 and cin == cout
        self.conv_exp = _conv(cin, mid, k, stride)
        self.bn1 = nn.BatchNorm2d(mid)
        self.act1 = nn.SiLU(inplace=True)
        self.se = nn.Identity()
        self.conv_pwl = _conv(mid, cout, 1)
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x):
        y = self.bn2(self.conv_pwl(self.se(self.act1(self.bn1(self.conv_exp(x))))))
        return y + x if self.has_residual else y


class _InvertedResidual(nn.Module):
    """timm InvertedResidual ('ir', MBConv): conv_pw, bn1, act1, conv_dw, bn2, act2, se, conv_pwl, bn3."""

    def __init__(self, cin, cout, k, stride, exp_ratio, se_ratio):
        super().__init__()
        mid = _make_divisible(cin * exp_ratio)
        self.has_residual = stride == 1 and cin == cout
        self.conv_pw = _conv(cin, mid, 1)
        self.bn1 = nn.BatchNorm2d(mid)
        self.act1 = nn.SiLU(inplace=True)
        self.conv_dw = _conv(mid, mid, k, stride, groups=mid)
        self.bn2 = nn.BatchNorm2d(mid)
        self.act2 = nn.SiLU(inplace=True)
        # the builder rescales se_ratio by 1/exp_ratio (reduction counted from the block's input), rounds with round()
        self.se = _SqueezeExcite(mid, round(mid * se_ratio / exp_ratio)) if se_ratio else nn.Identity()
        self.conv_pwl = _conv(mid, cout, 1)
        self.bn3 = nn.BatchNorm2d(cout)

    def forward(self, x):
        y = self.act1(self.bn1(self.conv_pw(x)))
        y = self.se(self.act2(self.bn2(self.conv_dw(y))))
        y = self.bn3(self.conv_pwl(y))
        return y + x if self.has_residual else y


# (block, repeats, kernel, stride, expansion, channels, se_ratio)
_V2_ARCH = {
    # _gen_efficientnetv2_s(rw=False): efficientnetv2_rw_t = channel_multiplier 0.8, depth_multiplier 0.9
    "efficientnetv2_rw_t": dict(stages=[("cn", 2, 3, 1, 1, 24, 0), ("er", 4, 3, 2, 4, 48, 0), ("er", 4, 3, 2, 4, 64, 0),
                                        ("ir", 6, 3, 2, 4, 128, 0.25), ("ir", 9, 3, 1, 6, 160, 0.25),
                                        ("ir", 15, 3, 2, 6, 256, 0.25)],
                                channel_multiplier=0.8, depth_multiplier=0.9, num_features=1280, stem_size=24),
    # _gen_efficientnetv2_s(rw=True): first stage 'er', last stage 272 channels, 1792 features
    "efficientnetv2_rw_s": dict(stages=[("er", 2, 3, 1, 1, 24, 0), ("er", 4, 3, 2, 4, 48, 0), ("er", 4, 3, 2, 4, 64, 0),
                                        ("ir", 6, 3, 2, 4, 128, 0.25), ("ir", 9, 3, 1, 6, 160, 0.25),
                                        ("ir", 15, 3, 2, 6, 272, 0.25)],
                                channel_multiplier=1.0, depth_multiplier=1.0, num_features=1792, stem_size=24),
}


class EfficientNetV2(nn.Module):
    """timm.models.efficientnet.EfficientNet for the two V2 variants the reference builds: same module tree
    (conv_stem, bn1, act1, blocks.<stage>.<index>.*, conv_head, bn2, act2, global_pool, classifier), same shapes.
    `channel_multiplier` / `num_features` can be overridden to get a small encoder for tests."""

    def __init__(self, variant="efficientnetv2_rw_t", num_classes=1000, channel_multiplier=None, depth_multiplier=None,
                 num_features=None):
        super().__init__()
        import math
        cfg = _V2_ARCH[variant]
        cm = cfg["channel_multiplier"] if channel_multiplier is None else channel_multiplier
        dm = cfg["depth_multiplier"] if depth_multiplier is None else depth_multiplier

        def chs(c):
            return _make_divisible(c * cm)

        stem = chs(cfg["stem_size"])
        self.conv_stem = _conv(3, stem, 3, 2)
        self.bn1 = nn.BatchNorm2d(stem)
        self.act1 = nn.SiLU(inplace=True)
        stages, cin = [], stem
        for kind, reps, k, stride, exp, c, se in cfg["stages"]:
            cout, blocks = chs(c), []
            for i in range(int(math.ceil(reps * dm))):
                st = stride if i == 0 else 1
                if kind == "cn":
                    blocks.append(_ConvBnAct(cin, cout, k, st, skip=True))
                elif kind == "er":
                    blocks.append(_EdgeResidual(cin, cout, k, st, exp))
                else:
                    blocks.append(_InvertedResidual(cin, cout, k, st, exp, se))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.num_features = chs(cfg["num_features"]) if num_features is None else num_features
        self.conv_head = _conv(cin, self.num_features, 1)
        self.bn2 = nn.BatchNorm2d(self.num_features)
        self.act2 = nn.SiLU(inplace=True)
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Linear(self.num_features, num_classes)

    def forward_features(self, x):
        x = self.act1(self.bn1(self.conv_stem(x)))
        x = self.blocks(x)
        return self.act2(self.bn2(self.conv_head(x)))

    def forward(self, x):
        return self.classifier(self.global_pool(self.forward_features(x)).flatten(1))


class CurveEncoder(EfficientNetV2):
    """The encoder GCURLNet / PolyRegNet regress from: timm's `efficientnetv2_rw_s` (model.py:189,427) by default, with a
    `num_outputs`-wide linear head.  `width` scales the channels (tests and the smoke run use a narrow one)."""

    def __init__(self, num_outputs=160, width=1.0, num_features=1792, variant="efficientnetv2_rw_s"):
        super().__init__(variant, num_classes=num_outputs, channel_multiplier=_V2_ARCH[variant]["channel_multiplier"] * width,
                         num_features=num_features)


class GCURLNet(nn.Module):
    """model.py:179-203: encoder -> [B,160] raw knots -> split at 48 / 96 -> CURLLayer.

    The reference's constructor is broken (undefined self.num_spaces..., model.py:191) and needs a
    network download (model.py:189); here the head width is what the layer consumes
    (num_lab_points + num_rgb_points + num_hsv_points) and the backbone is injectable.
    `encoder_size` (optional): the encoder sees the image resized to this square while the curves are
    applied at full resolution -- the low-res-encode / full-res-apply shape of infer.py:32-44.
    `foreground_masks`: see CURLLayer."""

    def __init__(self, num_lab_points=48, num_rgb_points=48, num_hsv_points=64, backbone=None, encoder_size=None,
                 foreground_masks=False):
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
        self.curllayer = CURLLayer(num_lab_points, num_rgb_points, num_hsv_points, foreground_masks=foreground_masks)

    def predict_knots(self, img):
        x = img
        if self.encoder_size is not None and tuple(img.shape[-2:]) != (self.encoder_size, self.encoder_size):
            x = nn.functional.interpolate(img, size=(self.encoder_size, self.encoder_size), mode="bilinear",
                                          align_corners=False, antialias=True)
        return self.backbone(x)  # model.py:196

    def forward(self, img, mask, L=None, R=None, H=None, target=None, criterion=None):
        """L, R, H are accepted and ignored, exactly as in the reference (model.py:195-199 overwrites them).
        target / criterion (not in the reference): given a target image and a CURLLoss, the training step's two calls
        (main.py:283-285) run with the layer and the loss' pointwise terms as ONE forward kernel (_LayerLossFn) and the
        result is (img, gradient_regulariser, loss) -- the values and gradients of `criterion(self(img, mask)[0], target, mask)`."""
        curves = self.predict_knots(img)
        L, R, H = curves[:, :self.curve_break_1], \
            curves[:, self.curve_break_1:self.curve_break_2], \
            curves[:, self.curve_break_2:]
        if target is not None:
            lay = self.curllayer  # (its foreground_masks / paper_pwl options are the stand-alone layer's: not taken here)
            L, R, H = L[:, :lay.num_lab_points], R[:, :lay.num_rgb_points], H[:, :lay.num_hsv_points]  # model.py:153,159,165
            out, reg, rgb, cosine, lab, hsv, Lp, Lt = _LayerLossFn.apply(img, mask, L, R, H, target)
            ssim = (1.0 - criterion.msssim_layer(Lp, Lt)).mean() if criterion.msssim_layer is not None else 0.0
            return out, reg, (rgb + cosine + lab + hsv + 10 * ssim) / 5  # model.py:111-116
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
        if backbone is None:  # model.py:456: timm.create_model('efficientnetv2_rw_t')
            backbone = EfficientNetV2("efficientnetv2_rw_t") if feature_width == 1024 else \
                CurveEncoder(num_outputs=1, num_features=feature_width, variant="efficientnetv2_rw_t")
        if use_sync_bn:
            backbone = nn.SyncBatchNorm.convert_sync_batchnorm(backbone)  # model.py:457-458
        backbone.classifier = nn.Sequential(  # model.py:459-463
            nn.Linear(feature_width, 1024), nn.Linear(1024, 512), nn.Linear(512, 512),
            nn.Linear(512, self.num_spaces * self.num_channels * self.num_coeffs))
        self.backbone = backbone
        self.rgb2lab, self.lab2rgb = colors.RGB2LAB(), colors.LAB2RGB()
        self.rgb2hsv, self.hsv2rgb = colors.RGB2HSV(), colors.HSV2RGB()
        self.sigmoid = nn.Sigmoid()
        # model.py:476-484: the coordinate ramps are frozen nn.Parameters, hence state-dict keys `x` and `y` of every
        # reference checkpoint (int64 arange for spatial=True, zero-width float for spatial=False).  The fused kernel
        # forms column/width and row/height itself; the parameters exist so that checkpoints load key for key.
        if not spatial:
            x, y = torch.zeros(1, 0, 1, self.max_resolution), torch.zeros(1, 0, self.max_resolution, 1)
        else:
            x = torch.arange(0, self.max_resolution).reshape(1, 1, 1, self.max_resolution)
            y = torch.arange(0, self.max_resolution).reshape(1, 1, self.max_resolution, 1)
        self.x = nn.Parameter(x, requires_grad=False)
        self.y = nn.Parameter(y, requires_grad=False)

    def _check_resolution(self, img):
        assert img.shape[2] <= self.max_resolution and img.shape[3] <= self.max_resolution, \
            "img width and height must be less than `max_resolution`, set for instance to: {}".format(self.max_resolution)

    def generate_coefficients(self, img, mask):
        """model.py:522-527."""
        coeffs = self.backbone(img * mask).reshape(img.shape[0], self.num_spaces, self.num_channels, self.num_coeffs)
        return coeffs[:, 0], coeffs[:, 1], coeffs[:, 2]

    def generate_residual(self, img, R, L, H):
        """model.py:499-515, one kernel (differentiable w.r.t. R, L, H)."""
        self._check_resolution(img)  # model.py:491
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
        self._check_resolution(input_img)  # model.py:491
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
        # A mask of ONE image ([1,1,H,W] or [1,H,W]: ops._mask) is expanded over the batch for the
        # kernel: its sum s[4] and its zeros are then counted `rep` times.  model.py:90 takes mask.sum() of the mask AS GIVEN
        # (so the three L1 terms are rep times larger than with a [B,1,H,W] copy of it), while model.py:98's mean runs over
        # the broadcast tensor (zeros counted per image).
        rep = 1 if mask is None else pred.shape[0] // mask.shape[0]
        unmasked = 3.0 * s[4] / rep
        rgb, lab, hsv = s[0] / unmasked, s[2] / unmasked, s[3] / unmasked
        # model.py:98 adds torch.logical_not(mask) -- 1 where the mask is EXACTLY 0 -- and takes the mean over the broadcast
        # [B,B,H,W].  bool / uint8 masks (data.py:190): their zeros are n - sum; a float mask with values strictly inside
        # (0, 1) has none of those counted, so its zeros are counted as such (on the mask as given, times `rep`).
        n_zero = (mask == 0).sum().double() * rep if (mask is not None and mask.is_floating_point()) else n - s[4]
        cosine = 1.0 - s[1] / n - n_zero / n
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


class _LayerLossFn(torch.autograd.Function):
    """(img, mask, L, R, H, target) -> (out, reg, rgb_l1, cosine, lab_l1, hsv_l1, L_pred, L_target): CURLLayer.forward and
    CURLLoss' pointwise terms in ONE forward pass (ops.layer_loss_forward); the backward is the two existing kernels in
    sequence -- the loss terms' pullback to the prediction, added to whatever gradient `out` received itself, then the
    layer's backward with the forward's knot workspace."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, img, mask, L, R, H, target):
        out, reg, sums, Lp, Lt, ws = ops.layer_loss_forward(img, mask, L, R, H, target)
        s = sums.sum(0)
        n = float(out.shape[0] * out.shape[2] * out.shape[3])
        rep = 1 if mask is None else out.shape[0] // mask.shape[0]  # a one-image mask broadcast over the batch: _LossTermsFn
        unmasked = 3.0 * s[4] / rep
        rgb, lab, hsv = s[0] / unmasked, s[2] / unmasked, s[3] / unmasked
        n_zero = (mask == 0).sum().double() * rep if (mask is not None and mask.is_floating_point()) else n - s[4]
        cosine = 1.0 - s[1] / n - n_zero / n  # model.py:98 (see _LossTermsFn)
        ctx.save_for_backward(img, L.contiguous(), R.contiguous(), H.contiguous(), ws, out, target, unmasked)
        ctx.mask, ctx.n = mask, n
        ctx.mark_non_differentiable(Lt)
        ctx.set_materialize_grads(False)  # an output nobody used arrives as None, not as a zero image to be added
        f = torch.float32
        return out, reg, rgb.to(f), cosine.to(f), lab.to(f), hsv.to(f), Lp, Lt

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_out, g_reg, g_rgb, g_cos, g_lab, g_hsv, g_Lp, _g_Lt):
        img, L, R, H, ws, out, target, unmasked = ctx.saved_tensors
        zero = torch.zeros((), dtype=torch.float64, device=out.device)
        d = lambda g: zero if g is None else g.double()  # noqa: E731
        w = torch.stack((d(g_rgb) / unmasked, -d(g_cos) / ctx.n, d(g_lab) / unmasked, d(g_hsv) / unmasked)).to(torch.float32)
        g_pred = ops.loss_terms_backward(out, target, ctx.mask, w, g_Lp)
        if g_out is not None:
            g_pred = g_pred + g_out
        g_img, gL, gR, gH = ops.curl_layer_backward(img, ctx.mask, L, R, H, g_pred, g_reg, ctx.needs_input_grad[0], workspace=ws)
        return g_img, None, gL, gR, gH, None


class CURLLayerWithLoss(nn.Module):
    """Not in the reference: its training step's two calls -- `net_output_img = net(...)` and `criterion(net_output_img, gt,
    mask)` (main.py:283-285) -- as one module, so that the layer and the loss' pointwise terms run as ONE forward kernel.
    forward(img, mask, L, R, H, target) -> (out, reg, loss) with `loss` = CURLLoss()(out, target, mask) and `out`, `reg` =
    CURLLayer()(img, mask, L, R, H): the same values (tests/test_gpu_backward.py), the same gradients.  The MS-SSIM term is
    the reference's layer on the two L planes, as in CURLLoss."""

    def __init__(self, num_lab_points=48, num_rgb_points=48, num_hsv_points=64, ssim_window_size=5, num_channel=1,
                 msssim_layer="reference"):
        super().__init__()
        self.num_lab_points, self.num_rgb_points, self.num_hsv_points = num_lab_points, num_rgb_points, num_hsv_points
        self.msssim_layer = metric.MSSSIMMetric(num_channel=num_channel) if msssim_layer == "reference" else msssim_layer

    def forward(self, img, mask, L, R, H, target):
        L, R, H = L[:, :self.num_lab_points], R[:, :self.num_rgb_points], H[:, :self.num_hsv_points]  # model.py:153,159,165
        out, reg, rgb, cosine, lab, hsv, Lp, Lt = _LayerLossFn.apply(img, mask, L, R, H, target)
        ssim = (1.0 - self.msssim_layer(Lp, Lt)).mean() if self.msssim_layer is not None else 0.0
        return out, reg, (rgb + cosine + lab + hsv + 10 * ssim) / 5  # model.py:111-116


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
