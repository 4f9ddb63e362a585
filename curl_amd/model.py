"""Drop-in for the curve classes of the reference's model.py: CURLLayer (model.py:121-176) and
GCURLNet (model.py:179-203).

CURLLayer.forward is ONE fused HIP kernel over the pixels (plus a per-image knot-prep kernel);
the encoder of GCURLNet is stock PyTorch-ROCm convolutions, as BASELINE.json's north_star asks.
"""
import torch
import torch.nn as nn

from . import colors, ops


class _CurlLayerFn(torch.autograd.Function):
    """Autograd node around the fused forward/backward kernels."""

    @staticmethod
    def forward(ctx, img, mask, L, R, H):
        out, reg = ops.curl_layer_forward(img, mask, L, R, H)
        ctx.save_for_backward(img, L.contiguous(), R.contiguous(), H.contiguous())
        ctx.mask = mask
        return out, reg

    @staticmethod
    def backward(ctx, grad_out, grad_reg):
        img, L, R, H = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0]
        g_img, gL, gR, gH = ops.curl_layer_backward(img, ctx.mask, L, R, H, grad_out.contiguous(), grad_reg, need_img)
        return g_img, None, gL, gR, gH


class CURLLayer(nn.Module):
    """model.py:121-176.  Same constructor arguments, same forward signature and returns."""

    def __init__(self, num_lab_points=48, num_rgb_points=48, num_hsv_points=64):
        super().__init__()
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
        if torch.is_grad_enabled() and any(t.requires_grad for t in (img, L, R, H)):
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
