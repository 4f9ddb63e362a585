"""Drop-in for the reference's colors.py: four nn.Modules with the same names and state-dict keys.

The constant matrices are kept as (frozen) parameters only so that checkpoints written by the
reference load unchanged (state keys `rgb2lab.rgb_to_xyz`, ..., colors.py:14-25,75-86,184;
convert_state.py:11); the kernels bake the same float32 constants (csrc/curl_math.h)."""
import torch
import torch.nn as nn

from . import ops


def _frozen(t):
    return nn.Parameter(t, requires_grad=False)


_WHITE = [0.950456, 1.0, 1.088754]


class RGB2LAB(nn.Module):
    """colors.py:4-62.  sRGB in [0,1] -> Lab normalised to [0,1] (L/100, (a/110+1)/2, (b/110+1)/2)."""

    def __init__(self):
        super().__init__()
        m = torch.tensor([[0.412453, 0.212671, 0.019334],
                          [0.357580, 0.715160, 0.119193],
                          [0.180423, 0.072169, 0.950227]], dtype=torch.float)
        f = torch.tensor([[0.0, 500.0, 0.0], [116.0, -500.0, 200.0], [0.0, 0.0, -200.0]], dtype=torch.float)
        self.rgb_to_xyz = _frozen(m.t()[None, None].contiguous())
        self.fxfyfz_to_lab = _frozen(f.t()[None, None].contiguous())
        self.xyz_to_rgb_mult = _frozen(torch.tensor(_WHITE, dtype=torch.float).reshape(1, 3, 1, 1))
        self.lab_to_fxfyfz_offset = _frozen(torch.tensor([16.0, 0.0, 0.0], dtype=torch.float).reshape(1, 3, 1, 1))

    def forward(self, img):
        return ops.rgb2lab(img)


class LAB2RGB(nn.Module):
    """colors.py:65-123.  Output is not clamped."""

    def __init__(self):
        super().__init__()
        m = torch.tensor([[3.2404542, -0.9692660, 0.0556434],
                          [-1.5371385, 1.8760108, -0.2040259],
                          [-0.4985314, 0.0415560, 1.0572252]], dtype=torch.float)
        f = torch.tensor([[1 / 116.0, 1 / 116.0, 1 / 116.0], [1 / 500.0, 0, 0], [0, 0, -1 / 200.0]],
                         dtype=torch.float)
        self.xyz_to_rgb = _frozen(m.t()[None, None].contiguous())
        self.lab_to_fxfyfz = _frozen(f.t()[None, None].contiguous())
        self.xyz_to_rgb_mult = _frozen(torch.tensor(_WHITE, dtype=torch.float).reshape(1, 3, 1, 1))
        self.lab_to_fxfyfz_offset = _frozen(torch.tensor([16.0, 0.0, 0.0], dtype=torch.float).reshape(1, 3, 1, 1))

    def forward(self, img):
        return ops.lab2rgb(img)


class HSV2RGB(nn.Module):
    """colors.py:126-177."""

    def forward(self, img):
        return ops.hsv2rgb(img)


class RGB2HSV(nn.Module):
    """colors.py:180-242.  Hue terms add on ties; values are clamped to [1e-9, 1]."""

    def __init__(self):
        super().__init__()
        self.comparison_zero = _frozen(torch.tensor(0.0, dtype=torch.float))

    def forward(self, img):
        return ops.rgb2hsv(img)
