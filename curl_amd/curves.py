"""Drop-in for the reference's curves.py: same function names, arguments and returns.

The pixel work is one HIP kernel launch per call (the reference materialises a
[B,K-2,H,W] temporary per curve, curves.py:31-32)."""
from . import ops
from ._lib import F_EXACT_ORDER, F_PWL  # noqa: F401  (re-exported option bits)


def apply_curve(img, C, slope_sqr_diff, channel_in, channel_out, exact_order=True):
    """curves.py:4-38.  Applies the curve with knots C [B,K] (already exp'd): the scale computed
    from channel_in multiplies channel_out, then the whole image is clamped to [0,1].
    `slope_sqr_diff` [B] is accumulated in place and returned, as in the reference.

    exact_order=True (default) evaluates the sum of curves.py:31-32 term by term in torch's
    order, which reproduces the reference bit for bit; False uses the collapsed a + b*x form."""
    return ops.apply_curve(img, C, slope_sqr_diff, channel_in, channel_out,
                           flags=F_EXACT_ORDER if exact_order else 0)


def adjust_rgb(img, R, exact_order=False):
    """curves.py:90-133.  R [B,3*K] raw parameters (exp applied inside).  Returns (img, reg[B]).
    The reference's wrapper seeds the regulariser with None and raises (curves.py:111); the
    semantics here are the evident intent: seed zeros(B)."""
    return ops.adjust_rgb(img, R, flags=F_EXACT_ORDER if exact_order else 0)


def adjust_lab(img, L, exact_order=False):
    """curves.py:136-180."""
    return ops.adjust_lab(img, L, flags=F_EXACT_ORDER if exact_order else 0)


def adjust_hsv(img, S, exact_order=False):
    """curves.py:41-87: H->H, H->S (on the adjusted hue), S->S, V->V."""
    return ops.adjust_hsv(img, S, flags=F_EXACT_ORDER if exact_order else 0)
