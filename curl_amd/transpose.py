"""Drop-in for the reference's transpose.py (numpy axis swaps, views) plus the device-side
fused forms of the file edge: u8 HWC <-> f32 CHW in one kernel each."""
import numpy as np

from .ops import f32chw_to_u8hwc, u8hwc_to_f32chw  # noqa: F401


def swapimdims_3HW_HW3(img):
    """transpose.py:4-16: channels-first -> channels-last, 3-D or 4-D numpy array (a view)."""
    if img.ndim == 3:
        return np.transpose(img, (1, 2, 0))
    elif img.ndim == 4:
        return np.transpose(img, (0, 2, 3, 1))


def swapimdims_HW3_3HW(img):
    """transpose.py:19-31: channels-last -> channels-first."""
    if img.ndim == 3:
        return np.transpose(img, (2, 0, 1))
    elif img.ndim == 4:
        return np.transpose(img, (0, 3, 1, 2))
