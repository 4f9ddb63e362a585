"""curl_amd -- MI355X-native implementation of the CURL per-pixel colour-curve hot path.

Drop-in mirror of the reference's interface for this path (same names, argument
meaning and error behaviour):

    curl_amd.curves     apply_curve, adjust_rgb, adjust_lab, adjust_hsv     (curves.py)
    curl_amd.colors     RGB2LAB, LAB2RGB, RGB2HSV, HSV2RGB                  (colors.py)
    curl_amd.transpose  swapimdims_3HW_HW3, swapimdims_HW3_3HW              (transpose.py)
    curl_amd.model      CURLLayer, GCURLNet                                 (model.py:121-203)
    curl_amd.infer      CLI with the flags of infer.py:14-17

All pixel arithmetic runs in hand-written HIP kernels (curl_amd/csrc) behind the C ABI
of include/curl_hip.h; importing an op without the built library raises ImportError.
"""
__version__ = "0.1.0"
