"""Checkpoint key compatibility with the reference (convert_state.py:4-16): checkpoints saved from a
DataParallel / DistributedDataParallel wrapper carry a `module.` prefix, and older ones store the four colour
matrices as plain 3x3 tensors instead of the (1,1,3,3) transposed layout of colors.py:14,22,75,83."""
from collections import OrderedDict

_MATRIX_KEYS = ("rgb2lab.rgb_to_xyz", "rgb2lab.fxfyfz_to_lab", "lab2rgb.xyz_to_rgb", "lab2rgb.lab_to_fxfyfz")


def convert_state_dict(model_state_dict):
    out = OrderedDict()
    for key, value in model_state_dict.items():
        name = key[len("module."):] if key.startswith("module.") else key
        if value.dim() == 2 and name.endswith(_MATRIX_KEYS):
            value = value.t()[None, None]
        out[name] = value
    return out
