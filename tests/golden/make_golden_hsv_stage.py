"""Golden vectors for the fused HSV stage (model.py:163-169) by RUNNING THE REFERENCE's primitives.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden_hsv_stage.py

    rgb -> colors.RGB2HSV (colors.py:195-242) -> adjust_hsv order over curves.apply_curve (curves.py:41-87; the wrapper
    itself raises as written, SURVEY.md 0.2: stage order restated in make_golden.ref_adjust, regulariser seeded zeros(B))
    -> * mask (model.py:166) -> colors.HSV2RGB (colors.py:131-177)

Every array is a seeded input or the reference's output on it.  Nothing from oracle/ or curl_amd/ is used.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402  (imports the reference's colors / curves; defines ref_adjust, make_masks)

OUT = os.path.dirname(os.path.abspath(__file__))


def ref_hsv_stage(rgb, mask, H):
    hsv = MG.colors.RGB2HSV()(rgb)                       # model.py:163
    hsv, reg = MG.ref_adjust(hsv, H[:, :64], MG.HSV_PAIRS)  # model.py:165
    hsv = hsv * mask                                     # model.py:166
    return MG.colors.HSV2RGB()(hsv), reg                 # model.py:169


def main():
    g = torch.Generator().manual_seed(4242)
    B, H, W = 2, 24, 40
    store = {}
    inputs = {
        "img": torch.rand(B, 3, H, W, generator=g),
        "img8": torch.randint(0, 256, (B, 3, H, W), generator=g).float() / 255,   # 8-bit grid: channel ties
        "wide": torch.rand(B, 3, H, W, generator=g) * 2 - 0.5,                     # out of range: RGB2HSV clamps first
    }
    # tie / degenerate pixels in the first row of "img8": grey, black, white, primaries, two-channel ties
    ties = torch.tensor([[.5, .5, .5], [0, 0, 0], [1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [.7, .7, .2], [.2, .7, .7],
                         [.7, .2, .7], [.3, .3, .30000001]])
    inputs["img8"][:, :, 0, :ties.shape[0]] = ties.t()[None]
    for k, v in inputs.items():
        store[k] = MG.npy(v)
    masks = MG.make_masks(B, H, W, g)
    for k, m in masks.items():
        store[f"mask_{k}"] = MG.npy(m)
    for sig_name, sigma in (("s01", 0.1), ("s05", 0.5)):
        Hh = torch.randn(B, 64, generator=g) * sigma
        store[f"{sig_name}_H"] = MG.npy(Hh)
        for in_name, x in inputs.items():
            for mk, m in masks.items():
                mf = m.float() if m.dtype == torch.bool else m
                out, reg = ref_hsv_stage(x, mf, Hh)
                store[f"{sig_name}_{in_name}_{mk}_out"] = MG.npy(out)
                store[f"{sig_name}_{in_name}_{mk}_reg"] = MG.npy(reg)
    np.savez_compressed(os.path.join(OUT, "hsv_stage.npz"), **store)
    print("hsv_stage: keys", len(store), os.path.getsize(os.path.join(OUT, "hsv_stage.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
