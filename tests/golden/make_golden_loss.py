"""Golden vectors for the per-pixel terms of CURLLoss (model.py:78-116).

model.py cannot be imported (timm/torchvision) and CURLLoss.__init__ builds MSSSIMMetric, whose constructor calls
.cuda() (metric.py:116): the class cannot be instantiated in the build container.  The four pointwise terms are
therefore computed with the REFERENCE's colors.py (imported) and torch, in the statement order of
model.py:89-109, with gradients from autograd through them.  Nothing from oracle/ or curl_amd/ is used.

    python tests/golden/make_golden_loss.py      (build container only)
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("CURL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import colors  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)
rgb2lab, rgb2hsv = colors.RGB2LAB(), colors.RGB2HSV()


def cone(x):  # model.py:62-76
    hsv = torch.clamp(rgb2hsv(x), 0.0, 1.0)
    hue = 2 * math.pi * hsv[:, 0]
    val, sat = hsv[:, 2], hsv[:, 1]
    return torch.stack((val * sat * torch.cos(hue), val * sat * torch.sin(hue), val), 1)


def ref_terms(pr, tg, m):  # model.py:89-109
    unm = pr.shape[1] * m.sum()
    p, t = pr * m, tg * m
    rgb = F.l1_loss(p, t, reduction='sum') / unm
    base = F.cosine_similarity(p, t, dim=1)
    cosv = (1.0 - (base + torch.logical_not(m)).mean(dim=(1, 2))).mean()
    lt = torch.clamp(rgb2lab(t), 0.0, 1.0)
    lp = torch.clamp(rgb2lab(p), 0.0, 1.0)
    lab = F.l1_loss(lp, lt, reduction='sum') / unm
    hsv = F.l1_loss(cone(p), cone(t), reduction='sum') / unm
    return rgb, cosv, lab, hsv, lp[:, 0:1], lt[:, 0:1]


def main():
    g = torch.Generator().manual_seed(404)
    B, H, W = 2, 24, 36
    pred = torch.rand(B, 3, H, W, generator=g) * 1.1 - 0.05
    target = torch.rand(B, 3, H, W, generator=g)
    target[:, :, :4] = pred[:, :, :4].clamp(0, 1)  # zero-difference pixels (sign(0) = 0)
    pred[:, :, 4:6] = 0.0
    target[:, :, 5:6] = 0.0                        # black pixels: the eps path of cosine_similarity
    mask = torch.rand(B, 1, H, W, generator=g) > 0.25
    wl = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(5))
    weights = [1.3, 0.7, 2.0, 0.5]
    store = dict(pred=pred.numpy(), target=target.numpy(), mask=mask.numpy(), wl=wl.numpy(),
                 weights=np.array(weights, np.float32))
    for mk, m in (("bool", mask), ("f32", mask.float())):
        pr = pred.clone().requires_grad_(True)
        rgb, cosv, lab, hsv, Lp, Lt = ref_terms(pr, target, m)
        total = weights[0] * rgb + weights[1] * cosv + weights[2] * lab + weights[3] * hsv + (Lp * wl).sum() * 1e-3
        total.backward()
        for name, v in (("rgb", rgb), ("cos", cosv), ("lab", lab), ("hsv", hsv), ("Lp", Lp), ("Lt", Lt)):
            store[f"{mk}_{name}"] = v.detach().numpy()
        store[f"{mk}_grad_pred"] = pr.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **store)
    print("loss.npz", os.path.getsize(os.path.join(OUT, "loss.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
