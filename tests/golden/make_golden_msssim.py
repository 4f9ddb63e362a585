"""Golden vectors for MS-SSIM (metric.py:75-211), the remaining term of CURLLoss (model.py:103-105).

`metric` imports in the build container, but MSSSIMMetric builds its Gaussian window with `.cuda()`
(metric.py:116) and compute_ssim calls `.cuda()` on intermediates (metric.py:155-162): with no GPU here the
class cannot run as written.  For the duration of this script `torch.Tensor.cuda` is replaced by the identity, so
the REFERENCE's own class, unedited, runs on the CPU; its outputs and autograd gradients are stored.
Nothing from oracle/ or curl_amd/ is used.

    python tests/golden/make_golden_msssim.py      (build container only)
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("CURL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self  # the only change to the environment; the class is untouched
import metric  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def main():
    g = torch.Generator().manual_seed(77)
    store = {}
    # (tag, window, channels, shape): the loss's configuration (model.py:48: window 11, 1 channel = the L plane),
    # the class default (3 channels), a small window, and a non-square frame that is transposed by metric.py:178
    cases = [("loss", 11, 1, (2, 1, 96, 128)), ("rgb", 11, 3, (2, 3, 80, 96)), ("w5", 5, 1, (1, 1, 64, 64))]
    for tag, ws, ch, shape in cases:
        m = metric.MSSSIMMetric(window_size=ws, num_channel=ch)
        a = torch.rand(*shape, generator=g)
        b = (a + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
        a.requires_grad_(True)
        out = m(a, b)
        w = torch.rand(shape[0], generator=g)
        (out * w).sum().backward()
        ssim, cs = m.compute_ssim(a.detach(), b)
        store.update({f"{tag}_a": a.detach().numpy(), f"{tag}_b": b.numpy(), f"{tag}_w": w.numpy(),
                      f"{tag}_out": out.detach().numpy(), f"{tag}_grad_a": a.grad.numpy(),
                      f"{tag}_ssim": ssim.numpy(), f"{tag}_cs": cs.numpy(),
                      f"{tag}_window": m.gaussian_window.numpy(), f"{tag}_cfg": np.array([ws, ch])})
    # identical images: MS-SSIM = 1
    m = metric.MSSSIMMetric(window_size=11, num_channel=1)
    x = torch.rand(1, 1, 96, 96, generator=g)
    store["same_out"] = m(x, x).numpy()
    store["same_x"] = x.numpy()
    np.savez_compressed(os.path.join(OUT, "msssim.npz"), **store)
    print({k: v.shape for k, v in store.items()})


if __name__ == "__main__":
    main()
