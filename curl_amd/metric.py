"""Drop-ins for the reference's metric.py: the masked PSNR (PSNRMetric, metric.py:28-72) on the HIP reduction
kernels, and MS-SSIM (MSSSIMMetric, metric.py:75-211): on a HIP device its five-level SSIM statistics run on the
LDS-tiled stencil kernels of kernels/msssim.inc (forward and backward w.r.t. the first image); the stock-torch
grouped-conv2d form of the reference remains for CPU tensors, windows larger than 11 and a second image that needs a
gradient.  Device-agnostic (the reference's `.cuda()` calls are gone)."""
from math import exp

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class PSNRMetric(nn.Module):
    def __init__(self, max_intensity=1.0):
        super().__init__()
        self.max_intensity = max_intensity

    @staticmethod
    def compute_psnr(image_batchA, image_batchB, mask_batch, max_intensity=1.0):
        """metric.py:50-68: per-image masked PSNR on the device, nan-mean over the batch; None if all NaN."""
        psnr_mean = ops.psnr_per_image(image_batchA, image_batchB, mask_batch, max_intensity).nanmean()
        return psnr_mean if not psnr_mean.isnan() else None

    def forward(self, image_batchA, image_batchB, mask_batch):
        return PSNRMetric.compute_psnr(image_batchA, image_batchB, mask_batch, max_intensity=self.max_intensity)


class _MSSSIMStatsFn(torch.autograd.Function):
    """The five-level SSIM statistics on the HIP kernels; differentiable w.r.t. the first image."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, img1, img2, window_size):
        ctx.save_for_backward(img1, img2)
        ctx.window_size = window_size
        return ops.msssim_stats(img1, img2, window_size)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_ssims, g_mcs):
        img1, img2 = ctx.saved_tensors
        return ops.msssim_stats_backward(img1, img2, g_ssims, g_mcs, ctx.window_size), None, None


class MSSSIMMetric(nn.Module):
    """metric.py:75-211.  Same constructor, `compute_ssim`, `compute_msssim`, `forward` and state (the
    `msssim_weights` parameter); the window is a buffer that follows `.to(device)`.
    On a HIP device `compute_msssim` gets the per-level statistics from the fused kernels (ops.msssim_stats: one
    launch per level, separable window through LDS; the whole CURLLoss forward + backward went from 4.8 to 0.83 ms at
    32x256x256) when the
    window fits (<= 11), the images are large enough for five levels and only the first image needs a gradient;
    otherwise -- and always on the CPU -- the stock-torch form below, with the 2-D Gaussian window applied as two
    1-D passes (it is an outer product, metric.py:99-101)."""

    def __init__(self, window_size=11, num_channel=3):
        super().__init__()
        self.msssim_weights = nn.Parameter(torch.FloatTensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333]),
                                           requires_grad=False)
        self.levels = self.msssim_weights.size()[0]
        self.window_size = window_size
        self.num_channel = num_channel
        g = MSSSIMMetric.gaussian(window_size, 1.5)
        self.register_buffer("gaussian_1d", g, persistent=False)
        self.register_buffer("gaussian_window", MSSSIMMetric.create_window(window_size, num_channel), persistent=False)

    @staticmethod
    def gaussian(window_size, sigma):
        gauss = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
        return gauss / gauss.sum()

    @staticmethod
    def create_window(window_size, num_channel):
        g = MSSSIMMetric.gaussian(window_size, 1.5).unsqueeze(1)
        return g.mm(g.t()).float()[None, None].expand(num_channel, 1, window_size, window_size).contiguous()

    def _blur(self, x):
        C, ws = self.num_channel, self.window_size
        g = self.gaussian_1d.to(x.dtype)
        x = F.conv2d(x, g.view(1, 1, ws, 1).expand(C, 1, ws, 1), padding=(ws // 2, 0), groups=C)
        return F.conv2d(x, g.view(1, 1, 1, ws).expand(C, 1, 1, ws), padding=(0, ws // 2), groups=C)

    def compute_ssim(self, img1, img2):
        """metric.py:120-166 -> (mean SSIM [B], contrast-structure term [B])."""
        mu1, mu2 = self._blur(img1), self._blur(img2)
        mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
        sigma1_sq = self._blur(img1 * img1) - mu1_sq
        sigma2_sq = self._blur(img2 * img2) - mu2_sq
        sigma12 = self._blur(img1 * img2) - mu1_mu2
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        v1 = 2.0 * sigma12 + C2
        v2 = sigma1_sq + sigma2_sq + C2
        ssim_map = ((2 * mu1_mu2 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2)
        return ssim_map.mean(dim=(1, 2, 3)), torch.mean(v1 / v2, dim=(1, 2, 3))

    def compute_msssim(self, img1, img2):
        """metric.py:168-208."""
        if img1.shape[2] != img2.shape[2]:
            img1 = img1.transpose(2, 3)
        if img1.shape != img2.shape:
            raise RuntimeError('Input images must have the same shape (%s vs. %s).', img1.shape, img2.shape)
        if img1.ndim != 4:
            raise RuntimeError('Input images must have four dimensions, not %d', img1.ndim)
        fused = (img1.is_cuda and img1.dtype == torch.float32 and img2.dtype == torch.float32 and self.levels == 5
                 and self.window_size <= 11 and self.window_size % 2 == 1 and min(img1.shape[2:]) >= 32
                 and not (torch.is_grad_enabled() and img2.requires_grad))
        if fused:
            ssims, mcs = _MSSSIMStatsFn.apply(img1, img2, self.window_size)
        else:
            ssims, mcs = [], []
            for _ in range(self.levels):
                ssim, cs = self.compute_ssim(img1, img2)
                ssims.append(ssim)
                mcs.append(cs)
                img1, img2 = F.avg_pool2d(img1, (2, 2)), F.avg_pool2d(img2, (2, 2))
            ssims, mcs = torch.stack(ssims, dim=1), torch.stack(mcs, dim=1)
        ssims = (ssims + 1) / 2
        mcs = (mcs + 1) / 2
        w = self.msssim_weights.reshape(1, -1)
        pow1, pow2 = mcs ** w, ssims ** w
        return torch.prod(pow1[:, :-1] * pow2[:, -1].reshape(-1, 1), dim=1)

    def forward(self, img1, img2):
        return self.compute_msssim(img1, img2)
