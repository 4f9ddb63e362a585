"""Drop-in for the masked PSNR of the reference's metric.py (PSNRMetric, metric.py:28-72).
MS-SSIM (metric.py:75-211) is grouped convolutions -- stock PyTorch-ROCm, not part of this path."""
import torch.nn as nn

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
