"""Drop-in for the reference's evaluate.py: `Evaluator(criterion, data_loader, split_name, log_dirpath, local_rank)`
with `evaluate(net, epoch=0, save_images=False) -> (loss, psnr, msssim)` and `save_images(batch, names, epoch)`.

What changes underneath: the masked PSNR and the MS-SSIM statistics are the HIP kernels (curl_amd.metric), and the
image dump of evaluate.py:57-66 -- `.cpu().numpy()`, `(x * 255).astype('uint8')`, `swapimdims_3HW_HW3`, imsave -- does
the quantisation and the CHW->HWC swap on the device (ops.f32chw_to_u8hwc, truncating like astype) and downloads
3 bytes per pixel instead of 12; files are written with PIL (matplotlib is not needed for that).
"""
import os

import torch

from . import metric, ops


class Evaluator:
    def __init__(self, criterion, data_loader, split_name, log_dirpath, local_rank=0):
        self.criterion = criterion
        self.data_loader = data_loader
        self.split_name = split_name
        self.log_dirpath = log_dirpath
        self.psnr = metric.PSNRMetric()
        self.msssim = metric.MSSSIMMetric()  # evaluate.py:44: defaults (window 11, 3 channels)
        self.local_rank = local_rank
        self.is_distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world_size = torch.distributed.get_world_size() if self.is_distributed else 1

    def save_images(self, net_output_img_batch, names, epoch):
        """evaluate.py:49-66: <log_dirpath>/<split>/<epoch+1>/<name>, rank 0 only."""
        if self.local_rank != 0:
            return
        from PIL import Image
        epoch_dirpath = os.path.join(self.log_dirpath, self.split_name.lower(), str(epoch + 1))
        os.makedirs(epoch_dirpath, exist_ok=True)
        hwc = ops.f32chw_to_u8hwc(net_output_img_batch.detach().float().contiguous()).cpu().numpy()
        for i in range(hwc.shape[0]):
            name = names[i] if os.path.splitext(names[i])[1] else names[i] + ".png"
            Image.fromarray(hwc[i]).save(os.path.join(epoch_dirpath, name))

    @torch.no_grad()
    def evaluate(self, net, epoch=0, save_images=False):
        """evaluate.py:68-139: mean loss, mean masked PSNR (batches whose PSNR is undefined are left out of its mean)
        and mean MS-SSIM of the masked images over the split, summed over ranks like the reference's gathers."""
        was_training = net.training
        net.eval()
        device = next(net.parameters()).device
        self.msssim.to(device)
        acc = torch.zeros(5, dtype=torch.float64, device=device)  # loss sum, batches, psnr sum, psnr batches, msssim sum
        one = torch.ones((), dtype=torch.float64, device=device)
        for data in self.data_loader:
            img, gt, mask = (data[k].to(device, non_blocking=True) for k in ("input_img", "output_img", "mask"))
            out = net(img, mask)
            out = out[0] if isinstance(out, tuple) else out      # the curve model also returns its regulariser
            loss = self.criterion(out, gt, mask)                 # evaluate.py:101
            p = self.psnr(gt, out, mask)                         # evaluate.py:102
            ms = self.msssim(gt * mask, out * mask).mean()       # evaluate.py:103-104
            acc += torch.stack((loss.double(), one, p.double() if p is not None else 0 * one,
                                one if p is not None else 0 * one, ms.double()))
            if save_images:
                self.save_images(out.clamp(0, 1), data["name"], epoch)
        if self.is_distributed:
            torch.distributed.all_reduce(acc)
        net.train(was_training)
        return float(acc[0] / acc[1]), float(acc[2] / acc[3].clamp(min=1)), float(acc[4] / acc[1])
