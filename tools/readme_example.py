"""The usage snippet of README.md, runnable: python tools/readme_example.py (needs a HIP device)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curl_amd import colors, curves, metric, model
img  = torch.rand(4, 3, 1000, 1500, device="cuda")
mask = torch.ones(4, 1, 1000, 1500, dtype=torch.bool, device="cuda")
L, R, H = (torch.randn(4, n, device="cuda") * 0.1 for n in (48, 48, 64))
out, reg = model.CURLLayer()(img, mask, L, R, H)
lab      = colors.RGB2LAB()(img)
rgb, r   = curves.adjust_rgb(img, R)
psnr     = metric.PSNRMetric()(out, img, mask)
net  = model.TriSpaceRegNet(polynomial_order=4, spatial=True).cuda()
crit = model.CURLLoss().cuda()
crop, cmask = img[:, :, :256, :256].contiguous(), mask[:, :, :256, :256].contiguous()
loss = crit(net(crop, cmask), crop, cmask)
loss.backward()
from curl_amd import shard
both = torch.empty_like(img)
for rank in range(2):  # the split-pixels layout: every "rank" enhances its row slab of every image, in place
    shard.apply_row_slab(model.CURLLayer(), img, mask, L, R, H, rank, 2, out=both)
assert torch.equal(both, out)
print("ok", tuple(out.shape), float(psnr), float(loss.detach()))
