import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import curl_oracle as O
from curl_amd import ops
dev = torch.device("cuda:0")
real = np.load(os.path.join(ROOT, "tests/golden/real8.npz"))
g = torch.Generator().manual_seed(7)
u8 = real["dark_u8"]
img = O.u8hwc_to_f32chw(u8)[None]
B, _, H, W = img.shape
L, R, Hk = (torch.from_numpy(real["B_" + k]) for k in "LRH")
mask = torch.ones(B, 1, H, W, dtype=torch.bool)
w = torch.randn(B, 3, H, W, generator=g); wr = torch.rand(B, generator=g)
g64 = O.layer_gradients(img, mask.float(), L, R, Hk, w, wr)[0]
g32 = O.layer_gradients(img, mask.float(), L, R, Hk, w, wr, dtype=torch.float32)[0]
gi = ops.curl_layer_backward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev), w.to(dev), wr.to(dev))[0].cpu()
black = torch.from_numpy((u8 == 0).all(2))
for name, a in (("hip", gi), ("ref32", g32.float()), ("ref64", g64.float())):
    v = a[0][:, black]
    print(name, "at black px: mean |g| per channel", v.abs().mean(1).tolist(), "nonzero frac", float((v.abs().amax(0) > 0).float().mean()))
print("hip vs ref32 at black: max", float((gi - g32.float())[0][:, black].abs().max()), " elsewhere: max", float((gi - g32.float())[0][:, ~black].abs().max()))
print("ref32 vs ref64 at black: max", float((g32.double() - g64)[0][:, black].abs().max()))
# one black pixel in detail
i = black.nonzero()[0]
print("px", i.tolist(), "hip", gi[0, :, i[0], i[1]].tolist(), "ref32", g32[0, :, i[0], i[1]].tolist(), "ref64", g64[0, :, i[0], i[1]].tolist(), "w", w[0, :, i[0], i[1]].tolist())
