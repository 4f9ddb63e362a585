import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curl_amd import ops
dev = torch.device("cuda:0")
for (B, H, W) in ((3, 64, 64), (2, 250, 301), (2, 250, 300)):
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    w = torch.randn(B, 3, H, W, generator=g).to(dev)
    wr = torch.randn(B, generator=g).to(dev)
    L, R, Hk = ((torch.randn(B, n, generator=g) * 0.3).to(dev) for n in (48, 48, 64))
    soft = torch.rand(B, 1, H, W, generator=g).to(dev)
    ws = ops.curl_layer_forward(img, soft, L, R, Hk, return_workspace=True)[2]
    for kw in ({}, {"workspace": ws}):
        for wr_ in (None, wr):
            full = ops.curl_layer_backward(img, soft, L, R, Hk, w, wr_, **kw)
            full2 = ops.curl_layer_backward(img, soft, L, R, Hk, w, wr_, **kw)
            only = ops.curl_layer_backward(img, soft, L, R, Hk, w, wr_, need_grad_img=False, **kw)
            only2 = ops.curl_layer_backward(img, soft, L, R, Hk, w, wr_, need_grad_img=False, **kw)
            for nm, a, a2, b, b2 in zip("LRH", full[1:], full2[1:], only[1:], only2[1:]):
                d = (a - b).abs().view(B, -1, 16)
                print((B, H, W), list(kw), wr_ is not None, nm, "full==full", torch.equal(a, a2), "only==only", torch.equal(b, b2),
                      "curves that differ:", sorted(set(map(tuple, d.amax(2).nonzero().tolist()))), "max", float(d.max()), "of", float(a.abs().max()))
