"""CURLLoss pointwise terms (model.py:78-116): oracle and kernel arithmetic (host twin) vs golden values computed
with the reference's colors.py and torch autograd."""
import numpy as np
import pytest
import torch

import curl_oracle as O


def terms_from_sums(sums, n_px_total):
    """Assemble the four loss terms of model.py:89-109 from per-image sums [B,5]."""
    s = sums.sum(0)
    unmasked = 3.0 * s[4]
    rgb, lab, hsv = s[0] / unmasked, s[2] / unmasked, s[3] / unmasked
    # (base + not mask).mean over the broadcast [B,B,H,W] == mean(base) + mean(not mask)
    cosine = 1.0 - s[1] / n_px_total - (n_px_total - s[4]) / n_px_total
    return rgb, cosine, lab, hsv


def test_oracle_matches_golden(golden):
    g = golden("loss")
    pred, tgt = torch.from_numpy(g["pred"]), torch.from_numpy(g["target"])
    for mk, m in (("bool", torch.from_numpy(g["mask"])), ("f32", torch.from_numpy(g["mask"]).float())):
        rgb, cosv, lab, hsv, Lp, Lt = O.curl_loss_terms(pred, tgt, m)
        for name, v in (("rgb", rgb), ("cos", cosv), ("lab", lab), ("hsv", hsv)):
            assert abs(float(v) - float(g[f"{mk}_{name}"])) <= 2e-7 * max(1.0, abs(float(v))), name
        assert np.abs(Lp.numpy() - g[f"{mk}_Lp"]).max() <= 2e-7


def test_twin_terms(twin, golden):
    g = golden("loss")
    m = g["mask"].astype(np.float32)
    sums, Lp, Lt = twin.loss_terms(g["pred"], g["target"], m)
    B, _, H, W = g["pred"].shape
    rgb, cosine, lab, hsv = terms_from_sums(sums, B * H * W)
    for name, v in (("rgb", rgb), ("cos", cosine), ("lab", lab), ("hsv", hsv)):
        assert abs(v - float(g[f"bool_{name}"])) <= 2e-6, (name, v, float(g[f"bool_{name}"]))
    assert np.abs(Lp - g["bool_Lp"]).max() <= 1e-6 and np.abs(Lt - g["bool_Lt"]).max() <= 1e-6


def test_twin_backward(twin, golden):
    """loss = 1.3 rgb + 0.7 cosine + 2.0 lab + 0.5 hsv + 1e-3 sum(Lp * wl): d/d pred vs autograd."""
    g = golden("loss")
    m = g["mask"].astype(np.float32)
    B, _, H, W = g["pred"].shape
    n, unmasked = B * H * W, 3.0 * m.sum()
    w = g["weights"]
    # weights of the per-pixel SUMS: d term / d sum
    w4 = np.array([w[0] / unmasked, -w[1] / n, w[2] / unmasked, w[3] / unmasked], np.float32)
    got = twin.loss_terms_bwd(g["pred"], g["target"], m, w4, g["wl"] * 1e-3)
    ref = g["bool_grad_pred"]
    d = np.abs(got - ref)
    scale = np.abs(ref).max()
    assert np.quantile(d, 0.995) <= 2e-4 * scale and d.max() <= 5e-2 * scale


def equal_and_black_case():
    """Pixel pairs a training step is full of and random floats never produce: prediction == target bit for bit (saturated
    highlights, black, a copied mid-tone), a BLACK prediction (the layer's clamp at 0) against a dark target, and the reverse."""
    g = torch.Generator().manual_seed(9)
    same = torch.tensor([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0.25, 0.5, 0.75], [1, 0, 0], [0.2, 0.2, 0.9]]).t()
    rnd = torch.rand(3, 10, generator=g)
    pred = torch.cat([same, rnd, torch.zeros(3, 8), torch.rand(3, 8, generator=g) * 0.1, torch.rand(3, 8, generator=g)], 1)
    tgt = torch.cat([same, rnd, torch.rand(3, 8, generator=g) * 0.1, torch.zeros(3, 8), torch.rand(3, 8, generator=g)], 1)
    n = pred.shape[1]
    return pred.reshape(1, 3, 1, n).contiguous(), tgt.reshape(1, 3, 1, n).contiguous(), torch.ones(1, 1, 1, n), len(same.t()) + 10


def oracle_loss_gradient(pred, tgt, mask, w, dtype):
    p = pred.detach().clone().to(dtype).requires_grad_(True)
    rgb, cosine, lab, hsv, _, _ = O.curl_loss_terms(p, tgt.to(dtype), mask.to(dtype))
    (w[0] * rgb + w[1] * cosine + w[2] * lab + w[3] * hsv).backward()
    return p.grad


def test_twin_backward_where_prediction_equals_target_and_at_black(twin):
    """torch.sign(0) = 0: where prediction and target are the same bits every L1 term's gradient vanishes, so the two colours
    must go through the same arithmetic (one converter for both; round 4 found the taped and the plain RGB2LAB an ulp apart).
    A black prediction: the clamp gate of model.py:55 on an L of exactly 0 passes the gradient in the reference."""
    # (on both twins, conftest.py: sign(0) must survive a*b+c contraction -- without the pragmas of curl_math_loss.h the
    # contracting one fails here, as the device did)
    pred, tgt, mask, n_same = equal_and_black_case()
    # (the cosine term is off here: at a black prediction its reference gradient is target / (1e-8 |target|) ~ 1e6 and would
    # be the whole scale; it has no gate at black and is covered by test_twin_backward)
    w = (1.3, 0.0, 2.0, 0.5)
    n = pred.shape[-1]
    for m in (mask, torch.full_like(mask, 0.37)):
        unmasked = 3.0 * float(m.sum())
        w4 = np.array([w[0] / unmasked, -w[1] / n, w[2] / unmasked, w[3] / unmasked], np.float32)
        g64 = oracle_loss_gradient(pred, tgt, m, w, torch.float64).numpy()
        g32 = oracle_loss_gradient(pred, tgt, m, w, torch.float32).numpy()
        got = twin.loss_terms_bwd(pred.numpy(), tgt.numpy(), m.numpy(), w4, None)
        scale = np.abs(g64).max()
        assert np.abs(g32 - g64).max() <= 1e-5 * scale, "the reference's float32 gradient is unambiguous here"
        d = np.abs(got - g64)[0, :, 0]
        assert d.max() <= 1e-5 * scale, (int(d.max(0).argmax()), float(d.max()), float(scale))
        assert np.abs(got[0, :, 0, :n_same]).max() <= 1e-6 * scale   # identical pixels: nothing is left


def test_twin_psnr_of_equal_images_is_infinite_under_any_mask(twin):
    """metric.py:35-62: equal images have zero squared error -- `10 log10(max^2 / 0) = +inf`, which nanmean keeps -- under a
    float mask with fractional values too: clamp(a) m - clamp(b) m is a difference of two equal ROUNDED products.  (On the
    contracting twin, as on the device in round 4, an fma(a, m, -(b m)) leaves a residue: 160 dB.)  And a plain case vs the oracle."""
    g = torch.Generator().manual_seed(3)
    a = torch.rand(1, 3, 9, 13, generator=g) * 1.4 - 0.2
    soft = torch.rand(1, 1, 9, 13, generator=g) * 0.9 + 0.05
    sse, ms = twin.psnr_sums(a[0].numpy(), a[0].clone().numpy(), soft[0, 0].numpy())
    assert sse == 0.0 and abs(ms - float(soft.sum())) < 1e-4
    b = torch.rand(1, 3, 9, 13, generator=g)
    sse, ms = twin.psnr_sums(a[0].numpy(), b[0].numpy(), soft[0, 0].numpy())
    want = float(O.psnr(a, b, soft))
    assert abs(10.0 * np.log10(1.0 / (sse / (3.0 * ms))) - want) < 1e-4


@pytest.mark.parametrize("tag", ["loss", "rgb", "w5"])
def test_msssim_host_mirror_vs_reference_golden(golden, tag):
    """curl_amd.metric.MSSSIMMetric (stock torch, separable window, no .cuda()) against outputs and gradients of
    the reference's own class (tests/golden/make_golden_msssim.py).  Runs on the CPU: nothing here is HIP."""
    import importlib.util
    import os
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # load curl_amd/metric.py without importing the package's HIP loader side (ops is only used by PSNRMetric)
    pkg = types.ModuleType("curl_amd_metric_only")
    pkg.__path__ = [os.path.join(root, "curl_amd")]
    sys.modules.setdefault("curl_amd_metric_only", pkg)
    sys.modules.setdefault("curl_amd_metric_only.ops", types.ModuleType("curl_amd_metric_only.ops"))
    spec = importlib.util.spec_from_file_location("curl_amd_metric_only.metric", os.path.join(root, "curl_amd", "metric.py"))
    metric = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(metric)
    g = golden("msssim")
    ws, ch = (int(v) for v in g[tag + "_cfg"])
    m = metric.MSSSIMMetric(window_size=ws, num_channel=ch)
    assert np.abs(m.gaussian_window.numpy() - g[tag + "_window"]).max() <= 1e-8
    assert list(m.state_dict().keys()) == ["msssim_weights"]  # what the reference's checkpoints hold for it
    a = torch.from_numpy(g[tag + "_a"]).requires_grad_(True)
    b = torch.from_numpy(g[tag + "_b"])
    out = m(a, b)
    assert np.abs(out.detach().numpy() - g[tag + "_out"]).max() <= 2e-6
    s, c = m.compute_ssim(a.detach(), b)
    assert np.abs(s.numpy() - g[tag + "_ssim"]).max() <= 2e-6 and np.abs(c.numpy() - g[tag + "_cs"]).max() <= 2e-6
    (out * torch.from_numpy(g[tag + "_w"])).sum().backward()
    ref = g[tag + "_grad_a"]
    assert np.abs(a.grad.numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
