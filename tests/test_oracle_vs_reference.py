"""Pins the oracle: every function of oracle/curl_oracle.py against the reference's own modules,
bit for bit, on seeded inputs incl. the edge cases of SURVEY.md section 4.  Build container only
(skips where /root/reference is absent)."""
import numpy as np
import pytest
import torch

import curl_oracle as O


def rnd(seed, *shape):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("K", [2, 3, 4, 16, 17, 18, 19, 40])
def test_apply_curve_bitexact(reference_modules, K):
    curves = reference_modules["curves"]
    g = torch.Generator().manual_seed(K)
    B, H, W = 2, 19, 23  # H*W odd: not a multiple of 4
    for sigma in (0.1, 1.0):
        C = torch.exp(torch.randn(B, K, generator=g) * sigma)
        for ci, co in [(0, 0), (0, 1), (1, 1), (2, 2), (2, 0), (1, 2)]:
            x = torch.rand(B, 3, H, W, generator=g) * 2 - 0.5
            r0 = torch.rand(B, generator=g)
            a, ra = curves.apply_curve(x, C, r0.clone(), ci, co)
            b, rb = O.apply_curve(x, C, r0.clone(), ci, co)
            assert torch.equal(a, b) and torch.equal(ra, rb)


def test_apply_curve_mutates_regulariser_in_place(reference_modules):
    curves = reference_modules["curves"]
    x, C = rnd(0, 1, 3, 4, 4), torch.exp(rnd(1, 1, 16))
    r1, r2 = torch.zeros(1), torch.zeros(1)
    _, out1 = curves.apply_curve(x, C, r1, 0, 0)
    _, out2 = O.apply_curve(x, C, r2, 0, 0)
    assert out1 is r1 and out2 is r2 and torch.equal(r1, r2) and r1.item() > 0


@pytest.mark.parametrize("name,ofn", [("RGB2LAB", O.rgb2lab), ("LAB2RGB", O.lab2rgb), ("RGB2HSV", O.rgb2hsv),
                                      ("HSV2RGB", O.hsv2rgb)])
def test_converters_bitexact(reference_modules, golden, name, ofn):
    mod = getattr(reference_modules["colors"], name)()
    g = golden("converters")
    inputs = [rnd(3, 2, 3, 17, 31), rnd(4, 2, 3, 17, 31) * 3 - 1,
              torch.randint(0, 256, (1, 3, 32, 32), generator=torch.Generator().manual_seed(5)).float() / 255]
    inputs += [torch.from_numpy(g[k]) for k in ("rgb_in", "lab_in", "hsv_in", "rgbwide_in")]
    for x in inputs:
        keep = x.clone()
        assert torch.equal(mod(x), ofn(x))
        assert torch.equal(x, keep), "converter must not modify its input"


def test_wrapper_bugs_still_present(reference_modules):
    """The semantics this repo implements exist BECAUSE these raise."""
    curves = reference_modules["curves"]
    x = rnd(0, 1, 3, 4, 4)
    for fn, n in ((curves.adjust_rgb, 48), (curves.adjust_lab, 48), (curves.adjust_hsv, 64)):
        with pytest.raises(TypeError):
            fn(x, torch.zeros(1, n))


def test_psnr_and_transpose(reference_modules):
    metric, transpose = reference_modules["metric"], reference_modules["transpose"]
    a, b = rnd(1, 2, 3, 9, 11), rnd(2, 2, 3, 9, 11) * 1.4 - 0.2
    m = (rnd(3, 2, 1, 9, 11) > 0.3).float()
    assert torch.equal(metric.PSNRMetric()(a, b, m), O.psnr(a, b, m))
    for arr in (rnd(4, 3, 5, 7).numpy(), rnd(5, 2, 3, 5, 7).numpy()):
        assert np.array_equal(transpose.swapimdims_3HW_HW3(arr), O.chw_to_hwc(arr))
        hwc = np.ascontiguousarray(O.chw_to_hwc(arr))
        assert np.array_equal(transpose.swapimdims_HW3_3HW(hwc), O.hwc_to_chw(hwc))
