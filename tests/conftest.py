import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("CURL_REFERENCE", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        path = os.path.join(GOLDEN, name + ".npz")
        if name == "real8" and not os.path.exists(path):
            # the one fixture that holds third-party pixels (tests/golden/README.md): a tree shipped without it loses these
            # tests, nothing else
            pytest.skip("tests/golden/real8.npz not present (tests/golden/README.md)")
        return np.load(path)
    return load


@pytest.fixture(scope="session")
def reference_modules():
    """The reference's own curves/colors/transpose/metric modules -- build container only."""
    if not os.path.isfile(os.path.join(REFERENCE, "curves.py")):
        pytest.skip("reference tree not present (it never travels to the GPU box)")
    sys.path.insert(0, REFERENCE)
    try:
        import colors
        import curves
        import metric
        import transpose
    finally:
        sys.path.remove(REFERENCE)
    return {"curves": curves, "colors": colors, "transpose": transpose, "metric": metric}


HIP_CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="session", params=["rounding", "contracting"])
def twin(request):
    """Test-only host build of curl_amd/csrc/curl_math.h (tests/twin/curl_twin.cpp), twice; every twin test runs on both.
    rounding:    g++, NO a*b+c contraction -- every operation rounds where the source says.
    contracting: hipcc's own clang for the host with ITS default for HIP device code, -ffp-contract=fast-honor-pragmas: products
                 are fused into fmas across statements wherever the optimiser likes, as in the kernels, and `#pragma clang fp
                 contract(off)` is honoured as there (plain `fast`, CUDA's default, lets the backend fuse whatever the pragmas
                 say).  Not the device's instruction selection, but the same middle end making the same kind of choice: the
                 CPU-side net for round 4's class of bug -- sign(a*b - c*d) of EQUAL products coming out +-1
                 (curl_math_loss.h) -- which the rounding twin cannot see."""
    if request.param == "rounding":
        return _twin("libcurl_twin.so", ["g++", "-O2", "-mfma", "-ffp-contract=off"])
    if not os.path.exists(HIP_CLANG):
        pytest.skip("hipcc's clang is not installed here")
    return _twin("libcurl_twin_contracting.so", [HIP_CLANG, "-O2", "-mfma", "-ffp-contract=fast-honor-pragmas"])


def _twin(name, compile_cmd):
    src = os.path.join(ROOT, "tests", "twin", "curl_twin.cpp")
    hdr = os.path.join(ROOT, "curl_amd", "csrc", "curl_math.h")
    hdr2 = os.path.join(ROOT, "curl_amd", "csrc", "curl_math_bwd.h")
    hdr3 = os.path.join(ROOT, "curl_amd", "csrc", "curl_math_poly.h")
    hdr4 = os.path.join(ROOT, "curl_amd", "csrc", "curl_math_loss.h")
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, name)
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(hdr2), os.path.getmtime(hdr3), os.path.getmtime(hdr4)):
        subprocess.check_call(compile_cmd + ["-fPIC", "-shared", "-std=c++17", "-DCURL_HOST_TWIN", "-Wno-unknown-pragmas", "-o", so, src])
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)

    def P(a):
        return None if a is None else a.ctypes.data_as(fp)

    def f32(a):
        return None if a is None else np.ascontiguousarray(a, dtype=np.float32)

    class Twin:
        @staticmethod
        def convert(op, x):
            x = f32(x)
            out = np.empty_like(x)
            B, _, H, W = x.shape
            lib.twin_convert({"rgb2lab": 0, "lab2rgb": 1, "rgb2hsv": 2, "hsv2rgb": 3}[op], P(x), P(out), B,
                             ctypes.c_long(H * W))
            return out

        @staticmethod
        def apply_curve(img, C, reg, cin, cout, mode):
            img, C = f32(img), f32(C)
            out = np.empty_like(img)
            reg = None if reg is None else f32(reg).copy()
            B, _, H, W = img.shape
            lib.twin_apply_curve(P(img), P(C), P(out), P(reg), B, ctypes.c_long(H * W), C.shape[1], cin, cout, mode)
            return out, reg

        @staticmethod
        def adjust(ncurves, img, raw, mode=0):
            img, raw = f32(img), f32(raw)
            out = np.empty_like(img)
            B, _, H, W = img.shape
            reg = np.empty(B, np.float32)
            lib.twin_adjust(ncurves, P(img), P(raw), P(out), P(reg), B, ctypes.c_long(H * W), raw.shape[1] // ncurves,
                            mode)
            return out, reg

        @staticmethod
        def layer(stage, img, mask, L, R, H, binary=False):
            img, mask, L, R, H = f32(img), f32(mask), f32(L), f32(R), f32(H)
            out = np.empty_like(img)
            B, _, Hh, W = img.shape
            reg = np.empty(B, np.float32)
            lib.twin_layer(stage, P(img), P(mask), P(L), P(R), P(H), P(out), P(reg), B, ctypes.c_long(Hh * W),
                           L.shape[1] // 3, R.shape[1] // 3, H.shape[1] // 4, int(binary))
            return out, reg

        @staticmethod
        def layer_bwd(img, mask, L, R, H, gout, greg, binary=False):
            img, mask, L, R, H, gout, greg = (f32(a) for a in (img, mask, L, R, H, gout, greg))
            B, _, Hh, W = img.shape
            gimg = np.empty_like(img)
            gL, gR, gH = np.empty_like(L), np.empty_like(R), np.empty_like(H)
            lib.twin_layer_bwd(P(img), P(mask), P(L), P(R), P(H), P(gout), P(greg), P(gimg), P(gL), P(gR), P(gH), B,
                               ctypes.c_long(Hh * W), L.shape[1] // 3, R.shape[1] // 3, H.shape[1] // 4, int(binary))
            return gimg, gL, gR, gH

        @staticmethod
        def trispace(img, coeffs, residual_only=False):
            img, coeffs = f32(img), f32(coeffs)
            out = np.empty_like(img)
            B, _, Hh, W = img.shape
            V = 5 if coeffs.shape[-1] == 126 else 3
            lib.twin_trispace(P(img), P(coeffs), P(out), B, Hh, W, V, int(residual_only))
            return out

        @staticmethod
        def trispace_rows(img, coeffs, residual_only=False):
            img, coeffs = f32(img), f32(coeffs)
            B, _, Hh, W = img.shape
            out = np.empty_like(img)
            lib.twin_trispace_rows(P(img), P(coeffs), P(out), B, Hh, W, int(residual_only))
            return out

        @staticmethod
        def poly_layer(img, coeffs):
            img, coeffs = f32(img), f32(coeffs)
            B, V, Hh, W = img.shape
            out = np.empty((B, 3, Hh, W), np.float32)
            lib.twin_poly_layer(P(img), P(coeffs), P(out), B, ctypes.c_long(Hh * W), V)
            return out

        @staticmethod
        def loss_terms(pred, tgt, mask):
            pred, tgt, mask = f32(pred), f32(tgt), f32(mask)
            B, _, Hh, W = pred.shape
            sums = np.zeros((B, 5), np.float64)
            Lp, Lt = np.empty((B, 1, Hh, W), np.float32), np.empty((B, 1, Hh, W), np.float32)
            lib.twin_loss_terms(P(pred), P(tgt), P(mask), sums.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), P(Lp), P(Lt),
                                B, ctypes.c_long(Hh * W))
            return sums, Lp, Lt

        @staticmethod
        def loss_terms_bwd(pred, tgt, mask, w4, gLp):
            pred, tgt, mask, w4, gLp = f32(pred), f32(tgt), f32(mask), f32(w4), f32(gLp)
            B, _, Hh, W = pred.shape
            g = np.empty_like(pred)
            lib.twin_loss_terms_bwd(P(pred), P(tgt), P(mask), P(w4), P(gLp), P(g), B, ctypes.c_long(Hh * W))
            return g

        @staticmethod
        def trispace_bwd(img, coeffs, gout, residual_only=False):
            img, coeffs, gout = f32(img), f32(coeffs), f32(gout)
            B, _, Hh, W = img.shape
            V = 5 if coeffs.shape[-1] == 126 else 3
            g = np.empty_like(coeffs)
            lib.twin_trispace_bwd(P(img), P(coeffs), P(gout), P(g), B, Hh, W, V, int(residual_only))
            return g

        @staticmethod
        def trispace_bwd_foldx(img, coeffs, gout, residual_only=False):
            img, coeffs, gout = f32(img), f32(coeffs), f32(gout)
            B, _, Hh, W = img.shape
            g = np.empty_like(coeffs)
            lib.twin_trispace_bwd_foldx(P(img), P(coeffs), P(gout), P(g), B, Hh, W, int(residual_only))
            return g

        @staticmethod
        def div_small_mismatches(dmax):
            lib.twin_div_small_mismatches.restype = ctypes.c_long
            return int(lib.twin_div_small_mismatches(int(dmax)))

        @staticmethod
        def psnr_sums(a, b, mask):
            """(sum of squared masked errors, sum of mask) of ONE image [3,H,W] (psnr.inc's per-pixel term)."""
            a, b, mask = f32(a), f32(b), f32(mask)
            sse, ms = ctypes.c_double(), ctypes.c_double()
            lib.twin_psnr_sums(P(a), P(b), P(mask), ctypes.c_long(a.shape[-1] * a.shape[-2]), ctypes.byref(sse), ctypes.byref(ms))
            return sse.value, ms.value

        @staticmethod
        def u8_edges(x):
            x = f32(x).ravel()
            unit = np.empty(256, np.float32)
            q = np.empty(x.size, np.uint8)
            lib.twin_u8_edges(P(unit), P(x), q.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
            return unit, q

    return Twin


def max_err(a, b):
    """max|a-b| / max(1, max|b|) -- absolute error in image units."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
