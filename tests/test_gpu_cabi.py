"""The C ABI without PyTorch: tests/cabi/host_example.cpp (hipMalloc'd buffers, its own stream, libcurlhip.so linked like any
shared library) is built with hipcc on the GPU box and must produce, bit for bit, what the Python surface produces."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_standalone_cxx_host_of_the_c_abi(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not present")
    from curl_amd import _lib, ops
    _lib.load()
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "host_example"
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "host_example.cpp"), "-L", libdir, "-lcurlhip",
                           f"-Wl,-rpath,{libdir}", "-o", str(exe)], stderr=subprocess.DEVNULL)
    g = torch.Generator().manual_seed(99)
    B, H, W = 2, 60, 100
    img = torch.rand(B, 3, H, W, generator=g)
    mask = torch.rand(B, 1, H, W, generator=g) > 0.25
    L, R, Hk = (torch.randn(B, n, generator=g) * 0.1 for n in (48, 48, 64))
    gout = torch.rand(B, 3, H, W, generator=g)
    for name, t in (("img.f32", img), ("L.f32", L), ("R.f32", R), ("H.f32", Hk), ("gout.f32", gout)):
        t.numpy().astype(np.float32).tofile(tmp_path / name)
    mask.numpy().astype(np.uint8).tofile(tmp_path / "mask.u8")
    p = subprocess.run([str(exe), str(tmp_path), str(B), str(H), str(W)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "curl_version" in p.stdout and "-> -5" in p.stdout  # CURL_E_MASK came back as a code + message
    dev = torch.device("cuda:0")
    out, reg = ops.curl_layer_forward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev))
    gi, gL, gR, gH = ops.curl_layer_backward(img.to(dev), mask.to(dev), L.to(dev), R.to(dev), Hk.to(dev), gout.to(dev))
    for name, t in (("out.f32", out), ("reg.f32", reg), ("gimg.f32", gi), ("gL.f32", gL), ("gR.f32", gR), ("gH.f32", gH)):
        got = np.fromfile(tmp_path / name, dtype=np.float32).reshape(tuple(t.shape))
        assert np.array_equal(got, t.cpu().numpy()), name
