"""Register budget of the hot kernels, read from hipcc's assembly of the product translation unit (cross-compiles without
a GPU, ~25 s).  Round 3 lost 6.7x on the polynomial path for one run because a change in a shared header made its kernel
spill 1.1 KB per lane -- no test saw it, only the bench did.  This one does: no product kernel may use scratch memory, and
the occupancy-defining VGPR counts are pinned (8 waves per SIMD for the fused curve kernels, 4 for the polynomial model,
3 for the layer backward)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    from curl_amd import build as B
    out = tmp_path_factory.mktemp("isa") / "curl.s"
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([HIPCC] + flags + ["-S", "--cuda-device-only", "-o", str(out), B.SRC],
                          stderr=subprocess.DEVNULL)
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", out.read_text(), flags=re.S):
        body = m.group(2)
        res[m.group(1)] = (int(re.search(r"next_free_vgpr (\d+)", body).group(1)),
                           int(re.search(r"private_segment_fixed_size (\d+)", body).group(1)))
    assert len(res) > 100
    return res


def test_no_product_kernel_spills(kernels):
    spilled = {k: v for k, v in kernels.items() if v[1] > 0}
    assert not spilled, spilled


@pytest.mark.parametrize("frag,max_vgprs", [
    ("stream_kernelI7OpLayerLi4ELi1ELi1ELb1ELi0E", 64),        # fused layer, bool mask: 8 waves per SIMD
    ("stream_kernelI10OpLabStageLi4ELi1ELi1ELb1ELi0E", 64),
    ("stream_kernelI10OpHsvStageLi4ELi2ELi1ELb1ELi0E", 64),
    ("stream_kernelI14OpTriSpaceRowsLi4ELi1ELi0ELb1ELi0E", 128),  # polynomial model: 4 waves per SIMD
    ("layer_bwd_kernelILi4ELi1E", 168),                          # layer backward: 3 waves per SIMD
])
def test_vgpr_budget(kernels, frag, max_vgprs):
    hits = {k: v for k, v in kernels.items() if frag in k}
    assert hits, frag
    for k, (vgprs, _) in hits.items():
        assert vgprs <= max_vgprs, (k, vgprs)
