"""The C-ABI library loads (no GPU needed) and exports exactly what include/curl_hip.h declares;
argument errors are reported through return codes + curl_last_error, never by crashing."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "curl_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|size_t|const char\*)\s+(curl_\w+)\s*\(", src, flags=re.M)
    assert len(names) >= 15
    return names


def test_library_exports_every_declared_symbol():
    from curl_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in curl_hip.h but not exported"
    assert sorted(names) == sorted(_lib.SIGNATURES), "ctypes table and header disagree"


def test_constants_match_header():
    from curl_amd import _lib
    src = open(HEADER).read()
    defs = dict(re.findall(r"#define\s+(CURL_\w+)\s+(0x[0-9a-fA-F]+|\d+)u?", src))
    assert int(defs["CURL_MASK_U8"]) == _lib.MASK_U8 and int(defs["CURL_MASK_F32"]) == _lib.MASK_F32
    assert int(defs["CURL_F_EXACT_ORDER"], 16) == _lib.F_EXACT_ORDER and int(defs["CURL_F_PWL"], 16) == _lib.F_PWL
    assert int(defs["CURL_MAX_KNOTS"]) == _lib.MAX_KNOTS
    assert int(defs["CURL_F_TUNE_NO_NT"], 16) == _lib.F_TUNE_NO_NT
    assert int(defs["CURL_F_DIAG_NO_MEM"], 16) == _lib.F_DIAG_NO_MEM
    assert int(defs["CURL_F_MASK_FIRST"], 16) == _lib.F_MASK_FIRST
    assert int(defs["CURL_F_WS_READY"], 16) == _lib.F_WS_READY and int(defs["CURL_F_DIAG_SKIP_PREP"], 16) == _lib.F_DIAG_SKIP_PREP
    for name in ("UNROLL", "BLOCK", "XCD", "OCC", "PREP"):   # tuning fields: shift and mask agree, fields do not overlap each other or the flags
        shift, mask = int(defs[f"CURL_F_TUNE_{name}_SHIFT"]), int(defs[f"CURL_F_TUNE_{name}_MASK"], 16)
        assert shift == getattr(_lib, f"F_TUNE_{name}_SHIFT") and mask >> shift << shift == mask and (mask >> shift) & 1
    fields = [int(defs[f"CURL_F_TUNE_{n}_MASK"], 16) for n in ("UNROLL", "BLOCK", "XCD", "OCC", "PREP")] + [
        int(defs[n], 16) for n in ("CURL_F_EXACT_ORDER", "CURL_F_PWL", "CURL_F_RESIDUAL_ONLY", "CURL_F_TUNE_NO_NT", "CURL_F_DIAG_NO_MEM",
                                    "CURL_F_DIAG_SKIP_PREP", "CURL_F_WS_READY", "CURL_F_MASK_FIRST")]
    assert sum(fields) == __import__("functools").reduce(lambda a, b: a | b, fields), "flag bits overlap"


def test_version_and_workspace_size():
    from curl_amd import _lib
    lib = _lib.load()
    assert lib.curl_version() >= 100
    assert lib.curl_workspace_bytes(32, 160) >= 32 * (20 + 160) * 4
    assert lib.curl_workspace_bytes(0, 160) == 0 and lib.curl_workspace_bytes(1, 0) > 0


def test_argument_errors_are_codes_not_crashes():
    """Validation happens before any HIP call, so it can be exercised without a GPU."""
    from curl_amd import _lib
    lib = _lib.load()
    assert lib.curl_rgb2lab_f32(None, None, 1, 4, 4, 0, None) == -1  # CURL_E_NULL
    assert b"NULL" in lib.curl_last_error()
    fake = ctypes.c_void_p(4096)
    assert lib.curl_rgb2lab_f32(fake, fake, 0, 4, 4, 0, None) == -2  # CURL_E_SHAPE
    assert lib.curl_rgb2lab_f32(fake, fake, 1, 4, 4, 0x2, None) == -6  # flag not valid for converters
    assert lib.curl_apply_curve_f32(fake, fake, fake, None, 1, 4, 4, 1, 0, 0, 0, None) == -3  # K < 2
    assert lib.curl_apply_curve_f32(fake, fake, fake, None, 1, 4, 4, 16, 3, 0, 0, None) == -2  # channel
    assert lib.curl_apply_curve_f32(fake, fake, fake, None, 1, 4, 4, 16, 0, 0, 0x3, None) == -6  # exclusive flags
    assert lib.curl_adjust_rgb_f32(fake, fake, fake, None, None, 0, 1, 4, 4, 16, 0, None) == -4  # workspace
    assert lib.curl_adjust_rgb_f32(fake, fake, fake, None, fake, 8, 1, 4, 4, 16, 0, None) == -4  # too small
    assert lib.curl_layer_fwd_f32(fake, None, 1, fake, fake, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, 16, 16, 0,
                                  None) == -5  # mask_kind set, mask NULL
    assert lib.curl_layer_fwd_f32(fake, None, 7, fake, fake, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, 16, 16, 0,
                                  None) == -5
    assert lib.curl_layer_fwd_f32(fake, None, 0, fake, fake, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, 300, 16, 0,
                                  None) == -3  # K > CURL_MAX_KNOTS
    assert lib.curl_u8hwc_to_f32chw(fake, fake, 1, 4, 4, 2, None) == -2  # Cin
    assert lib.curl_hsv_stage_f32(fake, None, 1, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, 0, None) == -5  # mask NULL
    assert lib.curl_hsv_stage_f32(fake, None, 0, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, 0x2, None) == -6  # no PWL
    assert lib.curl_hsv_stage_f32(fake, None, 0, fake, fake, None, fake, 8, 1, 4, 4, 16, 0, None) == -4  # workspace
    # round 4: the curve-collapse placement field takes 0, 1, 2; CURL_F_DIAG_SKIP_PREP belongs to the two layer-forward entries
    assert lib.curl_layer_fwd_f32(fake, None, 0, fake, fake, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, 16, 16, 3 << 23,
                                  None) == -6 and b"placement" in lib.curl_last_error()
    assert lib.curl_lab_stage_f32(fake, None, 0, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, _lib.F_DIAG_SKIP_PREP, None) == -6
    assert lib.curl_adjust_rgb_f32(fake, fake, fake, None, fake, 1 << 20, 1, 4, 4, 16, _lib.F_DIAG_SKIP_PREP, None) == -6
    assert lib.curl_layer_bwd_f32(fake, None, 0, fake, fake, fake, fake, None, None, fake, fake, fake, fake, 1 << 20, fake, 1 << 20,
                                  1, 4, 4, 16, 16, 16, _lib.F_DIAG_SKIP_PREP, None) == -6
    with pytest.raises(ValueError):
        _lib.check(-2, "x")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No fallback: a missing .so is an ImportError naming the build command."""
    import importlib
    from curl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    monkeypatch.undo()
    importlib.reload(_lib)
    _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "curl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "curl_oracle" not in txt and "libcurl_twin" not in txt, f


def test_row_slab_entries_validate_before_any_hip_call():
    """curl_layer_fwd_slab_f32 / curl_trispace_fwd_slab_f32: the slab must lie inside the image (CURL_E_SHAPE = -2)."""
    from curl_amd import _lib
    lib = _lib.load()
    fake = ctypes.c_void_p(4096)
    for row0, rows in ((-1, 2), (0, 0), (6, 4), (8, 1)):
        assert lib.curl_layer_fwd_slab_f32(fake, None, 0, fake, fake, fake, fake, None, fake, 1 << 20, 1, 8, 8, row0, rows,
                                           16, 16, 16, 0, None) == -2
        assert b"slab" in lib.curl_last_error()
        assert lib.curl_trispace_fwd_slab_f32(fake, fake, fake, 1, 8, 8, row0, rows, 126, 0, None) == -2
    assert lib.curl_trispace_fwd_slab_f32(fake, fake, fake, 1, 8, 8, 0, 4, 100, 0, None) == -3  # num_coeffs
    assert lib.curl_layer_fwd_slab_f32(fake, None, 0, fake, fake, fake, fake, None, fake, 1 << 20, 1, 8, 8, 0, 4,
                                       16, 16, 16, 0x3, None) == -6  # flags


def test_polynomial_entries_refuse_a_misaligned_coefficient_table():
    """The polynomial kernels copy an image's coefficient table into LDS as 8-byte pairs: a table at an odd float address is
    an argument error (CURL_E_SHAPE = -2) before any launch; the Python wrappers copy such a view instead."""
    from curl_amd import _lib
    lib = _lib.load()
    fake, odd = ctypes.c_void_p(4096), ctypes.c_void_p(4096 + 4)
    assert lib.curl_trispace_fwd_f32(fake, odd, fake, 1, 8, 8, 126, 0, None) == -2
    assert b"8-byte aligned" in lib.curl_last_error()
    assert lib.curl_trispace_fwd_slab_f32(fake, odd, fake, 1, 8, 8, 0, 4, 126, 0, None) == -2
    assert lib.curl_trispace_fwd_u8hwc(fake, odd, None, fake, 1, 8, 8, 126, 0, None) == -2
    assert lib.curl_trispace_bwd_f32(fake, odd, fake, fake, fake, 1 << 30, 1, 8, 8, 126, 0, None) == -2
    src = open(os.path.join(ROOT, "curl_amd", "ops.py")).read()
    assert src.count("coeffs.to(torch.float32).contiguous()") == 1 and src.count("_coeffs32(coeffs") >= 5  # one helper, used everywhere
    # only the 126-coefficient tables are read as pairs; an odd-float 35-coefficient table is not refused on
    # alignment (with NULL images the first complaint is the image pointer, never "aligned")
    three = ctypes.c_void_p(4096 + 2)
    assert lib.curl_trispace_fwd_f32(fake, three, fake, 1, 8, 8, 35, 0, None) == -2 and b"4-byte" in lib.curl_last_error()


def test_public_header_compiles_as_c99_and_cxx(tmp_path):
    """include/curl_hip.h is the boundary a cgo / JNI / ctypes maintainer reads: plain C, no torch or HIP types."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not present")
    src = tmp_path / "h.c"
    src.write_text('#include "curl_hip.h"\nint main(void) { return curl_version() > 0 ? 0 : CURL_K_UNEVEN(16, 14) + (int)CURL_F_WS_READY; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-I", inc, "-o", str(tmp_path / "c.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-x", "c++", "-c", str(src), "-I", inc, "-o", str(tmp_path / "cc.o")])
    txt = open(HEADER).read()
    assert "torch" not in txt.lower().replace("pytorch's caching allocator", "") or "at::" not in txt
    assert "hip/hip_runtime" not in txt


def test_fastcall_binding_builds_loads_and_declines_cpu_tensors():
    """The compiled binding of the three hot entry points (csrc/fastcall.cpp): built in-tree next to the library, bound to the
    addresses of the library _lib loaded, and -- without a GPU -- answering None (= "take the checked path") for CPU tensors.
    No compute call happens here."""
    import torch
    from curl_amd import _lib
    from curl_amd.build import build_fastcall, fastcall_path
    assert build_fastcall() == fastcall_path()
    assert set(_lib.FASTCALL_ABI) <= set(_lib.SIGNATURES)
    fast = _lib.fast()
    assert fast is not None and {"layer_fwd", "layer_bwd", "trispace_fwd_u8hwc", "bind_abi"} <= set(dir(fast))
    img, k48, k64 = torch.zeros(1, 3, 4, 4), torch.zeros(1, 48), torch.zeros(1, 64)
    assert fast.layer_fwd(img, None, k48, k48, k64, 0) is None
    assert fast.layer_bwd(img, None, k48, k48, k64, img, None, True, None, 0) is None
    assert fast.trispace_fwd_u8hwc(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), torch.zeros(1, 3, 3, 126), None) is None
