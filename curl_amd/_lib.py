"""ctypes binding of libcurlhip.so (include/curl_hip.h).

There is deliberately NO fallback: if the HIP library is missing the import of any
op fails loudly.  The CPU arithmetic of this path lives in the reference (and in
oracle/, which is test infrastructure); nothing here routes through either.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CURL_HIP_LIB", os.path.join(_HERE, "lib", "libcurlhip.so"))

# include/curl_hip.h
MASK_NONE, MASK_U8, MASK_F32 = 0, 1, 2
F_EXACT_ORDER = 0x1
F_PWL = 0x2
F_RESIDUAL_ONLY = 0x4
F_TUNE_UNROLL_SHIFT = 8
F_TUNE_BLOCK_SHIFT = 11
F_TUNE_XCD_SHIFT = 13
F_TUNE_OCC_SHIFT = 19  # resident workgroups per CU: 0 = per-operator default, 1 = no cap, 2..7
F_TUNE_PREP_SHIFT = 23  # where the curves are collapsed: 0 = by launch size, 1 = a launch of its own, 2 = inside the kernel
F_TUNE_NO_NT = 0x8000
F_DIAG_NO_MEM = 0x10000
F_DIAG_SKIP_PREP = 0x20000
F_WS_READY = 0x40000
F_MASK_FIRST = 0x400000  # foreground masks: fully masked-out wavefronts never read their pixels
MAX_KNOTS = 256

_c_f = ctypes.c_void_p  # device pointers travel as integers
_i = ctypes.c_int
_u = ctypes.c_uint
_sz = ctypes.c_size_t

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "curl_version": (_i, []),
    "curl_last_error": (ctypes.c_char_p, []),
    "curl_workspace_bytes": (_sz, [_i, _i]),
    "curl_apply_curve_f32": (_i, [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _i, _u, _c_f]),
    "curl_adjust_rgb_f32": (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _u, _c_f]),
    "curl_adjust_lab_f32": (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _u, _c_f]),
    "curl_adjust_hsv_f32": (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _u, _c_f]),
    "curl_rgb2lab_f32": (_i, [_c_f, _c_f, _i, _i, _i, _u, _c_f]),
    "curl_lab2rgb_f32": (_i, [_c_f, _c_f, _i, _i, _i, _u, _c_f]),
    "curl_rgb2hsv_f32": (_i, [_c_f, _c_f, _i, _i, _i, _u, _c_f]),
    "curl_hsv2rgb_f32": (_i, [_c_f, _c_f, _i, _i, _i, _u, _c_f]),
    "curl_lab_stage_f32": (_i, [_c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _u, _c_f]),
    "curl_hsv_stage_f32": (_i, [_c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _u, _c_f]),
    "curl_layer_fwd_f32": (_i, [_c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _sz,
                                _i, _i, _i, _i, _i, _i, _u, _c_f]),
    "curl_layer_fwd_slab_f32": (_i, [_c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _sz,
                                     _i, _i, _i, _i, _i, _i, _i, _i, _u, _c_f]),
    "curl_layer_bwd_scratch_bytes": (_sz, [_i, _i, _i]),
    "curl_layer_bwd_f32": (_i, [_c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f,
                                _c_f, _sz, _c_f, _sz, _i, _i, _i, _i, _i, _i, _u, _c_f]),
    "curl_trispace_fwd_f32": (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _u, _c_f]),
    "curl_trispace_fwd_slab_f32": (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _i, _u, _c_f]),
    "curl_trispace_fwd_u8hwc": (_i, [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _u, _c_f]),
    "curl_layer_fwd_u8hwc": (_i, [_c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _i, _i,
                                  _u, _c_f]),
    "curl_trispace_bwd_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "curl_trispace_bwd_f32": (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _u, _c_f]),
    "curl_poly_layer_f32": (_i, [_c_f, _c_f, _c_f, _i, _i, _i, _i, _c_f]),
    "curl_u8hwc_to_f32chw": (_i, [_c_f, _c_f, _i, _i, _i, _i, _c_f]),
    "curl_f32chw_to_u8hwc": (_i, [_c_f, _c_f, _i, _i, _i, _c_f]),
    "curl_psnr_scratch_bytes": (_sz, [_i, _i, _i]),
    "curl_psnr_f32": (_i, [_c_f, _c_f, _c_f, _i, _c_f, _c_f, _sz, _i, _i, _i, ctypes.c_float, _c_f]),
    "curl_msssim_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "curl_msssim_fwd_f32": (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _i, _c_f]),
    "curl_msssim_bwd_f32": (_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _i, _i, _c_f]),
    "curl_loss_terms_scratch_bytes": (_sz, [_i, _i, _i]),
    "curl_loss_terms_f32": (_i, [_c_f, _c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _sz, _i, _i, _i, _c_f]),
    "curl_loss_terms_bwd_f32": (_i, [_c_f, _c_f, _c_f, _i, _c_f, _c_f, _c_f, _i, _i, _i, _c_f]),
    "curl_layer_loss_fwd_f32": (_i, [_c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _sz, _c_f, _sz,
                                     _i, _i, _i, _i, _i, _i, _u, _c_f]),
    "curl_compose_white_u8hwc": (_i, [_c_f, _c_f, _i, _c_f, _i, _i, _i, _c_f]),
}

_lib = None


class CurlHipError(RuntimeError):
    """A HIP runtime error reported by libcurlhip.so (positive return code)."""


def load():
    """Load libcurlhip.so once.  Raises ImportError (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the HIP runtime of the process, so that the
    # device pointers and streams torch hands us belong to the runtime our kernels are launched on.  Loading
    # this library first would bind /opt/rocm's copy instead (observed: hipErrorNoDevice at the first launch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"curl_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_fast = None
# the entry points csrc/fastcall.cpp calls (their addresses in the library load() bound: CURL_HIP_LIB may name a variant)
FASTCALL_ABI = ("curl_layer_fwd_f32", "curl_layer_bwd_f32", "curl_trispace_fwd_u8hwc", "curl_workspace_bytes",
                "curl_layer_bwd_scratch_bytes")


def fast():
    """The compiled binding of the three hot entry points (curl_amd/_fastcall.*.so, csrc/fastcall.cpp), bound to the library
    load() loaded.  It is part of the build (__graft_entry__.build() asserts it loads); a tree without it warns once and uses
    the ctypes surface -- the same kernels.  CURL_NO_FASTCALL=1 selects the ctypes surface for every call (the A/B of the two
    host paths, tools/small_batch.py)."""
    global _fast
    if _fast is not None:
        return _fast or None
    if os.environ.get("CURL_NO_FASTCALL", "0") == "1":
        _fast = False
        return None
    lib = load()  # the kernels' library: absent -> ImportError, no way around it
    try:
        from . import _fastcall
    except ImportError as e:
        # The binding is host code over the SAME library: without it every call takes the ctypes surface (same kernels, same
        # results, ~10 us more host time per call).  Said once, loudly; tests/test_abi.py and tests/test_gpu_fastcall.py fail on it.
        import warnings
        warnings.warn(f"curl_amd: the compiled binding curl_amd/_fastcall.*.so is missing or does not load ({e}); calls go "
                      "through ctypes. Build it with `python -m curl_amd.build` (g++, a torch C++ extension).", RuntimeWarning)
        _fast = False
        return None
    _fastcall.bind_abi({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in FASTCALL_ABI})
    _fast = _fastcall
    return _fast


def check(rc, what):
    """Translate a C-ABI return code into the Python exception the reference's callers would see:
    argument errors -> ValueError (torch raises on bad shapes), HIP errors -> CurlHipError."""
    if rc == 0:
        return
    msg = load().curl_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise CurlHipError(f"{what}: {msg} (hipError {rc})")
