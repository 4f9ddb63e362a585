"""Build libcurlhip.so for gfx950 with hipcc (cross-compiles without a GPU), and the compiled host-side binding of its three
hot entry points (csrc/fastcall.cpp -> curl_amd/_fastcall.*.so: a torch C++ extension, host code only, g++).

    python -m curl_amd.build            # or __graft_entry__.build()
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "curl_kernels.hip")
INCLUDE = os.path.join(os.path.dirname(HERE), "include", "curl_hip.h")


def _deps():
    """Everything the one translation unit reads: csrc/*.h|.hip|.inc, csrc/kernels/*.inc, the public header."""
    out = [INCLUDE, os.path.abspath(__file__)]  # this file too: a change of FLAGS rebuilds
    for d, _, files in os.walk(os.path.join(HERE, "csrc")):
        out += [os.path.join(d, f) for f in files if f.endswith((".h", ".hip", ".inc"))]
    return out

OUT = os.path.join(HERE, "lib", "libcurlhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP vectoriser packs independent scalar float chains into v_pk_*_f32 with a shuffle per operand
# -- the packed forms this library wants are written out (curl_math_poly.h); without the pass the CURLLoss kernels are 5 %
# faster and nothing is slower (profiles/r02/noslp_ab.log)
# -ffp-contract=fast-honor-pragmas: hipcc's default for device code, spelled out because results depend on it -- products are
# fused into fmas across statements, EXCEPT where a block says `#pragma clang fp contract(off)` (differences that feed a sign
# or must cancel exactly: curl_math_loss.h, curl_math_bwd.h).  Plain `fast` would ignore those pragmas.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-fno-math-errno", "-fno-slp-vectorize",
         "-ffp-contract=fast-honor-pragmas"]


def build(force=False, verbose=False):
    """Compile curl_amd/csrc/curl_kernels.hip -> curl_amd/lib/libcurlhip.so (in-tree, so it travels with the repo)."""
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in _deps()):
        return OUT
    cmd = [HIPCC] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-o", OUT, SRC]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libcurlhip.so")
    if verbose:
        sys.stderr.write(res.stderr)
    return OUT


FAST_SRC = os.path.join(HERE, "csrc", "fastcall.cpp")


def fastcall_path():
    import sysconfig
    return os.path.join(HERE, "_fastcall" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_fastcall(force=False):
    """g++ csrc/fastcall.cpp -> curl_amd/_fastcall.<abi>.so against the torch this interpreter imports (host code: tensors in,
    the C ABI underneath; the kernels' library is not linked -- _lib.py hands its function addresses over at import)."""
    import sysconfig
    import torch
    out = fastcall_path()
    deps = [FAST_SRC, INCLUDE, os.path.abspath(__file__)]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    tdir = os.path.dirname(torch.__file__)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_fastcall", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include", "-I/opt/rocm/include",
           f"-I{sysconfig.get_paths()['include']}", FAST_SRC, "-o", out,
           f"-L{tdir}/lib", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_python", f"-Wl,-rpath,{tdir}/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("g++ failed building the _fastcall binding")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_fastcall(force="--force" in sys.argv))
