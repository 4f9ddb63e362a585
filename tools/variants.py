"""Experiment builds of libcurlhip.so (never the product library): each variant is the same translation unit
compiled with extra -D switches into curl_amd/lib/variants/libcurlhip_<name>.so.  Built in the CPU container
(hipcc cross-compiles gfx950) so the .so files travel to the GPU box with the snapshot.

    python tools/variants.py                 # build all
    python tools/variants.py stamp nofence   # build some
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import build as B  # noqa: E402

OUT_DIR = os.path.join(ROOT, "curl_amd", "lib", "variants")

# name -> extra compiler switches
VARIANTS = {
    "base": [],
    # in-kernel s_memtime / s_memrealtime stamps per wave (tools/stamp.py)
    "stamp": ["-DCURL_DIAG_STAMP"],
    # loads issued, arithmetic on synthetic values: separates "waiting for data" from "sharing the chip with traffic"
    "nodep": ["-DCURL_DIAG_NO_DEP"],
    # scheduling fences around the transcendental runs removed
    "nofence": ["-DCURL_NO_FENCE"],
    # dual-issue experiments (curl_math.h): issue priority by phase, constants / coefficients in VGPRs
    "prio1": ["-DCURL_EXP_PRIO=1"],
    "prio3": ["-DCURL_EXP_PRIO=3"],
    "vconst": ["-DCURL_EXP_VCONST"],
    "vconst_prio1": ["-DCURL_EXP_VCONST", "-DCURL_EXP_PRIO=1"],
    # the other way round: the UNPAIRABLE runs (transcendental, packed) at raised priority
    "ps_t1": ["-DCURL_PRIO_TRANS=1"],
    "ps_tp1": ["-DCURL_PRIO_TRANS=1", "-DCURL_PRIO_PK=1"],
    "ps_t2p1": ["-DCURL_PRIO_TRANS=2", "-DCURL_PRIO_PK=1"],
    "ps_tp3": ["-DCURL_PRIO_TRANS=3", "-DCURL_PRIO_PK=3"],
    "ps_p1": ["-DCURL_PRIO_PK=1"],
    # packed-FP32 helpers as scalar loops (plain instructions can pair, packed ones cannot)
    "nopk": ["-DCURL_NO_PK"],
    "nopk_t1": ["-DCURL_NO_PK", "-DCURL_PRIO_TRANS=1"],
}


def path(name):
    return os.path.join(OUT_DIR, f"libcurlhip_{name}.so")


def build(name, force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    out = path(name)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in B._deps()):
        return out
    cmd = [B.HIPCC] + B.FLAGS + VARIANTS[name] + ["-o", out, B.SRC]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"hipcc failed on variant {name}")
    return out


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(VARIANTS)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(lambda n: build(n, "--force" in sys.argv), names):
            print(o)
