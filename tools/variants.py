"""Experiment builds of libcurlhip.so (never the product library): each variant is the same translation unit
compiled with extra -D switches into curl_amd/lib/variants/libcurlhip_<name>.so.  Built in the CPU container
(hipcc cross-compiles gfx950) so the .so files travel to the GPU box with the snapshot.

    python tools/variants.py                 # build all
    python tools/variants.py slp             # build some
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curl_amd import build as B  # noqa: E402

OUT_DIR = os.path.join(ROOT, "curl_amd", "lib", "variants")

# name -> extra compiler switches.  Round 4 took the -D experiment switches of rounds 1-3 out of curl_amd/csrc (VERDICT r3
# item 6): the product sources build ONE configuration.  To rebuild a historical variant, apply
# tools/experiments/patches/r01-r03_experiment_switches.patch to a scratch checkout of the commit named in
# tools/experiments/README.md and pass its -D switch (the table there lists every switch, what it did and its log).
# What is left here are builds that need no source switch; new A/B candidates are built from a scratch copy of the tree
# (tools/ab.py takes any two .so paths).
VARIANTS = {
    "base": [],  # = the product library's switches
    "slp": ["-fslp-vectorize"],  # WITH hipcc's SLP vectoriser (the product build disables it: it packs independent scalar chains into v_pk_* with shuffles)
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
