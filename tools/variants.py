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
R1 = ["-DCURL_USE_PK", "-DCURL_PRIO_TRANS=0", "-DCURL_SELECT_BITWISE"]  # the round-1 code: packed helpers, every wave at one priority

VARIANTS = {
    "base": [],  # = the product library's switches
    # in-kernel s_memtime / s_memrealtime stamps per wave (tools/stamp.py)
    "stamp": ["-DCURL_DIAG_STAMP"],
    "r1_stamp": R1 + ["-DCURL_DIAG_STAMP"],
    # loads issued, arithmetic on synthetic values: separates "waiting for data" from "sharing the chip with traffic"
    "nodep": ["-DCURL_DIAG_NO_DEP"],
    # scheduling fences around the transcendental runs removed
    "nofence": ["-DCURL_NO_FENCE"],
    # ---- dual-issue record (curl_math.h) ----
    "r1": R1,
    # plain code at RAISED priority (the first guess; makes it worse)
    "r1_fast1": ["-DCURL_USE_PK", "-DCURL_EXP_PRIO=1"],
    # constants / coefficients of the scalar FMAs in VGPRs (an SGPR-operand instruction pairs with a plain one anyway)
    "r1_vconst": R1 + ["-DCURL_EXP_VCONST"],
    # the unpairable runs at raised priority: transcendental only, packed only, both
    "pk_t1": ["-DCURL_USE_PK", "-DCURL_PRIO_TRANS=1"],
    "pk_p1": ["-DCURL_USE_PK", "-DCURL_PRIO_TRANS=0", "-DCURL_PRIO_PK=1"],
    "pk_tp1": ["-DCURL_USE_PK", "-DCURL_PRIO_TRANS=1", "-DCURL_PRIO_PK=1"],
    "pk_tp3": ["-DCURL_USE_PK", "-DCURL_PRIO_TRANS=3", "-DCURL_PRIO_PK=3"],
    # scalar helpers: without priorities, with transcendental runs at 1 (= base), at 3
    "nopk_t0": ["-DCURL_PRIO_TRANS=0"],
    "nopk_t3": ["-DCURL_PRIO_TRANS=3"],
    # polynomial model: its packed Horner code at raised priority too / the converters' helpers packed again
    "slp": ["-fslp-vectorize"],  # WITH hipcc's SLP vectoriser (the product build disables it: it packs independent scalar chains into v_pk_* with shuffles)
    "sel_vthr": ["-DCURL_SELECT_VTHR"],  # thresholds of the selects in VGPRs (the compare then reads no SGPR)
    "sel_bitwise": ["-DCURL_SELECT_BITWISE"],
    "hue_bitwise": ["-DCURL_HUE_BITWISE"],  # only the hue terms' [c == max] factors in the sign-bit form  # threshold selects as sub / ashr / bitop3 (default: v_cmp + v_cndmask_e64)
    "poly_splat": ["-DCURL_POLY_SPLAT_FIRST"],  # the chains' first fma from a compiler-built {c, c} pair (v_mov per odd c)
    "poly_stage_r1": ["-DCURL_POLY_STAGE_GLOBAL"],  # row folds read global memory; pixel loads after the staging barrier
    "bwd_plain": ["-DCURL_TRI_BWD_PLAIN"],  # spatial polynomial backward: 126 monomials over flat tiles (round 1) instead of column strips
    "loss_bwd_vec1": ["-DCURL_LOSS_BWD_VEC1"],  # CURLLoss terms backward at one pixel per lane (84 VGPRs) instead of four (186)
    "bwd_s32": ["-DCURL_TRI_STRIP_STEPS_MAX=32"],  # rows per thread of the column strips capped at 32 (default 64)
    "r2_hsv": ["-DCURL_R2_HSV"],  # round-2 HSV code: hsv2rgb as two saturated ramps + two fmas per channel (default: one trapezoid), s masked by df != 0
    "bwd_interleave": ["-DCURL_BWD_INTERLEAVE"],  # layer backward: the lane's four pixels left to the compiler to interleave
    "bwd_w4": ["-DCURL_BWD_WAVES=4"],  # layer backward held to 128 VGPRs (four waves per SIMD; 72 bytes of scratch per lane)
    "nolazy": ["-DCURL_NO_LAZY_SELECT"],  # the Lab converters' threshold selects always executed (default: skipped by waves that need none)
    "nomem_inline": ["-DCURL_NOMEM_INLINE"],  # the no-memory diagnostics branch left to the compiler (ten v_mov splats on the product path)
    "mask_last": ["-DCURL_MASK_LOAD_LAST"],  # the mask's load behind the three plane loads (where the compiler put it once the bytes were one dword)
    "addr64": ["-DCURL_ADDR64"],  # streaming kernels: pointer + 64-bit lane offset (default: SGPR plane base + 32-bit byte offset)
    "aux_res2": ["-DCURL_RES_PSNR=2", "-DCURL_RES_EGRESS=2", "-DCURL_RES_INGRESS=2", "-DCURL_RES_LOSS=2", "-DCURL_RES_LOSS_BWD=2"],  # PSNR / byte edges / loss terms at 2 workgroups per CU (default: edges 4, the others uncapped)
    "aux_res4": ["-DCURL_RES_PSNR=4", "-DCURL_RES_EGRESS=4", "-DCURL_RES_INGRESS=4", "-DCURL_RES_LOSS=4", "-DCURL_RES_LOSS_BWD=4"],  # PSNR / byte edges / loss terms at 4 workgroups per CU (default: edges 4, the others uncapped)
    "aux_res6": ["-DCURL_RES_PSNR=6", "-DCURL_RES_EGRESS=6", "-DCURL_RES_INGRESS=6", "-DCURL_RES_LOSS=6", "-DCURL_RES_LOSS_BWD=6"],  # PSNR / byte edges / loss terms at 6 workgroups per CU (default: edges 4, the others uncapped)
    "aux_res3": ["-DCURL_RES_PSNR=3", "-DCURL_RES_EGRESS=3", "-DCURL_RES_INGRESS=3", "-DCURL_RES_LOSS=3", "-DCURL_RES_LOSS_BWD=3"],
    "aux_res5": ["-DCURL_RES_PSNR=5", "-DCURL_RES_EGRESS=5", "-DCURL_RES_INGRESS=5", "-DCURL_RES_LOSS=5", "-DCURL_RES_LOSS_BWD=5"],
    "grid_image_major": ["-DCURL_GRID_IMAGE_MAJOR"],  # stream kernels: grid = (images, tiles per image): consecutive workgroups walk different images
    "mask_sample": ["-DCURL_MASK_SAMPLE"],  # the knot-prep kernel samples the mask and the main kernel asks the workspace whether to test its mask first (exp33: +0.6 ... +1.6 % on all-ones masks)
    "aux_res0": ["-DCURL_RES_EGRESS=0", "-DCURL_RES_INGRESS=0", "-DCURL_RES_PSNR=0"],  # the byte edges uncapped (before exp27g)
    "pow24_direct": ["-DCURL_POW24_DIRECT"],  # fused stages: u^2.4 as 2^(2.4 log2 u) (default: u*u * 2^(0.4 log2 u)); -1 % and one test pixel over 1e-5
    "poly1": ["-DCURL_PRIO_POLY=1"],
    "poly1_t2": ["-DCURL_PRIO_POLY=1", "-DCURL_PRIO_TRANS=2"],
    "poly1_pk": ["-DCURL_PRIO_POLY=1", "-DCURL_USE_PK", "-DCURL_PRIO_PK=1"],
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
