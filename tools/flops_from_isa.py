"""Static FLOP and VALU-issue counts of a kernel's hot basic block from hipcc's assembly (the numbers bench.py's
`valu` rooflines use).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-math-errno -fno-slp-vectorize -S --cuda-device-only -o /tmp/curl.s curl_amd/csrc/curl_kernels.hip
    python tools/flops_from_isa.py /tmp/curl.s 'stream_kernelI7OpLayerLi4ELi1ELi1ELb1ELi0E' [px_per_lane=4] [--all]
(--all: sum every basic block instead of the largest one -- for straight-line kernels such as layer_bwd_kernel, whose
 reduction epilogue is a block of its own)

FLOPs per lane: v_pk_fma_f32 4, v_fma/v_fmac/v_fmamk/v_fmaak 2, v_pk_mul/v_pk_add 2, v_mul/v_add/v_sub/v_min/v_max/
v_med3/v_min3/v_max3 (f32) 1, v_exp/v_log/v_rcp/v_rsq/v_sqrt 1; integer, bit, move, convert instructions 0.
Issue cycles per wave-instruction per SIMD (tools/ubench/valu_rate.hip, issue_mix.hip): packed / min / max / med3 /
cmp / cvt 4, transcendental 8, every other VALU 4 alone or 2 when it pairs with another wave's.
"""
import re
import sys
from collections import Counter

FLOPS = [(r"v_pk_fma_f32", 4), (r"v_(fma|fmac|fmamk|fmaak|mad)_f32", 2), (r"v_pk_(mul|add)_f32", 2),
         (r"v_(mul|add|sub|subrev|min|max|med3|min3|max3)_f32", 1), (r"v_(exp|log|rcp|rsq|sqrt)_f32", 1)]
HALF = r"v_pk_|v_(min|max|med3|min3|max3)_f32|v_cmp|v_cndmask|v_cvt|v_bfi|v_floor|v_fract|v_ldexp|v_frexp|v_mad_u32"
QUART = r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32"


def kernel_blocks(path, frag):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and frag in l and l.rstrip().endswith(
        tuple([":"])) or (l.startswith("_Z") and frag in l and ": " in l and l.split(":")[0].find(frag) >= 0))
    blocks, cur = [], []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = []
            continue
        m = re.match(r"^([a-z_0-9]+)\s", t + " ")
        if m and m.group(1).startswith(("v_", "s_", "ds_", "global_", "buffer_")):
            cur.append(m.group(1))
    blocks.append(cur)
    return blocks


def flop_per_lane(path, frag, all_blocks=True):
    """FLOPs per lane of kernel `frag` in the assembly at `path` (all blocks, or the largest one): what bench.py's flop_px
    figures are -- tests/test_build_resources.py holds them to the build."""
    blocks = kernel_blocks(path, frag)
    hot = [i for b in blocks for i in b] if all_blocks else max(blocks, key=lambda b: sum(1 for i in b if i.startswith("v_")))
    total = 0
    for i in hot:
        if not i.startswith("v_"):
            continue
        for pat, f in FLOPS:
            if re.match(pat, i):
                total += f
                break
    return total


def main():
    path, frag = sys.argv[1], sys.argv[2]
    ppl = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else 4
    blocks = kernel_blocks(path, frag)
    if "--all" in sys.argv:  # straight-line kernels (no loop): every block runs at most once per wave
        hot = [i for b in blocks for i in b]
    else:
        hot = max(blocks, key=lambda b: sum(1 for i in b if i.startswith("v_")))
    valu = [i for i in hot if i.startswith("v_")]
    flops = 0
    for i in valu:
        for pat, f in FLOPS:
            if re.match(pat, i):
                flops += f
                break
    nq = sum(1 for i in valu if re.match(QUART, i))
    nh = sum(1 for i in valu if re.match(HALF, i) and not re.match(QUART, i))
    nf = len(valu) - nq - nh
    allv = sum(1 for b in blocks for i in b if i.startswith("v_"))
    print(f"kernel {frag}: {len(blocks)} blocks, {allv} VALU instructions in all; hot block {len(valu)} VALU "
          f"({nf} plain, {nh} half-rate, {nq} transcendental), {sum(1 for i in hot if i.startswith('ds_'))} LDS, "
          f"{sum(1 for i in hot if i.startswith('s_'))} scalar")
    print(f"  FLOP per lane {flops} = {flops / ppl:.1f} FLOP/px at {ppl} px per lane")
    print(f"  issue cycles per wave: unpaired {4 * nf + 4 * nh + 8 * nq}, fully paired {2 * nf + 4 * nh + 8 * nq}")
    print("  " + ", ".join(f"{k} {v}" for k, v in Counter(valu).most_common(14)))


if __name__ == "__main__":
    main()
