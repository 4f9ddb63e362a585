"""VALU instructions per basic block of one kernel in hipcc's assembly (see tools/flops_from_isa.py for the -S command):
    python tools/isa_blocks.py /tmp/curl.s 'stream_kernelI7OpLayerLi4ELi1ELi1ELb1ELi0E'
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and frag in l.split(":")[0] and ":" in l)
blocks, cur, names = [], [], ["entry"]
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append(cur)
        cur = []
        names.append(m.group(1))
        continue
    m = re.match(r"^([a-z_0-9]+)\s", t + " ")
    if m and m.group(1).startswith(("v_", "s_", "ds_", "global_", "buffer_")):
        cur.append(m.group(1))
blocks.append(cur)
for n, b in zip(names, blocks):
    v = [i for i in b if i.startswith("v_")]
    tr = sum(1 for i in v if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", i))
    print(f"{n:12s} {len(v):5d} VALU ({tr} transcendental) {len(b):5d} instructions")
