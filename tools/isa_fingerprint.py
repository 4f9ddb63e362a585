"""Per-kernel fingerprint of the product translation unit's gfx950 assembly (cross-compiles without a GPU, ~60 s).

    python tools/isa_fingerprint.py out.json            # write {kernel: {sha, n_inst, n_valu, vgprs, sgprs, scratch, lds}}
    python tools/isa_fingerprint.py out.json --diff base.json   # ... and list every kernel whose code changed

Used to show that a source clean-up (experiment switches taken out of curl_amd/csrc) leaves the machine code of every
product kernel as it was, and by tests/test_build_resources.py to pin the instruction counts of the hot kernels.
The hash covers the instruction stream only (mnemonics + operands; labels renumbered, comments and directives dropped)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def assemble(extra_flags=()):
    from curl_amd import build as B
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "curl.s")
        subprocess.check_call([hipcc] + flags + list(extra_flags) + ["-S", "--cuda-device-only", "-o", out, B.SRC],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels_of(asm):
    """{mangled kernel name: fingerprint dict} for every .amdhsa_kernel of the assembly text."""
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, flags=re.S):
        body = m.group(2)

        def num(key):
            r = re.search(key + r" (\d+)", body)
            return int(r.group(1)) if r else None
        meta[m.group(1)] = {"vgprs": num("next_free_vgpr"), "sgprs": num("next_free_sgpr"),
                            "scratch": num("private_segment_fixed_size"), "lds": num("group_segment_fixed_size")}
    out = {}
    for name, md in meta.items():
        m = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, flags=re.S | re.M)
        if not m:
            continue
        labels, lines = {}, []
        for ln in m.group(1).splitlines():
            ln = ln.split(";")[0].strip()
            if not ln or ln.startswith("."):
                if re.match(r"\.LBB\d+_\d+:", ln):
                    labels[ln[:-1]] = f"L{len(labels)}"
                    lines.append(ln)  # renamed below
                continue
            lines.append(ln)
        text = "\n".join(lines)
        for old, new in labels.items():
            text = re.sub(re.escape(old) + r"\b", new, text)
        inst = [l for l in text.splitlines() if not l.endswith(":")]
        md = dict(md)
        md.update({"sha": hashlib.sha256(text.encode()).hexdigest()[:16], "n_inst": len(inst),
                   "n_valu": sum(1 for l in inst if l.startswith("v_")),
                   "n_trans": sum(1 for l in inst if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", l))})
        out[name] = md
    return out


def main():
    out_path = sys.argv[1]
    fp = kernels_of(assemble())
    json.dump(fp, open(out_path, "w"), indent=0, sort_keys=True)
    print(f"{len(fp)} kernels -> {out_path}")
    if "--diff" in sys.argv:
        base = json.load(open(sys.argv[sys.argv.index("--diff") + 1]))
        changed = [k for k in sorted(set(fp) | set(base)) if fp.get(k, {}).get("sha") != base.get(k, {}).get("sha")]
        for k in changed:
            a, b = base.get(k), fp.get(k)
            print("CHANGED" if a and b else ("GONE   " if a else "NEW    "), k,
                  "" if not (a and b) else f"n_inst {a['n_inst']} -> {b['n_inst']}, vgprs {a['vgprs']} -> {b['vgprs']}")
        print(f"{len(changed)} of {len(set(fp) | set(base))} kernels differ")
        return 1 if changed else 0
    return 0


if __name__ == "__main__":
    sys.exit(main())
