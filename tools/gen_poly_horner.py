"""Generate curl_amd/csrc/poly_horner.inc: multivariate Horner evaluators for the polynomial layers of the
reference (ChannelPolyLayer / Deg4MobilePolyLayer, model.py:206-415).

The reference orders the monomials of total degree <= D in V variables by `generate_powers(D, V)`
(model.py:222-246): graded, and inside one degree lexicographic with variable 0 most significant -- e.g. for
D=4, V=5: 1, v0, v1, v2, v3, v4, v0^2, v0 v1, ... , v4^4 (126 terms; the explicit list of model.py:346-397).
`powers(D, V)` below produces that order independently; tests/test_poly.py checks it against the reference's own
generator (run from its source text in the build container) and against the committed golden table.

Horner form: P = sum_k v0^k Q_k(v1..), Q_k of degree <= D-k, recursively; #FMAs = #coefficients - 1
(125 for D=4,V=5 against 121 shared multiplies + 126 FMAs per output for the monomial form).
The generated code is generic in the value type F (float on the host twin, a 2-pixel packed vector on gfx950)
and reads coefficient t of the current output through the macro CURL_POLY_C(t).
"""
import itertools
import os
import re
import sys


def powers(degree, nvars):
    out = []
    for total in range(degree + 1):
        tuples = [t for t in itertools.product(range(total + 1), repeat=nvars) if sum(t) == total]
        tuples.sort(reverse=True)  # lexicographic, variable 0 most significant, high powers first
        out.extend(tuples)
    return out


def gen(degree, nvars):
    table = powers(degree, nvars)
    index = {t: i for i, t in enumerate(table)}
    counter = [0]
    lines = []
    order = []  # consumption order: position -> reference coefficient index

    def coef(exps):
        order.append(index[tuple(exps)])
        return f"CURL_POLY_C({len(order) - 1})"

    def horner(var, max_deg, prefix):
        """returns a C expression (string) for the polynomial in variables var.. with total degree <= max_deg,
        emitting statements for intermediate accumulators; coefficients are numbered in the order they are used"""
        if var == nvars:
            return coef(prefix)
        if var == nvars - 1:
            # univariate tail: plain Horner chain in one accumulator, highest power first
            if max_deg == 0:
                return coef(prefix + [0])
            name = f"t{counter[0]}"
            counter[0] += 1
            lines.append(f"  F {name} = CURL_POLY_SPLAT({coef(prefix + [max_deg])});")
            for k in range(max_deg - 1, -1, -1):
                lines.append(f"  {name} = CURL_POLY_FMA({name}, v[{var}], {coef(prefix + [k])});")
            return name
        q_top = horner(var + 1, 0, prefix + [max_deg])
        name = f"t{counter[0]}"
        counter[0] += 1
        if q_top.startswith("CURL_POLY_C"):
            lines.append(f"  F {name} = CURL_POLY_SPLAT({q_top});")
        else:
            lines.append(f"  F {name} = {q_top};")
        for k in range(max_deg - 1, -1, -1):
            q = horner(var + 1, max_deg - k, prefix + [k])
            if q.startswith("CURL_POLY_C"):
                lines.append(f"  {name} = CURL_POLY_FMA({name}, v[{var}], {q});")
            else:
                lines.append(f"  {name} = CURL_POLY_FMAV({name}, v[{var}], {q});")
        return name

    result = horner(0, degree, [])
    assert sorted(order) == list(range(len(table)))
    n_fma = sum(1 for l in lines if "CURL_POLY_FMA" in l)
    # lock-step form: every statement runs over the NP pixel groups of the lane (NP independent chains), and a
    # scheduling fence bounds how far ahead coefficient reads can be hoisted.  Fences are placed where the number of
    # coefficients consumed so far is a multiple of 4 (>= FENCE_EVERY since the last one): every fenced region then
    # reads a 16-byte-aligned run of the consumption-order table, which the compiler merges into ds_read_b128.
    FENCE_EVERY = 12
    out_lines = []
    import re
    # A chain's initial value is a coefficient broadcast to every pixel (CURL_POLY_SPLAT) that the chain's FIRST fma
    # multiplies.  Materialising the broadcast costs a v_mov per odd-positioned coefficient on gfx950 (hipcc folds the
    # half-select into op_sel for an addend, not for a multiplicand): 160 per 4 pixels.  The generator therefore hands
    # the coefficient INDEX to that first fma (CURL_POLY_FMA_CC / CURL_POLY_FMAV_C: packed type = one v_pk_fma_f32 with
    # op_sel picking the half of the 8-byte pair the coefficient sits in; float = fmaf) and never forms the splat.
    splat = {}      # temp name -> consumption position of the coefficient it would have been a splat of
    read = set()    # consumption positions read so far
    last_fence = 0

    def cidx(expr):
        m = re.fullmatch(r"CURL_POLY_C\((\d+)\)", expr)
        return int(m.group(1)) if m else None

    for l in lines:
        m = re.match(r"  F (t\d+) = (.*);", l)
        if m:
            name, rhs = m.group(1), m.group(2)
            out_lines.append(f"  F {name}[NP];  // DECL")
            ms = re.fullmatch(r"CURL_POLY_SPLAT\((CURL_POLY_C\(\d+\))\)", rhs)
            if ms:
                splat[name] = cidx(ms.group(1))
            elif rhs in splat:
                splat[name] = splat[rhs]
            else:
                assert re.fullmatch(r"t\d+", rhs), rhs
                out_lines.append(f"  CURL_POLY_EACH {name}[p] = {rhs}[p];")
        else:
            m = re.match(r"  (t\d+) = (CURL_POLY_FMAV?)\((t\d+), (v\[\d+\]), (.*)\);", l)
            name, kind, a, var, add = m.groups()
            assert name == a
            var = var.replace("v[", "v[p][")
            qb = cidx(add)
            if qb is not None:
                read.add(qb)
            else:
                assert add not in splat, l
            if a in splat:
                qa = splat.pop(a)
                read.add(qa)
                if qb is not None:
                    out_lines.append(f"  CURL_POLY_EACH {name}[p] = CURL_POLY_FMA_CC({qa}, {var}, {qb});")
                else:
                    out_lines.append(f"  CURL_POLY_EACH {name}[p] = CURL_POLY_FMAV_C({qa}, {var}, {add}[p]);")
            elif qb is not None:
                out_lines.append(f"  CURL_POLY_EACH {name}[p] = CURL_POLY_FMA({name}[p], {var}, CURL_POLY_C({qb}));")
            else:
                out_lines.append(f"  CURL_POLY_EACH {name}[p] = CURL_POLY_FMAV({name}[p], {var}, {add}[p]);")
        # a fence only where the positions read so far are exactly a 16-byte-aligned prefix of the table
        n = len(read)
        if read == set(range(n)) and n % 4 == 0 and n - last_fence >= FENCE_EVERY:
            out_lines.append("  CURL_FENCE();")
            last_fence = n
    assert result not in splat, "degree-0 polynomial not supported here"
    assert read == set(range(len(table))), (sorted(set(range(len(table))) - read))
    assigned = set(re.findall(r"CURL_POLY_EACH (t\d+)\[p\] =", "\n".join(out_lines)))
    out_lines = [l.replace("  // DECL", "") for l in out_lines
                 if not l.endswith("// DECL") or re.match(r"  F (t\d+)\[", l).group(1) in assigned]
    body = "\n".join(out_lines)
    tab = ", ".join(str(i) for i in order)
    return table, f"""// degree {degree}, {nvars} variables: {len(table)} coefficients, {n_fma} FMAs per chain
// position in the order the Horner scheme consumes them -> coefficient index of the reference (generate_powers)
constexpr unsigned short kPolyOrder_d{degree}_v{nvars}[{len(table)}] = {{{tab}}};
// NP chains (pixel groups of one lane) advance in lock step.
// SEQ = true: c holds the coefficients in consumption order (the LDS copy: sequential, 16-byte aligned runs between
// fences -> ds_read_b128);
// SEQ = false: c is the reference's layout, indexed through the (compile-time) table.
template <class F, bool SEQ, int NP>
CURL_HD void poly_d{degree}_v{nvars}(F (&out)[NP], const F (&v)[NP][{nvars}], const float* c) {{
#define CURL_POLY_C(q) (SEQ ? PolyCoef<F>::seq(c, q) : PolyCoef<F>::ref(c[kPolyOrder_d{degree}_v{nvars}[q]]))
#define CURL_POLY_FMA_CC(qa, v, qb) (SEQ ? PolyCoef<F>::template fma_cc<qa, qb>(c, v) : CURL_POLY_FMA(CURL_POLY_C(qa), v, CURL_POLY_C(qb)))
#define CURL_POLY_FMAV_C(qa, v, t) (SEQ ? PolyCoef<F>::template fmav_c<qa>(c, v, t) : CURL_POLY_FMAV(CURL_POLY_C(qa), v, t))
#define CURL_POLY_EACH _Pragma("unroll") for (int p = 0; p < NP; ++p)
{body}
  CURL_POLY_EACH out[p] = {result}[p];
#undef CURL_POLY_EACH
#undef CURL_POLY_FMAV_C
#undef CURL_POLY_FMA_CC
#undef CURL_POLY_C
}}
"""


def gen_monomials(degree, nvars, chunk):
    """Straight-line code for the monomials themselves (the backward needs them: d out_o / d coef[o][t] = m_t):
    m[j] = prod_k pw[k][e_k] for the terms [c*chunk, (c+1)*chunk) of the reference order, from the power table
    pw[k][e] = v_k^e (e = 1..degree; exponent 0 factors are skipped)."""
    table = powers(degree, nvars)
    n_chunks = (len(table) + chunk - 1) // chunk
    out = [f"// monomials of degree <= {degree} in {nvars} variables, {n_chunks} chunk(s) of {chunk}",
           f"template <int C>\nCURL_HD void mono_d{degree}_v{nvars}(float (&m)[{chunk}], const float (&pw)[{nvars}][{degree + 1}]) {{"]
    for c in range(n_chunks):
        out.append(f"  if constexpr (C == {c}) {{")
        for j in range(chunk):
            t = c * chunk + j
            if t >= len(table):
                out.append(f"    m[{j}] = 0.0f;")
                continue
            factors = [f"pw[{k}][{e}]" for k, e in enumerate(table[t]) if e > 0]
            out.append(f"    m[{j}] = {' * '.join(factors) if factors else '1.0f'};")
        out.append("  }")
    out.append("}\n")
    return "\n".join(out)


def gen_collapse(degree, nvars):
    """Collapsing the LAST variable: P(v0..v_{n-2}, y) = sum_m m(v0..v_{n-2}) * (sum_j y^j c[(m, j)]).
    For a pixel row the last variable (row / height) is one value, so the inner sums are computed once per row
    and the per-pixel polynomial has nvars-1 variables: 70 coefficients instead of 126 for degree 4.
    Table: for every monomial of the (nvars-1)-variable reference order, the reference indices of
    (m, j = 0..degree), 0xFFFF where |m| + j > degree."""
    full = {t: i for i, t in enumerate(powers(degree, nvars))}
    small = powers(degree, nvars - 1)
    rows = []
    for m in small:
        rows.append("{" + ", ".join(str(full[m + (j,)]) if sum(m) + j <= degree else "0xFFFF" for j in range(degree + 1)) + "}")
    return (f"// y-collapse of the degree-{degree}, {nvars}-variable polynomial onto {nvars - 1} variables: row m = reference index of\n"
            f"// the {nvars - 1}-variable monomial; entry j = reference index ({nvars} variables) of m * v{nvars - 1}^j, 0xFFFF = none\n"
            f"constexpr unsigned short kPolyCollapse_d{degree}_v{nvars}[{len(small)}][{degree + 1}] = {{\n  " + ",\n  ".join(rows) + "};\n").replace("\\n", "\n")


def gen_fold(degree, nvars, order_small):
    """The same collapse indexed by CONSUMPTION position of the (nvars-1)-variable Horner scheme, 8 entries of 16 bits
    per position (16 bytes: one global_load_dwordx4 per folded coefficient on the device): entries 0..degree = reference
    indices (nvars variables) of m * v_last^j, 0xFFFF = none; entries degree+1.. = 0xFFFF padding."""
    full = {t: i for i, t in enumerate(powers(degree, nvars))}
    small = powers(degree, nvars - 1)
    rows = []
    for pos in range(len(small)):
        m = small[order_small[pos]]
        ent = [str(full[m + (j,)]) if sum(m) + j <= degree else "0xFFFF" for j in range(degree + 1)]
        ent += ["0xFFFF"] * (8 - len(ent))
        rows.append("{" + ", ".join(ent) + "}")
    return (f"// y-collapse by consumption position of the {nvars - 1}-variable scheme: kPolyFold[pos][j] = reference index of\n"
            f"// (monomial consumed at pos) * v{nvars - 1}^j, 0xFFFF = none; 16 bytes per position\n"
            f"struct alignas(16) PolyFoldRow {{ unsigned short j[8]; }};\n"
            f"constexpr PolyFoldRow kPolyFold_d{degree}_v{nvars}[{len(small)}] = {{\n  " + ",\n  ".join("{" + r + "}" for r in rows) + "};\n").replace("\\n", "\n")


def gen_foldx(degree, nvars, chunk):
    """Folding the variable BEFORE the last one out of the coefficient gradient (the backward's accumulation pass): a thread
    that walks down one image column sees one value x of variable nvars-2 (column / width), so it accumulates
    S[m'] = sum_rows g * m'(v0..v_{n-3}, y) over the monomials m' of the OTHER nvars-1 variables (70 instead of 126 for
    degree 4) and expands d coef[(m', x^j)] = x^j * S[m'] once at the end.  Emitted: per chunk of `chunk` monomials m'
    (reference order of nvars-1 variables) the pairs (m', j), cut into slices that fit the register file next to the
    accumulators: foldx_expand<C, S>(e, a, xp) writes e[i] = a[m'_i] * xp[j_i], kPolyFoldXIndex[C][S][i] is the reference
    index (nvars variables) of pair i (0xFFFF = padding)."""
    full = {t: i for i, t in enumerate(powers(degree, nvars))}
    small = powers(degree, nvars - 1)
    n_chunks = (len(small) + chunk - 1) // chunk
    per_chunk = []
    for c in range(n_chunks):
        pairs = []
        for jl in range(chunk):
            q = c * chunk + jl
            if q >= len(small):
                continue
            m = small[q]
            for j in range(degree - sum(m) + 1):
                pairs.append((jl, j, full[m[:-1] + (j,) + m[-1:]]))
        per_chunk.append(pairs)
    max_slice = 48
    geo = []
    for pairs in per_chunk:
        n_sl = (len(pairs) + max_slice - 1) // max_slice
        geo.append((len(pairs), n_sl, (len(pairs) + n_sl - 1) // n_sl))
    smax, lmax = max(g[1] for g in geo), max(g[2] for g in geo)
    out = [f"// x-fold of the coefficient gradient (degree {degree}, {nvars} variables): chunks of {chunk} monomials of the other {nvars - 1}\n"
           f"// variables, each expanded by the powers of variable {nvars - 2}; see tools/gen_poly_horner.py:gen_foldx",
           "template <int C>\nstruct PolyFoldX;"]
    for c, (n, n_sl, ln) in enumerate(geo):
        out.append(f"template <>\nstruct PolyFoldX<{c}> {{\n  static constexpr int kPairs = {n}, kSlices = {n_sl}, kSlice = {ln};\n}};")
    out.append(f"constexpr int kPolyFoldXChunks = {n_chunks}, kPolyFoldXMaxSlices = {smax}, kPolyFoldXMaxSlice = {lmax};")
    out.append(f"template <int C, int S>\nCURL_HD void foldx_expand(float (&e)[PolyFoldX<C>::kSlice], const float (&a)[{chunk}], const float (&xp)[{degree + 1}]) {{")
    rows = []
    for c, (n, n_sl, ln) in enumerate(geo):
        crow = []
        for sl in range(smax):
            part = per_chunk[c][sl * ln:(sl + 1) * ln] if sl < n_sl else []
            if sl < n_sl:
                out.append(f"  if constexpr (C == {c} && S == {sl}) {{")
                for i in range(ln):
                    if i < len(part):
                        jl, j, _ = part[i]
                        out.append(f"    e[{i}] = a[{jl}]" + (f" * xp[{j}];" if j else ";"))
                    else:
                        out.append(f"    e[{i}] = 0.0f;")
                out.append("  }")
            ent = [str(t) for _, _, t in part] + ["0xFFFF"] * (lmax - len(part))
            crow.append("{" + ", ".join(ent) + "}")
        rows.append("{" + ",\n   ".join(crow) + "}")
    out.append("}")
    out.append(f"constexpr unsigned short kPolyFoldXIndex[{n_chunks}][{smax}][{lmax}] = {{\n  " + ",\n  ".join(rows) + "};\n")
    return "\n".join(out).replace("\\n", "\n")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "curl_amd", "csrc", "poly_horner.inc")
    parts = ["// GENERATED by tools/gen_poly_horner.py -- do not edit.\n"
             "// Multivariate Horner evaluators in the coefficient order of the reference's generate_powers\n"
             "// (model.py:222-246).  F = float (host twin) or a packed 2-pixel vector (gfx950: v_pk_fma_f32).\n"
             "// CURL_POLY_C(q): coefficient q as an F; CURL_POLY_FMA / FMAV(a, v, q): a*v + q; CURL_POLY_FMA_CC(qa, v, qb) =\n"
             "// coef[qa]*v + coef[qb] and CURL_POLY_FMAV_C(qa, v, t) = coef[qa]*v + t: the first fma of a chain, taking its\n"
             "// initial coefficient by INDEX (see tools/gen_poly_horner.py).\n"]
    orders = {}
    for degree, nvars in ((4, 5), (4, 3), (4, 4)):
        table, code = gen(degree, nvars)
        parts.append(code)
        orders[nvars] = [int(x) for x in re.search(r"kPolyOrder_d4_v%d\[\d+\] = \{([^}]*)\}" % nvars, code).group(1).split(",")]
    parts.append(gen_collapse(4, 5))
    parts.append(gen_fold(4, 5, orders[4]))
    parts.append(gen_monomials(4, 5, 42))
    parts.append(gen_monomials(4, 3, 35))
    parts.append(gen_monomials(4, 4, 35))
    parts.append(gen_foldx(4, 5, 35))
    open(out, "w").write("\n".join(parts))
    print("wrote", out)


if __name__ == "__main__":
    main()
