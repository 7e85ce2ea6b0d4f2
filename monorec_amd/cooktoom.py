"""Cook-Toom (1-D Winograd) matrices F(m, r) in exact rational arithmetic, and the generator of `csrc/cooktoom_1d.h`.

F(m, r) computes m outputs of an r-tap correlation y_k = sum_j g_j d_{k+j} from n = m + r - 1 inputs with n multiplies:
    y = A^T [ (G g) o (B^T d) ]
The k x 1 / 1 x k stride-1 layers of layers.ConvReLU2 (reference model/layers.py:289-314, model/monorec/monorec_model.py:485-513) run this
way on the matrix cores (csrc/conv1d_wino.hip): the product over input channels of (G g) and (B^T d) is MFMA work, the transforms are a
few VALU instructions per lane.  F(2, 3) is hand written there; the larger forms - F(4, 3), F(2, 7), F(4, 7) - take their straight-line
transform code and the G tables of the weight packer from the header this module writes (`python tools/gen_cooktoom.py`), so kernel,
packer and tests share ONE derivation.  Interpolation points: 0, +-1, +-2, +-1/2, +-4 and infinity; every entry of A^T and B^T is a
dyadic rational (exact in fp32), G is rounded once per weight in double.
"""
from fractions import Fraction

# finite interpolation points by tile size n = m + r - 1
POINTS = {4: [0, 1, -1], 6: [0, 1, -1, 2, -2], 8: [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2)],
          10: [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2), 4, -4],
          # odd tile sizes: the phase filters of the stride-2 layers (polyphase_stride2 below): F(2,2), F(2,4) / F(4,2), F(4,4)
          3: [0, -1], 5: [0, 1, -1, 2], 7: [0, 1, -1, 2, -2, Fraction(1, 2)]}
FORMS = ((4, 3), (2, 7), (4, 7), (4, 4))  # what cooktoom_1d.h carries (F(2, 3) is written out in conv1d_wino.hip); F(4, 4): the 7-tap stride-2
                                          # layers as ONE 4-tap stride-1 correlation over [even rows | odd rows] (see stride2_as_stride1)


def cook_toom(m, r, points=None):
    """(A^T m x n, G n x r, B^T n x n) as Fractions; the last interpolation point is infinity.  A^T and G follow from the points,
    B^T is SOLVED from the bilinear identity sum_i A^T[k][i] G[i][j] B^T[i][l] = [l == k + j], so the triple is exact by construction."""
    n = m + r - 1
    pts = [Fraction(p) for p in (POINTS[n] if points is None else points)]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1
    at = [[(pts[i] ** k if i < n - 1 else Fraction(int(k == m - 1))) for i in range(n)] for k in range(m)]
    g = []
    for i in range(n - 1):
        norm = Fraction(1)
        for j in range(n - 1):
            if j != i:
                norm *= pts[i] - pts[j]
        g.append([pts[i] ** j / norm for j in range(r)])
    g.append([Fraction(0)] * (r - 1) + [Fraction(1)])
    rows = [[at[k][i] * g[i][j] for i in range(n)] for k in range(m) for j in range(r)]
    bt = [[Fraction(0)] * n for _ in range(n)]
    for col in range(n):
        rhs = [Fraction(int(col == k + j)) for k in range(m) for j in range(r)]
        sol = _solve(rows, rhs, n)
        for i in range(n):
            bt[i][col] = sol[i]
    return at, g, bt


def _solve(rows, rhs, n):
    a = [list(r_) + [b] for r_, b in zip(rows, rhs)]
    row = 0
    for col in range(n):
        p = next((i for i in range(row, len(a)) if a[i][col] != 0), None)
        if p is None:
            raise ValueError("under-determined")
        a[row], a[p] = a[p], a[row]
        inv = 1 / a[row][col]
        a[row] = [v * inv for v in a[row]]
        for i in range(len(a)):
            if i != row and a[i][col] != 0:
                f = a[i][col]
                a[i] = [vi - f * vr for vi, vr in zip(a[i], a[row])]
        row += 1
    if any(any(v != 0 for v in r_) for r_ in a[row:]):
        raise ValueError("inconsistent")
    return [a[i][n] for i in range(n)]


def polyphase_stride2(r, pad_lo):
    """An r-tap correlation with stride 2, y_i = sum_k g_k d_{2 i + k - pad_lo} (layers.ConvReLU2 with stride 2, reference model/layers.py:289-314
    and monorec_model.py:490-499; `pad_lo` = the low-side padding of PadSameConv2d), as two STRIDE-1 correlations on the even and the odd
    samples of the padded input p_j = d_{j - pad_lo}:  y_i = sum_t g_{2t} e_{i + t} + sum_t g_{2t+1} o_{i + t}  with e_j = p_{2j}, o_j = p_{2j+1}.
    Returns (even tap indices, odd tap indices): ceil(r/2) and floor(r/2) taps, each a Cook-Toom candidate F(m, .) - 7 taps: F(2,4) + F(2,3) =
    9 of 14 multiplies per 2 outputs, F(4,4) + F(4,3) = 13 of 28 per 4; 5 taps: 7 of 10 / 11 of 20.  (`pad_lo` only shifts which INPUT samples are
    even: the split of the taps is by tap index.)"""
    del pad_lo
    return list(range(0, r, 2)), list(range(1, r, 2))


def stride2_as_stride1(r, n):
    """An r-tap correlation with stride 2 over n input samples (n even, PadSameConv2d padding: layers.ConvReLU2 with stride 2, reference
    model/layers.py:241-252,289-314) as ONE stride-1 correlation with r2 = ceil(r / 2) taps over the concatenation [e | o] of the even and the
    odd input samples (e_j = d_{2j}, o_j = d_{2j+1}): y_i = sum_t ge_t e_{i + t - pad} + sum_t go_t o_{i + t - pad}.  Returns
    (r2, pad, even tap indices, odd tap indices): ge_t = g[even[t]], go_t = g[odd[t]], an index of None = a zero tap.  With the reference's
    low-side padding pad_lo = floor((2 (n/2 - 1) + r - n) / 2) = (r - 2) // 2 (n even) both phases share the SAME offset `pad`, so the pair is a
    plain 'same'-style filter over twice the channels - which is what lets the stride-1 Cook-Toom kernels run it on row-strided / column-
    deinterleaved views of the input: 7 taps -> 4 taps (F(4,4): 7 multiplies per 4 outputs and channel pair = 3.5 per output instead of 7),
    5 taps -> 3 taps (F(4,3): 3 instead of 5).  The kernels (csrc/conv1d_wino.hip) and the oracle-side numerics study use this one derivation."""
    assert n % 2 == 0 and r >= 2
    pad_lo = (2 * (n // 2 - 1) + r - n) // 2
    r2 = (r + 1) // 2
    # tap k reads d_{2 i + k - pad_lo}: even sample index when (k - pad_lo) is even -> e_{i + (k - pad_lo) / 2}, else o_{i + (k - pad_lo - 1) / 2}
    ev = {(k - pad_lo) // 2: k for k in range(r) if (k - pad_lo) % 2 == 0}
    od = {(k - pad_lo - 1) // 2: k for k in range(r) if (k - pad_lo) % 2 != 0}
    lo = min(min(ev), min(od))
    hi = max(max(ev), max(od))
    assert hi - lo + 1 == r2, (r, lo, hi)
    return r2, -lo, [ev.get(lo + t) for t in range(r2)], [od.get(lo + t) for t in range(r2)]


def correlate_stride2_polyphase(d, g, m, pad_lo):
    """Exact (Fraction) evaluation of the stride-2 correlation of polyphase_stride2 through F(m, r_even) + F(m, r_odd): the reference the kernel
    and the numerics study are held against.  d: input samples, g: taps; returns the outputs the zero-padded stride-2 correlation defines for
    i = 0 .. ceil(len(d) / 2) - 1."""
    r = len(g)
    n_out = -(-len(d) // 2)
    ev, od = polyphase_stride2(r, pad_lo)
    span = 2 * (n_out + m) + r + 2
    p = [Fraction(0)] * span
    for j, v in enumerate(d):
        p[j + pad_lo] = Fraction(v)
    phases = ([p[j] for j in range(0, span, 2)], [Fraction(g[k]) for k in ev]), ([p[j] for j in range(1, span, 2)], [Fraction(g[k]) for k in od])
    y = [Fraction(0)] * (-(-n_out // m) * m)
    for samples, taps in phases:
        rr = len(taps)
        if rr == 0:
            continue
        if rr == 1:                                   # one tap: a plain product per output
            for i in range(len(y)):
                y[i] += taps[0] * samples[i]
            continue
        at, gm, bt = cook_toom(m, rr)
        n = m + rr - 1
        u = [sum(gm[i][j] * taps[j] for j in range(rr)) for i in range(n)]
        for t0 in range(0, len(y), m):
            v = [sum(bt[i][c] * samples[t0 + c] for c in range(n)) for i in range(n)]
            for k in range(m):
                y[t0 + k] += sum(at[k][i] * u[i] * v[i] for i in range(n))
    return y[:n_out]


def identity_holds(m, r, at, g, bt):
    n = m + r - 1
    return all(sum(at[k][i] * g[i][j] * bt[i][col] for i in range(n)) == int(col == k + j)
               for k in range(m) for j in range(r) for col in range(n))


# ------------------------------------------------------------------------------------------------------------------ code generation
def _lit(fr):
    """fp32 literal of a dyadic rational (exact)."""
    fr = Fraction(fr)
    assert fr.denominator & (fr.denominator - 1) == 0, fr
    return repr(float(fr)) + "f"


def _chain(terms, src):
    """sum_i c_i src[i] as one left-to-right fmaf chain (terms = [(index, coefficient)], zeros dropped); +-1 as add / subtract."""
    terms = [(i, Fraction(c)) for i, c in terms if c != 0]
    if not terms:
        return "0.f"
    # a term with coefficient +-1 opens the chain (every other term is then ONE fmaf; opening with c * x costs a multiply of its own:
    # F(4,3)'s input transform 17 -> 13 instructions)
    lead = next((k for k, (_, c) in enumerate(terms) if c == 1), next((k for k, (_, c) in enumerate(terms) if c == -1), 0))
    terms = [terms[lead]] + terms[:lead] + terms[lead + 1:]
    i0, c0 = terms[0]
    e = f"{src}[{i0}]" if c0 == 1 else (f"-{src}[{i0}]" if c0 == -1 else f"{_lit(c0)} * {src}[{i0}]")
    for i, c in terms[1:]:
        if c == 1:
            e = f"({e} + {src}[{i}])"
        elif c == -1:
            e = f"({e} - {src}[{i}])"
        else:
            e = f"fmaf({_lit(c)}, {src}[{i}], {e})"
    return e


def _input_transform(m, r, bt):
    """v = B^T d.  Rows come in pairs whose odd-index coefficients differ in sign only (points +-p): v_p = E + O, v_p+1 = E - O."""
    n = m + r - 1
    out, p = [], 0
    while p < n:
        nxt = p + 1
        paired = nxt < n and all(bt[nxt][i] == (bt[p][i] if i % 2 == 0 else -bt[p][i]) for i in range(n)) and any(bt[p][i] != 0 for i in range(1, n, 2))
        if paired:
            out.append(f"    {{ const float e = {_chain([(i, bt[p][i]) for i in range(0, n, 2)], 'd')};")
            out.append(f"      const float o = {_chain([(i, bt[p][i]) for i in range(1, n, 2)], 'd')};")
            out.append(f"      v[{p}] = e + o; v[{nxt}] = e - o; }}")
            p += 2
        else:
            out.append(f"    v[{p}] = {_chain([(i, bt[p][i]) for i in range(n)], 'd')};")
            p += 1
    return out


def _output_transform(m, r, at):
    """y = A^T M: columns 1, 2 / 3, 4 / ... belong to points +-p: their sum feeds the even rows, their difference the odd ones."""
    n = m + r - 1
    out = []
    npair = (n - 2) // 2
    for q in range(npair):
        out.append(f"    const float s{q} = mm[{1 + 2 * q}] + mm[{2 + 2 * q}], t{q} = mm[{1 + 2 * q}] - mm[{2 + 2 * q}];")
    for k in range(m):
        e = "mm[0]" if at[k][0] == 1 else None
        assert at[k][0] in (0, 1)
        for q in range(npair):
            c = at[k][1 + 2 * q]
            assert at[k][2 + 2 * q] == (c if k % 2 == 0 else -c)
            name = f"s{q}" if k % 2 == 0 else f"t{q}"
            if c == 0:
                continue
            if e is None:
                e = name if c == 1 else f"{_lit(c)} * {name}"
            else:
                e = f"({e} + {name})" if c == 1 else f"fmaf({_lit(c)}, {name}, {e})"
        if n % 2 == 1:                                   # an unpaired finite point (odd tile sizes: F(4,4), F(2,4)): its own column, n - 2
            c = at[k][n - 2]
            if c != 0:
                if e is None:
                    e = f"mm[{n - 2}]" if c == 1 else f"{_lit(c)} * mm[{n - 2}]"
                else:
                    e = f"({e} + mm[{n - 2}])" if c == 1 else f"fmaf({_lit(c)}, mm[{n - 2}], {e})"
        c = at[k][n - 1]
        if c != 0:
            assert c == 1
            e = f"({e} + mm[{n - 1}])"
        out.append(f"    y[{k}] = {e};")
    return out


def generate_header():
    lines = ["// GENERATED by monorec_amd/cooktoom.py (python tools/gen_cooktoom.py) - do not edit.",
             "// Cook-Toom forms F(m, r) for the k x 1 / 1 x k stride-1 convolutions of layers.ConvReLU2 (reference model/layers.py:289-314):",
             "// y = A^T [ sum_cin (G g) o (B^T d) ]; interpolation points 0, +-1, +-2, +-1/2, +-4, infinity.  A^T / B^T are dyadic rationals",
             "// (exact fp32 literals); G is used by the host packer in double (each entry = one correctly rounded division).",
             "// tests/test_capi_and_host.py checks this file against its generator and the bilinear identity in exact arithmetic.",
             "#pragma once", ""]
    for m, r in FORMS:
        at, g, bt = cook_toom(m, r)
        assert identity_holds(m, r, at, g, bt)
        n = m + r - 1
        lines.append(f"// ---- F({m}, {r}): {n} multiplies per {m} outputs (direct: {m * r})")
        lines.append(f"static const double CT_G_{m}_{r}[{n}][{r}] = {{")
        for row in g:
            lines.append("    {" + ", ".join(f"{v.numerator}.0 / {v.denominator}.0" for v in row) + "},")
        lines.append("};")
        lines.append(f"__device__ __forceinline__ void ct_input_{m}_{r}(const float (&d)[{n}], float (&v)[{n}]) {{")
        lines += _input_transform(m, r, bt)
        lines.append("}")
        lines.append(f"__device__ __forceinline__ void ct_output_{m}_{r}(const float (&mm)[{n}], float (&y)[{m}]) {{")
        lines += _output_transform(m, r, at)
        lines.append("}")
        lines.append("")
    return "\n".join(lines)
