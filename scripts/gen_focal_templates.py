#!/usr/bin/env python3
"""Generates the coefficient / elimination-template tables of the two focal-length minimal solvers FROM THEIR EQUATIONS.

The reference solves both problems with an action-matrix template (solvers/p35pf.cc:86-886, solvers/relpose_6pt_focal.cc:54-1081):
the coefficients of a fixed set of polynomial equations, written into a sparse matrix [C0 | C1] whose rows are monomial multiples of
the equations and whose columns are monomials; C0^-1 C1 gives the action matrix.  To take the reference's DECISIONS a restatement
must produce the same numbers, so this script rebuilds the same object from its definition -

  * the equations (stated below from the geometry, with the file:line of the reference's statement),
  * the order of the monomials inside an equation and of the equations,
  * the rows (equation, multiplier) and the columns (monomials) of the template,
  * the order in which the products of one coefficient are added (ascending in the reversed index tuple - the order the
    reference's formulas are written in; floating-point addition is not associative, so the order is part of the arithmetic),

expands the polynomials symbolically (integer arithmetic on index tuples, below) and writes the result as TABLES: a term list per
coefficient and a (position, coefficient) list per matrix.  Nothing is read from /root/reference; the output is checked against
the reference's sources where they exist by tests/test_reference_focal_estimator.py (oracle == oracle/_ref bit for bit).

Usage: python scripts/gen_focal_templates.py   (writes oracle/src/focal_templates.inc and poselib_amd/csrc/pl_focal_templates.h)
"""
import itertools
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- polynomials in the unknowns whose coefficients are integer combinations of products of data entries d[i] -------------------
# {monomial (exponent tuple): {term (sorted tuple of d indices): integer}}
def p_add(p, q, s=1):
    r = defaultdict(lambda: defaultdict(int))
    for src, f in ((p, 1), (q, s)):
        for m, terms in src.items():
            for t, c in terms.items():
                r[m][t] += f * c
    return p_clean(r)


def p_clean(p):
    out = {}
    for m, terms in p.items():
        tt = {t: c for t, c in terms.items() if c != 0}
        if tt:
            out[m] = tt
    return out


def p_mul(p, q):
    r = defaultdict(lambda: defaultdict(int))
    for m1, t1 in p.items():
        for m2, t2 in q.items():
            m = tuple(a + b for a, b in zip(m1, m2))
            for a, ca in t1.items():
                for b, cb in t2.items():
                    r[m][tuple(sorted(a + b))] += ca * cb
    return p_clean(r)


def p_scale(p, s):
    return {m: {t: s * c for t, c in terms.items()} for m, terms in p.items()}


def p_sum(ps):
    r = {}
    for p in ps:
        r = p_add(r, p)
    return r


def linear(nvars, entries):
    """sum_k unknown_k * d[entries[k]]; entries[k] = (exponent tuple of the unknown, d index)"""
    return {m: {(i,): 1} for m, i in entries}


def term_key(t):
    return tuple(reversed(t))


# ---- P3.5Pf (solvers/p35pf.cc) -------------------------------------------------------------------------------------------------
# P = sum_k alpha_k N_k over the null space of the seven linear constraints (p35pf.cc:68-84), N 12 x 5 column-major in d, the 3 x 4
# matrix P column-major in each N_k (p35pf.cc:906-907): row r of the left 3 x 3 block is (d[12k+r], d[12k+3+r], d[12k+6+r]).
def p35_tables():
    nv = 5

    def unit(k):
        return tuple(1 if i == k else 0 for i in range(nv))

    a = [[linear(nv, [(unit(k), 12 * k + 3 * c + r) for k in range(nv)]) for c in range(3)] for r in range(3)]  # a[row][xyz]
    a1, a2, a3 = a
    X, Y, Z = 0, 1, 2

    def dot(u, v):
        return p_sum(p_mul(u[i], v[i]) for i in range(3))

    def prod(*fs):
        r = fs[0]
        for f in fs[1:]:
            r = p_mul(r, f)
        return r

    # A A^T = s^2 diag(f^2, f^2, 1): rows orthogonal, first two of equal norm (coefficients p35pf.cc:89-160)
    quadrics = [dot(a2, a3), dot(a1, a3), dot(a1, a2), p_add(dot(a1, a1), dot(a2, a2), -1)]
    # the five cubics that remove the f = 0 solutions (p35pf.cc:161-520): with W = a1 a1^T + a2 a2^T (indexed by column) and
    # n2 = |a2|^2, W + f^2 a3 a3^T = n2 I; f^2 eliminated between pairs of entries
    #   c0 = (Wzz - n2) a3y - Wyz a3z      c1 = Wyz a3y - (Wyy - n2) a3z      c2 = Wxz a3y - Wxy a3z
    #   c3 = a1z (a1 x a3)_y + a2y (a2 x a3)_z                                 c4 = Wyz a3x - Wxy a3z
    def W(i, j):
        return p_add(p_mul(a1[i], a1[j]), p_mul(a2[i], a2[j]))

    n2 = dot(a2, a2)
    cubics = [
        p_add(p_mul(p_add(W(Z, Z), n2, -1), a3[Y]), p_mul(W(Y, Z), a3[Z]), -1),
        p_add(p_mul(W(Y, Z), a3[Y]), p_mul(p_add(W(Y, Y), n2, -1), a3[Z]), -1),
        p_add(p_mul(W(X, Z), a3[Y]), p_mul(W(X, Y), a3[Z]), -1),
        p_add(p_mul(a1[Z], p_add(p_mul(a1[Z], a3[X]), p_mul(a1[X], a3[Z]), -1)),
              p_mul(a2[Y], p_add(p_mul(a2[X], a3[Y]), p_mul(a2[Y], a3[X]), -1))),
        p_add(p_mul(W(Y, Z), a3[X]), p_mul(W(X, Y), a3[Z]), -1),
    ]

    def monos(deg):
        out = [e for e in itertools.product(range(deg + 1), repeat=nv) if sum(e) == deg]
        out.sort(key=lambda e: tuple(reversed(e)))
        return out

    coeffs = []  # list of term lists
    where = {}  # (equation index, dehomogenised monomial) -> coefficient index
    for e, q in enumerate(quadrics):
        for m in monos(2):
            where[(e, m[:4])] = len(coeffs)
            coeffs.append(q.get(m, {}))
    for e, c in enumerate(cubics):
        for m in monos(3):
            where[(4 + e, m[:4])] = len(coeffs)
            coeffs.append(c.get(m, {}))
    assert len(coeffs) == 235

    # the template (p35pf.cc:522-871): unknowns (x, y, z, w) = alpha_0..3, alpha_4 = 1; rows = (equation, multiplier), the cubics
    # unmultiplied; columns = the 35 monomials of degree <= 3, the last ten the basis of the quotient ring
    def mono(s):
        return tuple(s.count(ch) for ch in "xyzw")

    rows = [(0, "x"), (0, "y"), (1, "y"), (1, "x"), (1, "z"), (0, "z"), (2, "z"), (2, "y"), (2, "x"), (3, "z"), (2, "w"), (1, "w"),
            (3, "w"), (0, "w"), (3, "y"), (3, ""), (2, ""), (0, ""), (1, ""), (3, "x"), (4, ""), (5, ""), (6, ""), (7, ""), (8, "")]
    cols = ["xxx", "xxy", "xyy", "yyy", "xxz", "xyz", "yyz", "xzz", "yzz", "zzz", "xxw", "xyw", "yyw", "xzw", "yzw", "xx", "xy",
            "yy", "xz", "yz", "xww", "yww", "zzw", "zww", "www", "", "x", "xw", "y", "yw", "z", "zz", "zw", "w", "ww"]
    col_of = {mono(c): i for i, c in enumerate(cols)}
    assert len(col_of) == 35
    entries = []  # (row, column, coefficient)
    for r, (e, mult) in enumerate(rows):
        mm = mono(mult)
        for (ee, m), ci in where.items():
            if ee != e:
                continue
            entries.append((r, col_of[tuple(x + y for x, y in zip(m, mm))], ci))
    entries.sort(key=lambda t: (t[1], t[0]))
    return dict(name="P35", ndata=60, coeffs=coeffs, powers=False, nrows=25, ncols=35, nelim=25, entries=entries)


# ---- shared focal length, six points (solvers/relpose_6pt_focal.cc) ---------------------------------------------------------------
# F = F0 + x F1 + y F2 (null space of the epipolar constraints, relpose_6pt_focal.cc:1086-1094, each 3 x 3 column-major in d),
# w = 1 / f^2, Q = diag(1, 1, w):  2 F Q F^T Q F - tr(F Q F^T Q) F = 0 (nine equations, column-major) and det F = 0
def six_tables():
    def mono(s):
        return tuple(s.count(ch) for ch in "xyw")

    one, mx, my, mw = mono(""), mono("x"), mono("y"), mono("w")
    F = [[{one: {(3 * j + i,): 1}, mx: {(9 + 3 * j + i,): 1}, my: {(18 + 3 * j + i,): 1}} for j in range(3)] for i in range(3)]
    Q = [{one: {(): 1}}, {one: {(): 1}}, {mw: {(): 1}}]  # diagonal

    def matmul(A, B):
        return [[p_sum(p_mul(A[i][k], B[k][j]) for k in range(3)) for j in range(3)] for i in range(3)]

    def times_Q(A):  # A Q
        return [[p_mul(A[i][j], Q[j]) for j in range(3)] for i in range(3)]

    Ft = [[F[j][i] for j in range(3)] for i in range(3)]
    FQ = times_Q(F)
    G = matmul(FQ, times_Q(Ft))  # F Q F^T Q
    GF = matmul(G, F)
    tr = p_sum(G[i][i] for i in range(3))
    eqs = [p_add(p_scale(GF[i][j], 2), p_mul(tr, F[i][j]), -1) for j in range(3) for i in range(3)]
    det = p_sum([
        p_mul(F[0][0], p_add(p_mul(F[1][1], F[2][2]), p_mul(F[1][2], F[2][1]), -1)),
        p_scale(p_mul(F[0][1], p_add(p_mul(F[1][0], F[2][2]), p_mul(F[1][2], F[2][0]), -1)), -1),
        p_mul(F[0][2], p_add(p_mul(F[1][0], F[2][1]), p_mul(F[1][1], F[2][0]), -1)),
    ])
    eqs.append(det)

    key = lambda m: (-(m[0] + m[1] + m[2]), -(m[0] + m[1]), -m[0])
    mon_e = sorted([(a, b, c) for a in range(4) for b in range(4 - a) for c in range(3)], key=key)
    mon_d = sorted([(a, b, 0) for a in range(4) for b in range(4 - a)], key=key)
    coeffs, where = [], {}
    for e in range(10):
        for m in (mon_e if e < 9 else mon_d):
            where[(e, m)] = len(coeffs)
            coeffs.append(eqs[e].get(m, {}))
        assert set(eqs[e]) <= set(mon_e if e < 9 else mon_d)
    assert len(coeffs) == 280

    # the template (relpose_6pt_focal.cc:1031-1043): equation e = 3 j + i (det = 9) times a power of w; an entry whose monomial is
    # not among the 46 columns is left out (the columns y^k w^4 were pruned by the template's generator: the reductions that are
    # used do not need them)
    rows = [(0, "ww"), (1, "ww"), (2, "ww"), (1, "w"), (0, "w"), (2, "w"), (3, "ww"), (4, "ww"), (5, "ww"), (3, ""), (3, "w"),
            (2, ""), (4, ""), (4, "w"), (5, "w"), (6, "ww"), (7, "ww"), (0, ""), (6, ""), (6, "w"), (5, ""), (7, ""), (7, "w"),
            (1, ""), (8, ""), (8, "w"), (8, "ww"), (9, ""), (9, "w"), (9, "ww"), (9, "www")]
    cols = ["xxxwwww", "xxywwww", "xyywwww", "xxxwww", "xxywww", "xyywww", "yyywww", "xxwwww", "xywwww", "xxxww", "xxyww",
            "xyyww", "yyyww", "xxwww", "xywww", "yywww", "xwwww", "xxxw", "xxyw", "xxww", "xwww", "xxx", "xxw", "xxy", "xyy",
            "xyyw", "xyww", "yyy", "yyyw", "yyww", "ywww", "", "x", "xx", "xy", "xyw", "xw", "xww", "y", "yy", "yyw", "yw", "yww",
            "w", "ww", "www"]
    col_of = {mono(c): i for i, c in enumerate(cols)}
    assert len(col_of) == 46
    entries = []
    for r, (e, mult) in enumerate(rows):
        mm = mono(mult)
        for (ee, m), ci in where.items():
            if ee != e:
                continue
            c = col_of.get(tuple(x + y for x, y in zip(m, mm)))
            if c is not None:
                entries.append((r, c, ci))
    entries.sort(key=lambda t: (t[1], t[0]))
    return dict(name="Six", ndata=27, coeffs=coeffs, powers=True, nrows=31, ncols=46, nelim=31, entries=entries)


# ---- output ---------------------------------------------------------------------------------------------------------------------
MULT_CODES = {1: 0, -1: 1, 2: 2, -2: 3, 3: 4, -3: 5, 6: 6, -6: 7}


def encode_terms(tb):
    """A term = m * d[a] * d[b] (* d[c]), a <= b <= c (c = 255: two factors), m one of +-1, +-2, +-3, +-6.  Evaluated from the left
    as the reference's formulas are written: t = m; t *= factor, factor by factor in ascending index order - where, for the
    six-point solver (`powers`), a repeated index is ONE factor, pow(d, 2) = d * d or pow(d, 3) (the libm call), and for P3.5Pf
    every index is a factor of its own.  (+-1 and +-2 scale exactly; 3 and 6 do not, so the position of m matters.)"""
    starts, flat = [0], []
    for terms in tb["coeffs"]:
        for t in sorted(terms, key=term_key):
            idx = list(t)
            if len(idx) == 2:
                idx.append(255)
            assert len(idx) == 3 and idx[0] <= idx[1] <= idx[2]
            flat.append((MULT_CODES[terms[t]], idx[0], idx[1], idx[2]))
        starts.append(len(flat))
    return starts, flat


def emit(tb, ns_comment):
    n = tb["name"]
    starts, flat = encode_terms(tb)
    out = []
    out.append(f"// {n}: {len(tb['coeffs'])} coefficients, {len(flat)} products; template {tb['nrows']} x {tb['ncols']}, "
               f"{len(tb['entries'])} entries")
    out.append(f"static constexpr int k{n}Coeffs = {len(tb['coeffs'])}, k{n}Terms = {len(flat)}, k{n}Rows = {tb['nrows']}, "
               f"k{n}Cols = {tb['ncols']}, k{n}Entries = {len(tb['entries'])};")

    def arr(ctype, name, vals, per=24):
        out.append(f"static constexpr {ctype} {name}[{len(vals)}] = {{")
        for i in range(0, len(vals), per):
            out.append("    " + ", ".join(str(v) for v in vals[i:i + per]) + ",")
        out.append("};")

    arr("uint16_t", f"k{n}TermStart", starts)
    # a term packed into 32 bits: multiplier code (index into {1, -1, 2, -2, 3, -3, 6, -6}) | a << 8 | b << 16 | c << 24
    arr("uint32_t", f"k{n}TermPacked", [m | (a << 8) | (b << 16) | (c << 24) for m, a, b, c in flat], per=12)
    # entries sorted by column, then row; ColStart[c] .. ColStart[c + 1]: the entries of column c
    colstart = [0] * (tb["ncols"] + 1)
    for r, c, ci in tb["entries"]:
        colstart[c + 1] += 1
    for c in range(tb["ncols"]):
        colstart[c + 1] += colstart[c]
    arr("uint16_t", f"k{n}ColStart", colstart)
    arr("uint8_t", f"k{n}EntryRow", [r for r, c, ci in tb["entries"]], per=32)
    arr("uint16_t", f"k{n}EntryCoeff", [ci for r, c, ci in tb["entries"]])
    return "\n".join(out)


HEADER = """// GENERATED by scripts/gen_focal_templates.py - do not edit.  The coefficient polynomials and the elimination templates of the
// two focal-length minimal solvers, rebuilt from their equations (the script states them and cites solvers/p35pf.cc and
// solvers/relpose_6pt_focal.cc): per coefficient the products in the order they are added, per matrix the entries by column.
"""


def main():
    tabs = [p35_tables(), six_tables()]
    body = "\n\n".join(emit(t, "") for t in tabs)
    with open(os.path.join(ROOT, "oracle", "src", "focal_templates.inc"), "w") as f:
        f.write("// ORACLE - TEST INFRASTRUCTURE ONLY.\n" + HEADER + "// (included inside namespace orc)\n" + body + "\n")
    with open(os.path.join(ROOT, "poselib_amd", "csrc", "pl_focal_templates.h"), "w") as f:
        f.write(HEADER + "#pragma once\n#include <cstdint>\nnamespace pl {\n" + body + "\n} // namespace pl\n")
    return tabs


if __name__ == "__main__":
    for t in main():
        print(t["name"], len(t["coeffs"]), "coefficients,", sum(len(c) for c in t["coeffs"]), "products,", len(t["entries"]), "entries")
