"""Experiment (round 3, CPU only): a COMPACT P3.5Pf solver derived from first principles, checked against the reference's p35pf.

Unknown: P = s K [R | t], K = diag(f, f, 1), in the 5-dimensional null space of the 7 linear constraints (three points with both
coordinates, the fourth with one): P = sum_k alpha_k N_k, alpha_5 = 1.  With a1, a2, a3 the rows of the left 3 x 3 block A:
  quadrics   a1.a2 = 0,  a1.a3 = 0,  a2.a3 = 0,  |a1|^2 = |a2|^2                                   (A A^T = s^2 diag(f^2, f^2, 1))
  cubics     (a2 x a3)_i (a2)_j = (a3 x a1)_j (a1)_i,  i, j = 1..3                                 (cof(A) = det(A) D^-1 A: the first
             two rows of the cofactor matrix are the SAME multiple of a1 and a2)
The quadrics alone have 16 roots, six of them with a1 || a2 isotropic (f = 0); the cubics remove exactly those.  Quadrics x {1, x1..x4}
and the nine cubics are 29 equations in the 35 monomials of degree <= 3: rank 25, null space 10 = the solver's solution count, standard
monomials {1, x1, x2, x3, x4, x1 x4, x2 x4, x3 x4, x4^2, x3^2} (found numerically, grevlex).  One Gauss-Jordan elimination of the 29 x 35
matrix expresses the other 25 monomials in those ten; the 10 x 10 action matrix of x4 follows by reading off rows; its eigenvectors
hold (1, x1, .., x4) at the roots.  Nothing of the reference's generated template is used; its OUTPUT is what this is compared with."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import ref_lib  # noqa: E402
from poselib_amd import synth  # noqa: E402

NV = 4


def monos_upto(deg):
    out = []
    for d in range(deg + 1):
        for c in itertools.combinations_with_replacement(range(NV), d):
            e = [0] * NV
            for k in c:
                e[k] += 1
            out.append(tuple(e))
    return out


COLS = sorted(monos_upto(3), key=lambda m: (sum(m), tuple(-v for v in reversed(m))), reverse=True)
CIDX = {m: i for i, m in enumerate(COLS)}
BASIS = [(0, 0, 2, 0), (1, 0, 0, 1), (0, 1, 0, 1), (0, 0, 1, 1), (0, 0, 0, 2), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1),
         (0, 0, 0, 0)]
BIDX = [CIDX[m] for m in BASIS]
ELIM = [c for c in range(len(COLS)) if c not in BIDX]
ACT = 3  # multiply by x4


def pmul(p, q):
    r = {}
    for e1, c1 in p.items():
        for e2, c2 in q.items():
            e = tuple(a + b for a, b in zip(e1, e2))
            r[e] = r.get(e, 0.0) + c1 * c2
    return r


def padd(p, q, s=1.0):
    r = dict(p)
    for e, c in q.items():
        r[e] = r.get(e, 0.0) + s * c
    return r


def null_space(x, X):
    rows = []
    for i in range(4):
        Xh = np.r_[X[i], 1.0]
        rows.append(np.r_[Xh, np.zeros(4), -x[i, 0] * Xh])
        if i < 3:
            rows.append(np.r_[np.zeros(4), Xh, -x[i, 1] * Xh])
    _, _, Vt = np.linalg.svd(np.array(rows))
    return Vt[7:].T


def equations(N):
    rows3 = []
    for rows in ([0, 1, 2], [4, 5, 6], [8, 9, 10]):
        v = []
        for row in rows:
            p = {}
            for k in range(5):
                e = [0] * NV
                if k < 4:
                    e[k] = 1
                p[tuple(e)] = p.get(tuple(e), 0.0) + N[row, k]
            v.append(p)
        rows3.append(v)
    a1, a2, a3 = rows3

    def dot(u, v):
        return padd(padd(pmul(u[0], v[0]), pmul(u[1], v[1])), pmul(u[2], v[2]))

    def cross(u, v):
        return [padd(pmul(u[1], v[2]), pmul(u[2], v[1]), -1.0), padd(pmul(u[2], v[0]), pmul(u[0], v[2]), -1.0),
                padd(pmul(u[0], v[1]), pmul(u[1], v[0]), -1.0)]

    quads = [dot(a1, a2), dot(a1, a3), dot(a2, a3), padd(dot(a1, a1), dot(a2, a2), -1.0)]
    c23, c31 = cross(a2, a3), cross(a3, a1)
    cubics = [padd(pmul(c23[i], a2[j]), pmul(c31[j], a1[i]), -1.0) for i in range(3) for j in range(3)]
    eqs = []
    for p in quads:
        for s in monos_upto(1):
            eqs.append({tuple(a + b for a, b in zip(e, s)): c for e, c in p.items()})
    return eqs + cubics


def solve(x, X, scale=1000.0):
    N = null_space(x / scale, X)
    M = np.zeros((29, 35))
    for r, p in enumerate(equations(N)):
        for e, c in p.items():
            M[r, CIDX[e]] += c
    M /= np.abs(M).max(1, keepdims=True)
    # the 25 eliminated monomials in terms of the basis: least squares = Gauss-Jordan on a consistent, rank-25 system
    red, *_ = np.linalg.lstsq(M[:, ELIM], -M[:, BIDX], rcond=None)  # 25 x 10
    AM = np.zeros((10, 10))
    for k, b in enumerate(BASIS):
        m = list(b)
        m[ACT] += 1
        c = CIDX[tuple(m)]
        if c in BIDX:
            AM[k, BIDX.index(c)] = 1.0
        else:
            AM[k] = red[ELIM.index(c)]
    ev, V = np.linalg.eig(AM)  # AM v = x4 v with v = basis monomials at a root
    out = []
    for k in range(10):
        if abs(ev[k].imag) > 1e-8 * max(1.0, abs(ev[k])):
            continue
        v = (V[:, k] / V[9, k]).real
        a = np.r_[v[5], v[6], v[7], v[8], 1.0]
        P = (N @ a).reshape(3, 4)
        n3 = np.linalg.norm(P[2, :3])
        P = P / n3
        if np.linalg.det(P[:, :3]) < 0:
            P = -P
        f = np.linalg.norm(P[0, :3])
        R = np.vstack([P[0, :3] / f, P[1, :3] / np.linalg.norm(P[1, :3]), P[2, :3]])
        out.append((scale * f, R, np.r_[P[0, 3] / f, P[1, 3] / f, P[2, 3]]))
    return out


def main():
    matched = total = extra = 0
    worst = 0.0
    for sd in range(400):
        d = synth.absolute_pose_scene(4, 0.0, 50000 + sd, noise_px=0.0 if sd % 2 else 2.0)
        f0, cx, cy = d["camera"]["params"]
        x = np.asarray(d["p2d"]) - np.array([cx, cy])
        X = np.asarray(d["p3d"])
        mine = solve(x, X)
        with ref_lib.reference():
            rp, rf = O.p35pf(x, X)
        for f, p in zip(rf, rp):
            Rr = synth.quat_to_rotmat(p[:4])
            e = min((abs(m[0] - f) / f + np.abs(m[1] - Rr).max() + np.abs(m[2] - p[4:]).max() / max(1.0, np.abs(p[4:]).max()) for m in mine), default=1.0)
            total += 1
            if e < 1e-6:
                matched += 1
                worst = max(worst, e)
        extra += max(0, len(mine) - len(rf))
    print(f"compact template (29 x 35 elimination, 10 x 10 action matrix) vs the reference's p35pf on 400 minimal problems: "
          f"{matched} of {total} solutions matched (focal length, R, t; worst matched error {worst:.1e}); {extra} real roots the reference does not return")


if __name__ == "__main__":
    main()
