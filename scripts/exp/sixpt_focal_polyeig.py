"""Experiment (CPU, numpy): the 6-point shared-focal relative pose solver from first principles, as a polynomial
eigenvalue problem in w = 1/f^2 (hidden variable), checked against the reference's relpose_6pt_shared_focal
(oracle/_ref).  Nothing of the reference's generated template is used.

F = N0 + x N1 + y N2 (null space of the six epipolar constraints), Q = diag(1, 1, w):
    det F = 0                                            (1 cubic in x, y)
    2 F Q F^T Q F - trace(F Q F^T Q) F = 0               (9 cubics in x, y; quadratic in w)
=> (C0 + w C1 + w^2 C2) m(x, y) = 0 with m the 10 monomials of degree <= 3.
"""
import ctypes as C
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))

MON = [(a, b) for d in (3, 2, 1, 0) for a in range(d, -1, -1) for b in [d - a]]  # x^a y^b, graded
IDX = {m: i for i, m in enumerate(MON)}


def pmul(p, q):
    r = {}
    for (a, b), u in p.items():
        for (c, d), v in q.items():
            r[(a + c, b + d)] = r.get((a + c, b + d), 0.0) + u * v
    return r


def padd(p, q, s=1.0):
    r = dict(p)
    for k, v in q.items():
        r[k] = r.get(k, 0.0) + s * v
    return r


def build(N):
    """N: 9 x 3 (column-major F vectors).  Returns C0, C1, C2 (10 x 10)."""
    F = [[None] * 3 for _ in range(3)]
    for r in range(3):
        for c in range(3):
            k = c * 3 + r
            F[r][c] = {(0, 0): N[k, 0], (1, 0): N[k, 1], (0, 1): N[k, 2]}
    Cs = [np.zeros((10, 10)) for _ in range(3)]
    # det
    det = {}
    for perm in itertools.permutations(range(3)):
        sign = np.linalg.det(np.eye(3)[list(perm)])
        det = padd(det, pmul(pmul(F[0][perm[0]], F[1][perm[1]]), F[2][perm[2]]), sign)
    for m, v in det.items():
        Cs[0][0, IDX[m]] = v
    # polynomials in w: dict wdeg -> poly
    def wmul(P, Q):
        R = {}
        for i, p in P.items():
            for j, q in Q.items():
                R[i + j] = padd(R.get(i + j, {}), pmul(p, q))
        return R
    def wadd(P, Q, s=1.0):
        R = {k: dict(v) for k, v in P.items()}
        for k, v in Q.items():
            R[k] = padd(R.get(k, {}), v, s)
        return R
    q = [0, 0, 1]  # w degree of Q's diagonal
    # G = F Q F^T: G[i][j] = sum_k q_k F[i][k] F[j][k]
    G = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(3):
            acc = {}
            for k in range(3):
                acc = wadd(acc, {q[k]: pmul(F[i][k], F[j][k])})
            G[i][j] = acc
    # trace(G Q) = sum_i q_i G[i][i]
    tr = {}
    for i in range(3):
        tr = wadd(tr, {k + q[i]: v for k, v in G[i][i].items()})
    row = 1
    for i in range(3):
        for j in range(3):
            # (G Q F)[i][j] = sum_k G[i][k] q_k F[k][j]
            acc = {}
            for k in range(3):
                acc = wadd(acc, wmul({q[k]: {(0, 0): 1.0}}, wmul(G[i][k], {0: F[k][j]})))
            eq = wadd({k: {m: 2 * v for m, v in p.items()} for k, p in acc.items()}, wmul(tr, {0: F[i][j]}), -1.0)
            for wd, p in eq.items():
                for m, v in p.items():
                    Cs[wd][row, IDX[m]] = v
            row += 1
    return Cs


def nullspace(x1, x2):
    A = np.zeros((6, 9))
    for i in range(6):
        A[i] = np.concatenate([x1[i, 0] * x2[i], x1[i, 1] * x2[i], x1[i, 2] * x2[i]])
    _, _, vt = np.linalg.svd(A)
    return vt[6:].T  # 9 x 3


def solve(x1, x2):
    import scipy.linalg
    N = nullspace(x1, x2)
    C0, C1, C2 = build(N)
    Z, I = np.zeros((10, 10)), np.eye(10)
    A = np.block([[Z, I], [-C0, -C1]])
    B = np.block([[I, Z], [Z, C2]])
    w, V = scipy.linalg.eig(A, B)
    out = []
    for k in range(20):
        if not np.isfinite(w[k]) or abs(w[k].imag) > 1e-8 * (1 + abs(w[k].real)) or w[k].real < 1e-8:
            continue
        v = V[:10, k].real
        x, y = v[IDX[(1, 0)]] / v[IDX[(0, 0)]], v[IDX[(0, 1)]] / v[IDX[(0, 0)]]
        out.append((1 / np.sqrt(w[k].real), x, y))
    return sorted(out), w


def scene(rng):
    f = rng.uniform(300, 3000)
    X = rng.uniform(-1, 1, (6, 3)) * [2, 2, 1] + [0, 0, 5]
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(rng.normal(size=3) * 0.2).as_matrix()
    t = rng.normal(size=3)
    t /= np.linalg.norm(t)
    X2 = X @ R.T + t
    x1 = f * X[:, :2] / X[:, 2:]
    x2 = f * X2[:, :2] / X2[:, 2:]
    return f, x1, x2


def main():
    import ref_lib
    lib = C.CDLL(ref_lib.build())
    rng = np.random.default_rng(1)
    tot = match = extra = 0
    for trial in range(200):
        f, x1, x2 = scene(rng)
        scale = rng.uniform(500, 2000)  # the estimator works on normalised pixels: focal of order 1
        b1 = np.c_[x1 / scale, np.ones(6)]
        b2 = np.c_[x2 / scale, np.ones(6)]
        b1 /= np.linalg.norm(b1, axis=1)[:, None]
        b2 /= np.linalg.norm(b2, axis=1)[:, None]
        poses = np.zeros((60, 7))
        foc = np.zeros(60)
        n = lib.ref_relpose_6pt_shared_focal(b1.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p),
                                             poses.ctypes.data_as(C.c_void_p), foc.ctypes.data_as(C.c_void_p))
        ref_f = sorted(set(np.round(foc[:n], 12)))
        mine, w = solve(b1, b2)
        mine_f = [m[0] for m in mine]
        for rf in ref_f:
            tot += 1
            if any(abs(rf - mf) < 1e-6 * rf for mf in mine_f):
                match += 1
        for mf in mine_f:
            if not any(abs(rf - mf) < 1e-6 * rf for rf in ref_f):
                extra += 1
        if trial < 3:
            print("ref", ref_f, "\nmine", mine_f, "\n gt", f / scale)
            print(" eig", np.sort_complex(w[np.isfinite(w)]))
    print(f"reference focal lengths found: {match} of {tot}; mine not in the reference's: {extra}")


if __name__ == "__main__":
    main()
