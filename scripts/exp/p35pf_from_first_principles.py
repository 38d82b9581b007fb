"""Experiment (round 3, CPU only): is there a P3.5Pf formulation that can be written down WITHOUT the reference's generated
elimination template and still has the reference's solutions?

P = K [R | t] up to scale, K = diag(f, f, 1), lives in the 5-dimensional null space of the 7 linear constraints (three points, both
coordinates; the fourth, one coordinate).  The rows a1, a2, a3 of its left 3 x 3 block must satisfy
    a1 . a2 = 0,   a1 . a3 = 0,   a2 . a3 = 0,   |a1|^2 = |a2|^2
- four quadrics in the four de-homogenised null-space coordinates, 16 complex solutions.  Solved here GENERICALLY (Macaulay matrix of
degree 6, null space, eigenvectors of a random multiplication matrix - numpy, nothing of the reference), then compared with the
reference's own p35pf (oracle/_ref: solvers/p35pf.cc compiled against the eigen shim).

Result (20 random minimal problems, with and without noise): the REAL solutions of the four quadrics are exactly the reference's
solutions - 82 of 82 matched in focal length and rotation, same number of real solutions in every problem; the six extra roots of
the quadric system were complex every time.  So an independent solver exists; what it still lacks is an elimination small enough for
one GPU lane (the generic route needs the null space of a 280 x 210 matrix per sample) - DESIGN.md 8."""
import sys, itertools
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, oracle_lib as O, ref_lib
from poselib_amd import synth

def nullspace_P(x, X):
    # unknown P (3x4, row-major p = [P1 P2 P3]); constraints: P1.Xh - u P3.Xh = 0 (4 points), P2.Xh - v P3.Xh = 0 (first 3 points)
    rows = []
    for i in range(4):
        Xh = np.r_[X[i], 1.0]
        rows.append(np.r_[Xh, np.zeros(4), -x[i, 0] * Xh])
        if i < 3:
            rows.append(np.r_[np.zeros(4), Xh, -x[i, 1] * Xh])
    M = np.array(rows)              # 7 x 12
    _, _, Vt = np.linalg.svd(M)
    return Vt[7:].T                 # 12 x 5

def quadrics(N):
    # P(alpha) = sum alpha_k N[:,k]; a1 = P[0,:3], a2 = P[1,:3], a3 = P[2,:3]
    A = [N[0:3], N[4:7], N[8:11]]   # each 3 x 5: a_r = A[r] @ alpha
    def bil(r, s):                  # symmetric 5x5 form of a_r . a_s
        Q = A[r].T @ A[s]
        return 0.5 * (Q + Q.T)
    return [bil(0, 1), bil(0, 2), bil(1, 2), bil(0, 0) - bil(1, 1)]

def solve_quadrics(Qs):
    # dehomogenise alpha_5 = 1 -> 4 quadrics in 4 unknowns; Macaulay matrix of degree 5
    nv = 4
    def monos(deg):
        out = []
        for d in range(deg + 1):
            for c in itertools.combinations_with_replacement(range(nv), d):
                e = [0] * nv
                for k in c: e[k] += 1
                out.append(tuple(e))
        return out
    D = 6
    cols = monos(D); idx = {m: i for i, m in enumerate(cols)}
    polys = []
    for Q in Qs:
        p = {}
        for i in range(5):
            for j in range(5):
                e = [0] * nv
                if i < 4: e[i] += 1
                if j < 4: e[j] += 1
                p[tuple(e)] = p.get(tuple(e), 0.0) + Q[i, j]
        polys.append(p)
    rows = []
    for p in polys:
        for m in monos(D - 2):
            r = np.zeros(len(cols))
            for e, c in p.items():
                r[idx[tuple(a + b for a, b in zip(e, m))]] += c
            rows.append(r)
    Mac = np.array(rows)
    _, s, Vt = np.linalg.svd(Mac)
    rank = (s > 1e-9 * s[0]).sum()
    K = Vt[rank:].T                                   # null space: columns = evaluations of the monomial vector at the roots (mixed)
    nsol = K.shape[1]
    # multiplication by x0 restricted to monomials of degree <= D-1
    low = [m for m in cols if sum(m) <= D - 1]
    li = [idx[m] for m in low]
    def shifted(v):
        return [idx[tuple(a + (1 if k == v else 0) for k, a in enumerate(m))] for m in low]
    B = K[li]
    sols = None
    # random linear combination of the variables as the multiplier (generic)
    w = np.random.RandomState(0).randn(nv)
    Bs = sum(w[v] * K[shifted(v)] for v in range(nv))
    Mx, *_ = np.linalg.lstsq(B, Bs, rcond=None)
    ev, V = np.linalg.eig(Mx)
    Z = K @ V                                          # monomial vectors at the roots (up to scale)
    one = idx[tuple([0] * nv)]
    out = []
    for k in range(nsol):
        z = Z[:, k] / Z[one, k]
        alpha = np.array([z[idx[tuple(1 if q == v else 0 for q in range(nv))]] for v in range(nv)])
        out.append(alpha)
    return nsol, np.array(out)

def poses_from(alpha_all, N, x, X):
    res = []
    for al in alpha_all:
        if np.abs(al.imag).max() > 1e-7: continue
        a = np.r_[al.real, 1.0]
        P = (N @ a).reshape(3, 4)
        a1, a2, a3 = P[0, :3], P[1, :3], P[2, :3]
        n3 = np.linalg.norm(a3)
        if n3 < 1e-12: continue
        P = P / n3
        if np.linalg.det(np.vstack([P[0,:3], P[1,:3], P[2,:3]])) < 0: P = -P
        f = np.linalg.norm(P[0, :3])
        R = np.vstack([P[0, :3] / f, P[1, :3] / np.linalg.norm(P[1,:3]), P[2, :3]])
        t = np.r_[P[0, 3] / f, P[1, 3] / f, P[2, 3]]
        res.append((f, R, t))
    return res

ok = tot = 0
for seed in range(20):
    d = synth.absolute_pose_scene(4, 0.0, 9000 + seed, noise_px=0.0 if seed % 2 == 0 else 1.0)
    f0, cx, cy = d["camera"]["params"]
    x = np.asarray(d["p2d"]) - np.array([cx, cy]); X = np.asarray(d["p3d"])
    xs = x / 1000.0
    N = nullspace_P(xs, X)
    nsol, al = solve_quadrics(quadrics(N))
    mine = poses_from(al, N, xs, X)
    with ref_lib.reference():
        rp, rf = O.p35pf(x, X)
    mine_f = sorted(1000.0 * m[0] for m in mine)
    matched = 0
    for fr, pr in zip(rf, rp):
        Rr = synth.quat_to_rotmat(pr[:4])
        best = min((abs(1000.0 * m[0] - fr) / fr + np.abs(m[1] - Rr).max() for m in mine), default=9)
        matched += best < 1e-5
    tot += len(rf); ok += matched
    print(seed, "macaulay null dim", nsol, "real sols mine", len(mine), "ref", len(rf), "matched", matched, np.round(sorted(rf), 2), np.round(mine_f, 2))
print("matched", ok, "of", tot)
