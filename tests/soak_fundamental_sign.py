#!/usr/bin/env python
"""Soak for VERDICT r4 "next 1": the sign of F.  N random fundamental-matrix problems (8 - 3000 correspondences, 5 - 70 %
outliers, random thresholds and RANSAC seeds, a third of them with the real-focal-length filter) through ransac_fundamental
and estimate_fundamental of (a) the oracle and (b) the reference's own sources (oracle/_ref), compared SIGN-SENSITIVELY.
Writes profiles/r05_soak_fundamental_sign.md.  CPU only.   python tests/soak_fundamental_sign.py [problems=300] [seed=0]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import ref_lib as R  # noqa: E402
from poselib_amd import synth  # noqa: E402


def main():
    n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rs = np.random.RandomState(seed)
    rows = {fn: dict(runs=0, bit=0, close=0, negated=0, other=0, decisions=0, masks=0) for fn in ("ransac_fundamental", "estimate_fundamental")}
    worst = 0.0
    t0 = time.time()
    for i in range(n_prob):
        n = int(rs.randint(8, 3000))
        d = synth.fundamental_scene(n, float(rs.uniform(0.05, 0.7)), 70000 + 1000 * seed + i)
        for fn in rows:
            opt = {"max_error": float(rs.uniform(0.5, 3.0)), "real_focal_check": bool(i % 3 == 0),
                   "ransac": {"seed": int(rs.randint(0, 1 << 30)), "max_iterations": 2000}}
            Fa, ka, sa = getattr(O, fn)(d["x1"], d["x2"], opt)
            with R.reference() as ref:
                Fb, kb, sb = getattr(ref, fn)(d["x1"], d["x2"], opt)
            Fa, Fb = np.asarray(Fa), np.asarray(Fb)
            r = rows[fn]
            r["runs"] += 1
            r["decisions"] += (sa["iterations"], sa["refinements"], sa["num_inliers"]) == (sb["iterations"], sb["refinements"], sb["num_inliers"])
            r["masks"] += bool(np.array_equal(ka, kb))
            scale = max(np.abs(Fb).max(), 1e-300)
            if np.array_equal(Fa, Fb):
                r["bit"] += 1
            elif np.abs(Fa - Fb).max() <= 1e-9 * scale:
                r["close"] += 1
                worst = max(worst, np.abs(Fa - Fb).max() / scale)
            elif np.abs(Fa + Fb).max() <= 1e-9 * scale:
                r["negated"] += 1
            else:
                r["other"] += 1
    out = os.path.join(os.path.dirname(HERE), "profiles", "r05_soak_fundamental_sign.md")
    with open(out, "w") as f:
        f.write("# r05 - sign of F: oracle vs the reference's own sources (oracle/_ref), sign-SENSITIVE\n\n")
        f.write(f"`python tests/soak_fundamental_sign.py {n_prob} {seed}` ({time.time() - t0:.0f} s, CPU): {n_prob} random problems, "
                "8 - 3000 correspondences, 5 - 70 % outliers, max_error 0.5 - 3 px, random RANSAC seeds, every third problem with "
                "`real_focal_check`.\n\n")
        f.write("| entry point | runs | decisions identical | masks identical | F bit-identical | same sign, <= 1e-9 | NEGATED | other |\n|---|---|---|---|---|---|---|---|\n")
        for fn, r in rows.items():
            f.write(f"| `{fn}` | {r['runs']} | {r['decisions']} | {r['masks']} | {r['bit']} | {r['close']} | **{r['negated']}** | {r['other']} |\n")
        f.write(f"\nworst relative difference among the same-sign, not bit-identical runs: {worst:.3g}\n\n")
        f.write("Round 4 (one-sided Jacobi in three variants, cubic coefficients by polynomial arithmetic): the oracle returned -F of the "
                "reference-sources build in 20 - 37 % of such runs (VERDICT r4).  Round 5: one Eigen-ordered two-sided Jacobi routine "
                "(`oracle/eigen_shim/Eigen/src/JacobiSVD3x3.h`; product: `poselib_amd/csrc/pl_svd3.h`, bit-identical, "
                "`tests/test_hostmath_vs_oracle.py`), the 7-point cubic summed in the order of `relpose_7pt.cc:22-37`.\n")
    print(open(out).read())
    bad = sum(r["negated"] + r["other"] for r in rows.values())
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
