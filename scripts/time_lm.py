#!/usr/bin/env python
"""Time of one LM iteration of k_lm (one workgroup per task): pl_refine_model on resident problems, slope between a short and a long
run (launch + synchronisation cancel).  python scripts/time_lm.py [truncated|cauchy]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P
from poselib_amd import synth
loss = (sys.argv[1] if len(sys.argv) > 1 else "truncated").upper()
rs = np.random.RandomState(0)
def timed(fn, reps=15):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    return (time.perf_counter() - t0) / reps, r
for kind, name in ((P.KIND_ABS, "abs"), (P.KIND_REL, "rel"), (P.KIND_FUND, "fund"), (P.KIND_HOM, "hom")):
    for n in (300, 1500, 2750, 5000, 10000):
        if kind == P.KIND_ABS:
            d = synth.absolute_pose_scene(n, 0.5, 77); a, b = (d["p2d"] - 500.0) / 1000.0, d["p3d"]
            q = d["q_gt"] + 0.003 * rs.randn(4); m0 = P.CameraPose(q / np.linalg.norm(q), d["t_gt"] + 0.003 * rs.randn(3)); thr = 12 / 1000.0
        else:
            gen = {P.KIND_REL: synth.relative_pose_scene, P.KIND_FUND: synth.fundamental_scene, P.KIND_HOM: synth.homography_scene}[kind]
            d = gen(n, 0.5, 78); a, b = (d["x1"] - 500.0) / 1000.0, (d["x2"] - 500.0) / 1000.0; thr = 1 / 1000.0
        pr = P.Problem(kind, a, b)
        # a decent start: the RANSAC result of a short run, then perturbed by re-running the refinement from it
        m0 = pr.run({"max_error": thr, "ransac": {"max_iterations": 2000, "min_iterations": 2000, "seed": 3}})[0] if kind != P.KIND_ABS else m0
        out = []
        for mi in (2, 40):
            bo = dict(loss_type=loss, loss_scale=thr, max_iterations=mi, gradient_tol=0.0, step_tol=0.0, relative_cost_tol=0.0)
            dt, (m, it) = timed(lambda: pr.refine(m0, bo))
            out.append((dt, it))
        (t1, i1), (t2, i2) = out
        per = (t2 - t1) / max(1, i2 - i1) * 1e6
        print(f"{name:4s} n={n:6d} {loss.lower():9s}: {t1*1e6:7.1f} us @ {i1} it, {t2*1e6:8.1f} us @ {i2} it -> {per:6.1f} us per LM iteration", flush=True)
        pr.close()
