#!/usr/bin/env python
"""Cycle breakdown of one LM iteration of k_lm (experiment build with -DPL_LM_PROFILE: scripts/exp/variants/lmprof/libposelib_amd.so):
thread 0's cycles per phase, per iteration, for a refinement task that runs alone on the device.
    python scripts/lm_profile.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poselib_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "scripts", "exp", "variants", os.environ.get("LM_PROFILE_VARIANT", "lmprof"), "libposelib_amd.so")
import poselib_amd as P
from poselib_amd import synth
L = _lib.lib()
prof = L.pl_debug_lm_profile
prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
rs = np.random.RandomState(0)
names = ["prepare+barrier", "sweep (thread 0)", "wait for slowest wave", "block reductions", "lm_solve", "solve+step+barrier", "update+copy+barriers"]
print("| estimator, n, loss | iterations | " + " | ".join(names) + " | sum of phases per iteration | passes per iteration |")
print("|---|---|" + "---|" * (len(names) + 2))
for kind, name in ((P.KIND_ABS, "abs"), (P.KIND_REL, "rel"), (P.KIND_HOM, "hom")):
    for n in ([int(x) for x in os.environ["LM_PROFILE_N"].split(",")] if os.environ.get("LM_PROFILE_N") else (300, 1500, 5000, 10000)):
        for loss in ("TRUNCATED", "CAUCHY"):
            if kind == P.KIND_ABS:
                d = synth.absolute_pose_scene(n, 0.5, 77); a, b = (d["p2d"] - 500.0) / 1000.0, d["p3d"]
                q = d["q_gt"] + 0.003 * rs.randn(4); m0 = P.CameraPose(q / np.linalg.norm(q), d["t_gt"] + 0.003 * rs.randn(3)); thr = 12 / 1000.0
            else:
                gen = {P.KIND_REL: synth.relative_pose_scene, P.KIND_HOM: synth.homography_scene}[kind]
                d = gen(n, 0.5, 78); a, b = (d["x1"] - 500.0) / 1000.0, (d["x2"] - 500.0) / 1000.0; thr = 1 / 1000.0
            pr = P.Problem(kind, a, b)
            if kind != P.KIND_ABS:
                m0 = pr.run({"max_error": thr, "ransac": {"max_iterations": 2000, "min_iterations": 2000, "seed": 3}})[0]
            bo = dict(loss_type=loss, loss_scale=thr, max_iterations=40, gradient_tol=0.0, step_tol=0.0, relative_cost_tol=0.0)
            pr.refine(m0, bo)
            buf = (C.c_ulonglong * 16)()
            prof(buf, 1)
            for _ in range(5):
                pr.refine(m0, bo)
            prof(buf, 1)
            v = list(buf)
            its = max(1, v[9])
            cells = [f"{v[i] / its:.0f}" for i in range(7)]
            tot = (v[0] + v[1] + v[2] + v[3] + v[5] + v[6]) / its
            print(f"| {name} n={n} {loss.lower()} | {its // 5} | " + " | ".join(cells) + f" | {tot:.0f} | {v[8] / its:.2f} |", flush=True)
            pr.close()
