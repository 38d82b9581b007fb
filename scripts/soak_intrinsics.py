#!/usr/bin/env python
"""Randomised soak of the intrinsics-refining bundle (k_lm_cam) against the oracle: random sizes, camera models, flag sets,
losses, start errors, outliers and masks.  Problems of up to 256 correspondences must be bit-identical; beyond, pose / camera
to 1e-8 (relative for the camera).  usage: soak_intrinsics.py [count] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LOSSES = ["TRIVIAL", "TRUNCATED", "HUBER", "CAUCHY", "TRUNCATED_CAUCHY", "TRUNCATED_LE_ZACH"]
bad, exact, worst_pose, worst_cam, it_diff = [], 0, 0.0, 0.0, 0
for k in range(count):
    n = int(rs.choice([rs.randint(4, 20), rs.randint(20, 257), rs.randint(257, 4000)]))
    d = synth.absolute_pose_scene(n, float(rs.choice([0.0, 0.2, 0.5])), 7000 + k)
    f, cx, cy = d["camera"]["params"]
    pix = np.asarray(d["p2d"])
    model = ["SIMPLE_PINHOLE", "PINHOLE", "OPENCV"][k % 3]
    if model == "SIMPLE_PINHOLE":
        par = [f, cx, cy]
    elif model == "PINHOLE":
        par = [f, f * (1 + 0.01 * rs.randn()), cx, cy]
    else:
        par = [f, f, cx, cy, -0.05 * rs.rand(), 0.01 * rs.randn(), 1e-3 * rs.randn(), 5e-4 * rs.randn()]
        pix = synth.opencv_distort_pixels(pix, par)
    nf = 1 if model == "SIMPLE_PINHOLE" else 2
    off = np.array(par)
    off[:nf] *= 1 + 0.03 * rs.randn(nf)
    off[nf:nf + 2] += 5 * rs.randn(2)
    cam0 = {"model": model, "width": 1000, "height": 1000, "params": [float(v) for v in off]}
    q = d["q_gt"] + 0.005 * rs.randn(4)
    p0 = np.r_[q / np.linalg.norm(q), d["t_gt"] + 0.005 * rs.randn(3)]
    flags = {"refine_focal_length": bool(rs.rand() < 0.8), "refine_principal_point": bool(rs.rand() < 0.5),
             "refine_extra_params": bool(rs.rand() < 0.5)}
    if not any(flags.values()):
        flags["refine_focal_length"] = True
    bo = dict(flags, loss_type=LOSSES[int(rs.randint(6))], loss_scale=float(rs.choice([0.5, 2.0, 8.0])),
              max_iterations=int(rs.choice([5, 25, 100])), lambda_update=int(rs.randint(2)), damping=int(rs.randint(2)))
    mask = (rs.rand(n) < 0.7) if rs.rand() < 0.4 else None
    sel = slice(None) if mask is None else mask
    if mask is not None and mask.sum() < 3:
        mask, sel = None, slice(None)
    rp, rc, st = O.bundle_adjust_camera(pix[sel], d["p3d"][sel], cam0, p0, bo)
    pr = P.Problem(P.KIND_ABS, pix, d["p3d"])
    pose, cam, it = pr.bundle_adjust(P.CameraPose(p0[:4], p0[4:]), cam0, bo, mask=mask)
    pr.close()
    got = np.r_[pose.q, pose.t]
    gc = np.asarray(cam.params)
    m = n if mask is None else int(mask.sum())
    same = np.array_equal(got, rp, equal_nan=True) and np.array_equal(gc, rc, equal_nan=True) and it == st.iterations
    exact += same
    if n <= 256:  # (the kernel sums the robust cost in order when the PROBLEM has at most 256 correspondences)
        if not same:
            bad.append((k, n, m, model, bo, it, st.iterations, float(np.nanmax(np.abs(got - rp))), float(np.nanmax(np.abs(gc - rc)))))
    else:
        dp = float(np.abs(got - rp).max())
        dc = float(np.abs(gc - rc).max() / max(1.0, np.abs(rc).max()))
        worst_pose, worst_cam = max(worst_pose, dp), max(worst_cam, dc)
        it_diff += it != st.iterations
        if not (dp < 1e-8 and dc < 1e-8):
            bad.append((k, n, m, model, bo, it, st.iterations, dp, dc))
print(f"{count} problems: {exact} bit-identical (pose, camera, iterations); above 256 correspondences: worst |dpose| {worst_pose:.2e}, "
      f"worst rel |dcamera| {worst_cam:.2e}, {it_diff} with another iteration count; disagreements: {len(bad)}")
for b in bad[:10]:
    print("  ", b)
sys.exit(1 if bad else 0)
