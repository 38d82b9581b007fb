#!/usr/bin/env python
"""Wall time of pl_bundle_adjust_camera (k_lm_cam) per LM iteration for a few sizes (diagnostic)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P
from poselib_amd import synth
rs = np.random.RandomState(0)
for n in (256, 1500, 6000, 20000):
    d = synth.absolute_pose_scene(n, 0.0, 99)
    f, cx, cy = d["camera"]["params"]
    cam0 = dict(d["camera"], params=[f * 1.02, cx + 3, cy - 2])
    q = d["q_gt"] + 0.002 * rs.randn(4)
    p0 = P.CameraPose(q / np.linalg.norm(q), d["t_gt"] + 0.002 * rs.randn(3))
    pr = P.Problem(P.KIND_ABS, d["p2d"], d["p3d"])
    for flags, name in (({"refine_focal_length": True}, "K=7"), ({"refine_focal_length": True, "refine_principal_point": True}, "K=9")):
        bo = dict(flags, loss_type="CAUCHY", loss_scale=1.0)
        pr.bundle_adjust(p0, cam0, bo)
        t0 = time.perf_counter()
        for _ in range(20):
            pose, cam, it = pr.bundle_adjust(p0, cam0, bo)
        dt = (time.perf_counter() - t0) / 20
        pose2, it2 = pr.refine(p0, dict(loss_type="CAUCHY", loss_scale=1.0), camera=cam0)
        t0 = time.perf_counter()
        for _ in range(20):
            pr.refine(p0, dict(loss_type="CAUCHY", loss_scale=1.0), camera=cam0)
        dt2 = (time.perf_counter() - t0) / 20
        print(f"n={n:6d} {name}: k_lm_cam {dt*1e6:8.1f} us per call, {it} iterations -> {dt*1e6/max(it,1):7.1f} us per iteration | pose-only k_lm: {dt2*1e6:8.1f} us, {it2} iterations -> {dt2*1e6/max(it2,1):6.1f} us per iteration")
    pr.close()
