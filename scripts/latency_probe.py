"""Single-problem latency of the one-shot front-ends vs resident problems (default options, ~1000 iterations)."""
import time, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poselib_amd as P
from poselib_amd import synth

def t(fn, n=30):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3

for N in (500, 2000, 5000):
    d = synth.absolute_pose_scene(N, 0.5, 5)
    opt = {"ransac": {"seed": 1}}
    one = t(lambda: P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt))
    par = d["camera"]["params"]
    x = (d["p2d"] - par[-2:]) / par[0]
    create = t(lambda: P.Problem(P.KIND_ABS, x, d["p3d"]).close())
    pr = P.Problem(P.KIND_ABS, x, d["p3d"])
    o2 = {"max_error": 12.0 / par[0], "ransac": {"seed": 1}}
    run = t(lambda: pr.run(o2))
    info = pr.run(o2)[1]
    h = synth.homography_scene(N, 0.5, 6)
    hom = t(lambda: P.estimate_homography(h["x1"], h["x2"], opt))
    r = synth.relative_pose_scene(N, 0.5, 7)
    rel = t(lambda: P.estimate_relative_pose(r["x1"], r["x2"], r["camera1"], r["camera2"], opt))
    print(f"N={N}: estimate_absolute_pose {one:.2f} ms | Problem create+destroy {create:.2f} ms | resident run {run:.2f} ms "
          f"(iterations {info['iterations']}, refinements {info['refinements']}) | estimate_homography {hom:.2f} ms | "
          f"estimate_relative_pose {rel:.2f} ms")
