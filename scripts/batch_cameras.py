"""pl_estimate_batch on absolute-pose problems whose pixels come from an OPENCV camera against the same scenes through a
SIMPLE_PINHOLE camera (OPENCV absolute pose inside the lock-step groups - within 15 % of the pinhole rate),
and the same batch with PROSAC / with warm starts; pl_last_batch_report shows where the items went.
    python scripts/batch_cameras.py [problems=1024]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
OCV = {"model": "OPENCV", "width": 1000, "height": 1000, "params": [1000.0, 1000.0, 500.0, 500.0, 0.01, -0.002, 1e-4, -1e-4]}


def distort(p2d, cam):
    f, cx, cy = cam["params"]
    xn = (np.asarray(p2d) - [cx, cy]) / f
    r2 = (xn ** 2).sum(1)
    k1, k2, p1, p2 = OCV["params"][4:]
    rad = 1 + k1 * r2 + k2 * r2 ** 2
    xd = np.c_[xn[:, 0] * rad + 2 * p1 * xn[:, 0] * xn[:, 1] + p2 * (r2 + 2 * xn[:, 0] ** 2),
               xn[:, 1] * rad + p1 * (r2 + 2 * xn[:, 1] ** 2) + 2 * p2 * xn[:, 0] * xn[:, 1]]
    return xd * OCV["params"][:2] + OCV["params"][2:4]


scenes = []
for i in range(count):
    rs = synth.Stream(910000 + i)
    n = int(rs.uniform(1, 500, 5001)[0])
    outl = float(rs.uniform(1, 0.3, 0.7)[0])
    scenes.append(synth.absolute_pose_scene(n, outl, 5000 + i))


def rate(name, probs):
    b = P.Batch(probs)
    for _ in range(3):
        b.run(max_in_flight=10)
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        b.run(max_in_flight=10)
        ts.append(time.perf_counter() - t)
    rep = P.last_batch_report()
    print(f"| {name} | {count / np.median(ts):.0f} | {1e3 * np.median(ts):.1f} | {rep['grouped']} | {rep['solo']} | {rep['fallback']} |", flush=True)


print(f"| {count} absolute-pose problems per call (N ~ U[500, 5000], 30 - 70 % outliers, default options) | problems/s | ms per call | grouped | solo | fallback |")
print("|---|---|---|---|---|---|")
rate("SIMPLE_PINHOLE", [("abs", d["p2d"], d["p3d"], d["camera"], {"ransac": {"seed": i}}) for i, d in enumerate(scenes)])
rate("OPENCV (same rays)", [("abs", distort(d["p2d"], d["camera"]), d["p3d"], OCV, {"ransac": {"seed": i}}) for i, d in enumerate(scenes)])
prosac = []
for i, d in enumerate(scenes):
    order = np.argsort(~d["inlier_gt"], kind="stable")
    prosac.append(("abs", np.asarray(d["p2d"])[order], np.asarray(d["p3d"])[order], d["camera"], {"ransac": {"seed": i, "progressive_sampling": True}}))
rate("SIMPLE_PINHOLE, PROSAC", prosac)
rate("SIMPLE_PINHOLE, warm start (true pose)",
     [("abs", d["p2d"], d["p3d"], d["camera"], {"ransac": {"seed": i}, "initial_model": P.CameraPose(d["q_gt"], d["t_gt"])}) for i, d in enumerate(scenes)])
