"""Time of the two focal-length minimal solvers on the device: pl_solve_focal_batch on `count` explicit minimal problems (one launch
sequence: setup + solve kernels, inputs uploaded and results downloaded inside the call).
    POSELIB_AMD_LIB=<build> python scripts/focal_solver_bench.py [count=32768]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rng = np.random.default_rng(3)
d = synth.absolute_pose_scene(4000, 0.0, 77, noise_px=0.5)
f, cx, cy = d["camera"]["params"]
x = np.asarray(d["p2d"]) - [cx, cy]
X = np.asarray(d["p3d"])
idx = np.array([rng.choice(4000, 4, replace=False) for _ in range(count)])
p35 = np.concatenate([x[idx].reshape(count, 8), X[idx].reshape(count, 12)], axis=1)
r = synth.relative_pose_scene(4000, 0.0, 78, noise_px=0.5)
fr, cx, cy = r["camera1"]["params"]


def unit(p):
    b = np.c_[(np.asarray(p) - [cx, cy]) / 800.0, np.ones(len(p))]
    return b / np.linalg.norm(b, axis=1)[:, None]


b1, b2 = unit(r["x1"]), unit(r["x2"])
idx = np.array([rng.choice(4000, 6, replace=False) for _ in range(count)])
six = np.concatenate([b1[idx].reshape(count, 18), b2[idx].reshape(count, 18)], axis=1)
for name, data in (("p35pf", p35), ("relpose_6pt_shared_focal", six)):
    P.solve_focal_batch(name, data)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        models, counts = P.solve_focal_batch(name, data)
        ts.append(time.perf_counter() - t)
    print(f"{name}: {count} minimal problems, {1e3 * np.median(ts):.2f} ms per call (min {1e3 * min(ts):.2f}), {counts.sum()} solutions", flush=True)
