"""How often does the device return -q where the oracle returns q (same rotation, the other sign of the quaternion)?
    python scripts/exp/quat_sign_soak.py [problems per kind=400] [seed=1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import poselib_amd as P
from poselib_amd import synth
total = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for kind in ("abs", "rel"):
    flips = same = other = 0
    for i in range(total):
        n = int(rng.integers(300, 4000)); outl = float(rng.uniform(0.1, 0.6)); opt = {"ransac": {"seed": int(rng.integers(1 << 30))}}
        if kind == "abs":
            d = synth.absolute_pose_scene(n, outl, 70000 + i)
            m, info = P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
            om, omask, ost = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
            m = m.pose
        else:
            d = synth.relative_pose_scene(n, outl, 70000 + i)
            m, info = P.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
            om, omask, ost = O.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
        qa, qb = np.asarray(m.q), np.asarray(om[:4])
        if np.abs(qa - qb).max() < 1e-6:
            same += 1
        elif np.abs(qa + qb).max() < 1e-6:
            flips += 1
            print("FLIP", kind, i, n, info["iterations"], ost["iterations"], info["refinements"], ost["refinements"])
        else:
            other += 1
    print(f"{kind}: {total} problems: same sign {same}, opposite sign {flips}, different rotation {other}")
