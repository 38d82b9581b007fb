import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import poselib_amd as P
from poselib_amd import synth
for n in (8, 11, 24):
    for its in (1000, 5000, 30000):
        opt = {"ransac": {"seed": 5 + n, "max_iterations": its, "min_iterations": its}}
        d = synth.absolute_pose_scene(max(n, 8), 0.25, 71 + n)
        img, info = P.estimate_absolute_pose(d["p2d"][:n], d["p3d"][:n], d["camera"], opt)
        pose, mask, st = O.estimate_absolute_pose(d["p2d"][:n], d["p3d"][:n], d["camera"], opt)
        print(n, its, {k: info[k] for k in ("iterations", "refinements", "num_inliers", "model_score")},
              {k: st[k] for k in ("iterations", "refinements", "num_inliers", "model_score")}, (np.array(info["inliers"]) == mask).all())
