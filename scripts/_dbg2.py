import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import poselib_amd as P
from poselib_amd import synth
d = synth.fundamental_scene(2000, 0.4, 77)
opt = {"ransac": {"seed": 3}}
F, info = P.estimate_fundamental(d["x1"], d["x2"], opt)
Fo, mask, st = O.estimate_fundamental(d["x1"], d["x2"], opt)
print(os.environ.get("MODE"), {k: info[k] for k in ("iterations", "refinements", "num_inliers", "model_score")}, {k: st[k] for k in ("iterations", "refinements", "num_inliers", "model_score")})
