"""Reproduce item 1186 of tests/parity_soak_estimate_batch.py 4000 11 (a PROSAC homography problem whose device model was -1 x the
oracle's): the device's and the oracle's H, their ratio, under the default LM sums and the reference-order ones."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import poselib_amd as P
from poselib_amd import synth
rng = np.random.default_rng(11)
target = 1186
for i in range(target + 1):
    kind = ["abs", "rel", "fund", "hom"][int(rng.integers(4))]
    n = int(rng.integers(12, 3000)); outl = float(rng.uniform(0.1, 0.6))
    feat = ["plain", "prosac", "warm", "opencv", "prosac+warm"][int(rng.integers(5))]
    ro = {"seed": int(rng.integers(1 << 30))}
    if "prosac" in feat:
        ro["progressive_sampling"] = True
        if rng.random() < 0.4:
            ro["max_prosac_iterations"] = int(rng.integers(10, 400))
    if "warm" in feat and kind in ("abs", "rel"):
        rng.normal(size=4)
        if kind == "abs":
            rng.normal(size=3)
print(i, kind, feat, n, outl, ro)
d = synth.homography_scene(n, outl, 60000 + i)
order = np.argsort(~d["inlier_gt"], kind="stable")
x1, x2 = np.asarray(d["x1"])[order], np.asarray(d["x2"])[order]
opt = {"ransac": ro}
oh, omask, ost = O.estimate_homography(x1, x2, opt)
for mode in (0, 1):
    P.set_lm_mode(mode)
    H, info = P.estimate_homography(x1, x2, opt)
    H = np.asarray(H)
    print("lm mode", mode, "iterations", info["iterations"], ost["iterations"], "refinements", info["refinements"], ost["refinements"], "inliers", info["num_inliers"], ost["num_inliers"])
    print(" device H:", H.ravel()); print(" oracle H:", np.asarray(oh).ravel()); print(" ratio:", (H / np.asarray(oh)).ravel())
P.set_lm_mode(0)
(res,) = P.estimate_batch([("hom", x1, x2, opt)], max_in_flight=1)
print("batch H:", np.asarray(res[0]).ravel())
