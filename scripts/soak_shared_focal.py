"""Soak (CPU): the oracle's shared-focal relative pose estimator against the reference sources (oracle/_ref) on random problems."""
import sys
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, oracle_lib as O, ref_lib
from poselib_amd import synth
rng = np.random.default_rng(0)
same=0; tot=0; maskd=0; fd=[]
for k in range(120):
    n = int(rng.integers(50, 2500)); outl = rng.uniform(0.05, 0.6)
    d = synth.relative_pose_scene(n, outl, 9000+k, focal=float(rng.uniform(500, 2500)), noise_px=float(rng.uniform(0.1,1.5)))
    f, cx, cy = d["camera1"]["params"]
    opt = {"max_error": float(rng.uniform(1,3)), "ransac": {"seed": k, "max_iterations": 5000}}
    po, fo, mo, so = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], opt)
    with ref_lib.reference():
        pr, fr, mr, sr = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], opt)
    tot+=1
    eq = so["iterations"]==sr["iterations"] and so["refinements"]==sr["refinements"] and (mo==mr).all()
    same += eq
    fd.append(abs(fo-fr)/max(fr,1e-9))
    if not eq: print(k, n, round(outl,2), "iters", so["iterations"], sr["iterations"], "ref", so["refinements"], sr["refinements"], "inl", so["num_inliers"], sr["num_inliers"], "focal", fo, fr, f)
print("identical decisions:", same, "of", tot, "; focal rel diff median %.1e max %.1e" % (np.median(fd), np.max(fd)))
