"""How long is ONE k_score_mfma<10> launch (pl_ransac_stats.score_kernel_ms, HIP events) when single problems follow a grouped
warm-up?  Prints the per-problem kernel time of 40 single problems after 3 grouped steps, then after 2 s of idling."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import poselib_amd as P
from poselib_amd import synth

scs = [synth.absolute_pose_scene(5000, 0.7, 1001 + 7919 * k) for k in range(16)]
def norm(sc):
    par = np.asarray(sc["camera"]["params"], float)
    return (np.asarray(sc["p2d"], float) - par[-2:]) / par[0], np.asarray(sc["p3d"], float), par[0]
probs = []
for sc in scs:
    x, X, f = norm(sc)
    probs.append(P.Problem(P.KIND_ABS, x, X))
thr = 12.0 / f
opt = lambda seed: {"max_error": thr, "ransac": {"max_iterations": 100000, "min_iterations": 100000, "seed": seed}}
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = P.RansacBatch([probs[j % 16] for j in range(NB)], [opt(j) for j in range(NB)])
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    batch.run(8, 16)
def series(tag):
    out = []
    for j in range(40):
        _, info = probs[0].run(opt(7000 + j))
        out.append(info["score_kernel_ms"] / max(info["score_kernel_launches"], 1))
    print(tag, " ".join(f"{1e3 * v:.0f}" for v in out), "us")
series("after grouped steps:")
time.sleep(2.0)
series("after 2 s idle:     ")
batch.run(8, 16)
series("after one more step:")
