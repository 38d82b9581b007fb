"""The headline scorer alone: P3P hypotheses of random minimal samples of a 5000-correspondence scene (70 % outliers) through
pl_debug_score_stream, N times; run under `rocprofv3 --kernel-trace --stats` (or PMC passes) to read k_score_mfma<10>'s duration.
    python scripts/exp/score_stream_bench.py [hypotheses] [repeats]
Prints the wall time per call and a checksum of the counts (variants of the kernel must agree - unless they are timing-only
experiment builds)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import poselib_amd as P
from poselib_amd import synth

H = int(sys.argv[1]) if len(sys.argv) > 1 else 129425
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = synth.absolute_pose_scene(5000, 0.7, 4242)
cam = d["camera"]
par = np.asarray(cam["params"], float)
x = (np.asarray(d["p2d"], float) - par[-2:]) / par[0]
X = np.asarray(d["p3d"], float)
rs = np.random.RandomState(5)
B = int(H / 1.25) + 1000
idx = np.stack([rs.permutation(5000)[:3] for _ in range(B)])
bear = np.concatenate([x[idx], np.ones((B, 3, 1))], axis=2)
bear /= np.linalg.norm(bear, axis=2, keepdims=True)
rec, cnt = P.solve_batch(0, bear, X[idx])
models = np.concatenate([rec[b, :cnt[b], :7] for b in range(B)])
models = models[np.isfinite(models).all(1)][:H]
prob = P.Problem(P.KIND_ABS, x, X)
thr = 12.0 / par[0]
c0, s0, path = prob.score_stream(models, thr)
t = time.perf_counter()
for _ in range(reps):
    c, s, path = prob.score_stream(models, thr)
dt = (time.perf_counter() - t) / reps
print(f"{len(models)} hypotheses, path {path}: {1e3 * dt:.3f} ms per call (wall, copies included); inliers: sum {int(c.sum())}, max {int(c.max())}, checksum {int((c.astype(np.uint64) * (1 + np.arange(len(c), dtype=np.uint64) % 97)).sum())}")
