"""One estimator through pl_estimate_batch a few times (for rocprofv3 --kernel-trace --stats): python scripts/focal_batch_trace.py pnpf|shared_focal [problems] [n] [calls] [workers]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402

name = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
workers = int(sys.argv[5]) if len(sys.argv) > 5 else 8
da = [synth.absolute_pose_scene(n, 0.4, 7100 + k) for k in range(4)]
dr = [synth.relative_pose_scene(n, 0.4, 7000 + k) for k in range(4)]


def item(j):
    if name == "pnpf":
        d = da[j % 4]
        return ("abs", d["p2d"], d["p3d"], d["camera"], {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": j}})
    d = dr[j % 4]
    return ("shared_focal", d["x1"], d["x2"], d["camera1"]["params"][1:3], {"max_error": 2.0, "ransac": {"seed": j}})


for k in range(calls):
    b = P.Batch([item(j) for j in range(count)])
    t0 = time.perf_counter()
    b.run(workers)
    dt = time.perf_counter() - t0
    print(f"call {k}: {count / dt:.0f} problems/s ({dt * 1e3:.1f} ms)", flush=True)
st = [r[1] for r in b.results()]
print("mean iterations", sum(s["iterations"] for s in st) / count, "refinements", sum(s["refinements"] for s in st) / count, "hypotheses", sum(s["hypotheses"] for s in st) / count)
