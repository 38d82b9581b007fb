"""configs[4] through pl_estimate_batch for several (host threads, group size) settings in ONE process:
    python scripts/batch_sweep.py 4096 8:0 16:128 24:128 ...      (threads:group, group 0 = the library's own choice)
Prints problems/s per setting and checks that every setting returns the same iterations / inliers (a problem's result does not
depend on its group)."""
import os, sys, time
if "--queues" in sys.argv:
    k = sys.argv.index("--queues"); os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[k + 1]; del sys.argv[k:k + 2]
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import poselib_amd as P
from poselib_amd import synth

n_prob = int(sys.argv[1])
settings = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(8, 0)]
kinds = ("abs", "rel", "hom")
problems = []
t0 = time.perf_counter()
for i in range(n_prob):
    rs = synth.Stream(900000 + i)
    n = int(rs.uniform(1, 500, 5001)[0]); outl = float(rs.uniform(1, 0.3, 0.7)[0]); kind = kinds[i % 3]
    opt = {"ransac": {"seed": i}}
    if kind == "abs":
        d = synth.absolute_pose_scene(n, outl, 2000 + i); problems.append(("abs", d["p2d"], d["p3d"], d["camera"], opt))
    elif kind == "rel":
        d = synth.relative_pose_scene(n, outl, 2000 + i); problems.append(("rel", d["x1"], d["x2"], d["camera1"], d["camera2"], opt))
    else:
        d = synth.homography_scene(n, outl, 2000 + i, noise_px=0.3); problems.append(("hom", d["x1"], d["x2"], opt))
print(f"{n_prob} problems generated in {time.perf_counter() - t0:.1f} s", flush=True)
batch = P.Batch(problems)
ref = None
for cfg in settings:
    threads, group = cfg[0], cfg[1]
    if len(cfg) > 2:
        os.environ["POSELIB_AMD_GROUP_STEPS"] = str(cfg[2])
    else:
        os.environ.pop("POSELIB_AMD_GROUP_STEPS", None)
    if group:
        os.environ["POSELIB_AMD_BATCH_GROUP"] = str(group)
    else:
        os.environ.pop("POSELIB_AMD_BATCH_GROUP", None)
    for _ in range(3):
        batch.run(max_in_flight=threads)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); batch.run(max_in_flight=threads); ts.append(time.perf_counter() - t)
    st = batch.stats()
    key = (np.asarray(st[0]).copy(), np.asarray(st[1]).copy(), np.asarray(st[2]).copy())
    same = "first" if ref is None else str(all(bool((a == b).all()) for a, b in zip(ref, key)))
    if ref is None:
        ref = key
    print(f"threads {threads:3d} group {group:4d} steps {cfg[2] if len(cfg) > 2 else -1:2d}: {n_prob / np.median(ts):9.0f} problems/s (median of 4: {1e3 * np.median(ts):.1f} ms, min {1e3 * min(ts):.1f}); same results: {same}", flush=True)
