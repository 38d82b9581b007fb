"""configs[4] at 512 problems per call (what one of eight ranks sees of a 4096-problem batch): one call at a time against
K calls in flight from K host threads (double buffering of consecutive batches - pl_estimate_batch is re-entrant).
    python scripts/batch_overlap.py [problems_per_call=512] [threads_per_call=10]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import poselib_amd as P
from poselib_amd import synth

n_call = int(sys.argv[1]) if len(sys.argv) > 1 else 512
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kinds = ("abs", "rel", "hom")

def make(i):
    rs = synth.Stream(900000 + i)
    n = int(rs.uniform(1, 500, 5001)[0]); outl = float(rs.uniform(1, 0.3, 0.7)[0]); kind = kinds[i % 3]
    opt = {"ransac": {"seed": i}}
    if kind == "abs":
        d = synth.absolute_pose_scene(n, outl, 2000 + i); return ("abs", d["p2d"], d["p3d"], d["camera"], opt)
    if kind == "rel":
        d = synth.relative_pose_scene(n, outl, 2000 + i); return ("rel", d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
    d = synth.homography_scene(n, outl, 2000 + i, noise_px=0.3); return ("hom", d["x1"], d["x2"], opt)

K = 4
batches = [P.Batch([make(k * n_call + j) for j in range(n_call)]) for k in range(K)]
for b in batches:
    for _ in range(3):
        b.run(max_in_flight=threads)
ref = [tuple(np.asarray(x).copy() for x in b.stats()[:3]) for b in batches]
for inflight in (1, 2, 3, 4):
    wk = max(2, threads * 2 // (inflight + 1)) if inflight > 1 else threads
    def worker(k, reps):
        for _ in range(reps):
            batches[k].run(max_in_flight=wk)
    for reps in (4, 12):  # (an untimed round first: the pools' workers grow their arenas)
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(k, reps)) for k in range(inflight)]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
    same = all(all(bool((a == np.asarray(b)).all()) for a, b in zip(ref[k], batches[k].stats()[:3])) for k in range(inflight))
    print(f"{n_call} problems per call, {inflight} call(s) in flight x {wk} workers: {inflight * reps * n_call / dt:9.0f} problems/s"
          f" ({1e3 * dt / reps:.1f} ms per round of calls); same results: {same}", flush=True)
