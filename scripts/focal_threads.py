#!/usr/bin/env python
"""Throughput of the two focal-length front-ends from T host threads (one HIP stream and context each, problems of 2000
correspondences, 40 % outliers, default options): python scripts/focal_threads.py [T ...]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P
from poselib_amd import synth
n = 2000
da = [synth.absolute_pose_scene(n, 0.4, 7100 + k) for k in range(4)]
dr = [synth.relative_pose_scene(n, 0.4, 7000 + k) for k in range(4)]
pp = lambda d: d["camera1"]["params"][1:3]
def run_abs(j):
    d = da[j % 4]
    return P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": j}})
def run_rel(j):
    d = dr[j % 4]
    return P.estimate_shared_focal_relative_pose(d["x1"], d["x2"], pp(d), {"max_error": 2.0, "ransac": {"seed": j}})
import threading
for T in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32]:
    for name, run in (("pnpf_2000", run_abs), ("shared_focal_2000", run_rel)):
        N = max(64, 32 * T)
        nxt, lock, bar = [0], threading.Lock(), threading.Barrier(T + 1)
        def work(i):
            run(i), run(i + 1)  # every thread's context, stream and buffers exist before the clock starts
            bar.wait()
            while True:
                with lock:
                    j = nxt[0]; nxt[0] += 1
                if j >= N:
                    break
                run(j)
        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        for t in th: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print(f"{name:18s} {T:3d} threads: {N / dt:8.0f} problems/s ({1e3 * dt / N * T:6.2f} ms per problem and thread)", flush=True)
# the same problems as items of ONE pl_estimate_batch call (the library's own worker threads, one HIP stream each)
for T in (8, 16):
    for name, mk in (("pnpf_2000", lambda j: ("abs", da[j % 4]["p2d"], da[j % 4]["p3d"], da[j % 4]["camera"], {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": j}})),
                     ("shared_focal_2000", lambda j: ("shared_focal", dr[j % 4]["x1"], dr[j % 4]["x2"], pp(dr[j % 4]), {"max_error": 2.0, "ransac": {"seed": j}}))):
        b = P.Batch([mk(j) for j in range(512)])
        b.run(T)
        t0 = time.perf_counter()
        b.run(T)
        dt = time.perf_counter() - t0
        print(f"{name:18s} pl_estimate_batch, 512 items, {T:2d} workers: {512 / dt:8.0f} problems/s", flush=True)
