"""GPU soak: the focal-length estimators through pl_estimate_batch (lock-step groups, driver_focal_group.inc; launches of >= 4096
samples through the three-kernel solve stage with the packed eigenvalue iteration) against the single-problem entry points (one-kernel
solve stage, one matrix per wavefront) on random problems with random options - every field of every result, bit for bit.

    python scripts/soak_focal_group.py [problems per estimator=600] > profiles/r05_soak_focal_group.md
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402

STAT_KEYS = ("iterations", "refinements", "num_inliers", "hypotheses", "model_score", "inlier_ratio")


def make(name, k, rng):
    n = int(rng.choice([rng.integers(8, 40), rng.integers(40, 400), rng.integers(400, 3000), rng.integers(3000, 9000)], p=[0.1, 0.3, 0.5, 0.1]))
    outl = float(rng.uniform(0.0, 0.65))
    focal = float(rng.uniform(400, 3000))
    noise = float(rng.uniform(0.05, 2.0))
    ransac = {"seed": int(rng.integers(0, 2**31))}
    r = rng.random()
    if r < 0.2:
        ransac.update(min_iterations=int(rng.integers(10, 400)), max_iterations=int(rng.integers(400, 3000)))
    elif r < 0.35:
        ransac.update(min_iterations=int(rng.integers(1000, 4000)))
    elif r < 0.45:
        ransac.update(max_iterations=int(rng.integers(1, 300)), min_iterations=0)
    if rng.random() < 0.3:
        ransac.update(success_prob=float(rng.choice([0.9, 0.99, 0.999999])), dyn_num_trials_mult=float(rng.choice([1.0, 3.0, 10.0])))
    if name == "pnpf":
        d = synth.absolute_pose_scene(n, outl, 41000 + k, focal=focal, noise_px=noise)
        f, cx, cy = d["camera"]["params"]
        off = float(rng.uniform(0.6, 1.6))
        if rng.random() < 0.5:
            cam = dict(d["camera"], params=[off * f, cx, cy])
        else:
            cam = dict(d["camera"], model="PINHOLE", params=[off * f, off * f * float(rng.uniform(0.98, 1.02)), cx, cy])
        opt = {"max_error": float(rng.uniform(1.0, 12.0)), "estimate_focal_length": True, "ransac": ransac}
        if rng.random() < 0.3:
            opt["min_fov"] = float(rng.uniform(5.0, 60.0))
        if rng.random() < 0.2:
            opt["bundle"] = {"loss_type": str(rng.choice(["CAUCHY", "HUBER", "TRIVIAL", "TRUNCATED"])), "loss_scale": float(rng.uniform(0.5, 4.0))}
        return ("abs", d["p2d"], d["p3d"], cam, opt)
    d = synth.relative_pose_scene(n, outl, 42000 + k, focal=focal, noise_px=noise)
    pp = [float(x) + float(rng.uniform(-20, 20)) for x in d["camera1"]["params"][1:3]]
    opt = {"max_error": float(rng.uniform(0.5, 4.0)), "ransac": ransac}
    if rng.random() < 0.2:
        opt["bundle"] = {"loss_type": str(rng.choice(["CAUCHY", "HUBER", "TRIVIAL", "TRUNCATED"])), "loss_scale": float(rng.uniform(0.5, 4.0))}
    return ("shared_focal", d["x1"], d["x2"], pp, opt)


def single(pr):
    if pr[0] == "abs":
        return P.estimate_absolute_pose(pr[1], pr[2], pr[3], pr[4])
    return P.estimate_shared_focal_relative_pose(pr[1], pr[2], pr[3], pr[4])


def same(pr, a, b):
    (m1, i1), (m2, i2) = a, b
    if any(i1[k] != i2[k] for k in STAT_KEYS) or not np.array_equal(np.asarray(i1["inliers"]), np.asarray(i2["inliers"])):
        return False
    if not np.array_equal(np.r_[m1.pose.q, m1.pose.t], np.r_[m2.pose.q, m2.pose.t]):
        return False
    c1, c2 = (m1.camera, m2.camera) if pr[0] == "abs" else (m1.camera1, m2.camera1)
    return list(c1.params) == list(c2.params)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rng = np.random.default_rng(5)
    print("# r05 - soak: focal-length problems through `pl_estimate_batch` (lock-step groups) against their single calls (`scripts/soak_focal_group.py`)\n")
    print("Random problems (8 ... 9000 correspondences, 0 - 65 % outliers, focal lengths 400 - 3000 coming in 0.6 - 1.6 x off, both pinhole models, random thresholds, `min_fov`,")
    print("iteration bounds, success probabilities, bundle losses) in calls of random size and worker count; every field of every result (pose, camera, mask, statistics)")
    print("compared bit for bit.  The group path runs launches of >= 4096 samples through the three-kernel solve stage (four matrices per wavefront in the eigenvalue")
    print("kernel), the single calls the one-kernel form.\n")
    print("| estimator | problems | calls | identical to the single call | in a lock-step group (the rest: PROSAC-free items the group path leaves to the single path - fewer than sample + 4 points, long fixed runs) | seconds: batch calls / single calls |")
    print("|---|---|---|---|---|---|")
    for name in ("pnpf", "shared_focal"):
        problems = [make(name, k, rng) for k in range(count)]
        got, calls, t_b = [], 0, 0.0
        at = 0
        while at < count:
            sz = int(rng.choice([1, 3, 17, 64, 200]))
            chunk = problems[at:at + sz]
            b = P.Batch(chunk)
            t0 = time.perf_counter()
            b.run(int(rng.choice([1, 2, 8])))
            t_b += time.perf_counter() - t0
            got += b.results()
            at += len(chunk)
            calls += 1
        t0 = time.perf_counter()
        ref = [single(pr) for pr in problems]
        t_s = time.perf_counter() - t0
        good = sum(same(pr, g, r) for pr, g, r in zip(problems, got, ref))
        K = 4 if name == "pnpf" else 6
        grouped = sum(1 for pr in problems if len(pr[1]) >= K + 4 and len(pr[1]) <= 16384 and pr[4]["ransac"].get("min_iterations", 1000) <= 4096
                      and pr[4]["ransac"].get("max_iterations", 100000) > 0)
        print(f"| {name} | {count} | {calls} | **{good}** | {grouped} | {t_b:.1f} / {t_s:.1f} |", flush=True)
        if good != count:
            for k, (pr, g, r) in enumerate(zip(problems, got, ref)):
                if not same(pr, g, r):
                    print(f"\n* MISMATCH {name} k={k} n={len(pr[1])} opt={pr[4]}: batch {[g[1][s] for s in STAT_KEYS]} single {[r[1][s] for s in STAT_KEYS]}")
    print()


if __name__ == "__main__":
    main()
