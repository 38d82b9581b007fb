"""GPU soak of the two focal-length estimators: random problems through the C-ABI against the oracle.
    python scripts/soak_focal_gpu.py [problems per estimator] > profiles/r03_soak_focal_estimators.md
Demanded: iterations, refinements, inlier count and mask identical; pose and focal length bit for bit for both estimators at every
size (their kernels add in correspondence order; k_lm_cam's cost as well since round 4).  A third of the problems samples with
PROSAC, a third of the pnpf problems sets min_fov.    python scripts/soak_focal_gpu.py 300 20260925"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260924)
    rows = []
    for name in ("shared_focal", "pnpf"):
        bad, bitwise, t_dev, t_cpu, worst = 0, 0, 0.0, 0.0, 0.0
        for k in range(count):
            n = int(rng.choice([7, 12, 40, 120, 256, 257, 600, 1500, 4000]))
            outl = float(rng.uniform(0.0, 0.65))
            noise = float(rng.uniform(0.1, 1.5))
            focal = float(rng.uniform(400, 3000))
            ro = {"seed": int(rng.integers(0, 2 ** 31)), "max_iterations": int(rng.choice([200, 1000, 5000, 100000])),
                  "min_iterations": int(rng.choice([10, 100, 1000])), "success_prob": float(rng.choice([0.9, 0.999, 0.9999])),
                  "dyn_num_trials_mult": float(rng.choice([1.0, 3.0]))}
            if rng.integers(0, 3) == 0:
                ro["progressive_sampling"] = True
                ro["max_prosac_iterations"] = int(rng.choice([50, 1000, 100000]))
            if name == "shared_focal":
                d = synth.relative_pose_scene(n, outl, 50000 + k, noise_px=noise, focal=focal)
                pp = d["camera1"]["params"][1:3]
                opt = {"max_error": float(rng.uniform(0.8, 3.0)), "ransac": ro}
                t0 = time.perf_counter()
                pair, info = P.estimate_shared_focal_relative_pose(d["x1"], d["x2"], pp, opt)
                t_dev += time.perf_counter() - t0
                t0 = time.perf_counter()
                pose, f, mask, st = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], pp, opt)
                t_cpu += time.perf_counter() - t0
                got, gf = np.r_[pair.pose.q, pair.pose.t], pair.camera1.params[0]
            else:
                d = synth.absolute_pose_scene(n, outl, 60000 + k, noise_px=noise, focal=focal)
                opt = {"max_error": float(rng.uniform(2.0, 12.0)), "estimate_focal_length": True, "ransac": ro}
                if rng.integers(0, 3) == 0:
                    opt["min_fov"] = float(rng.choice([0.0, 1.0, 20.0, 60.0]))
                cam0 = dict(d["camera"], params=[focal * float(rng.uniform(0.7, 1.4))] + list(d["camera"]["params"][1:]))
                t0 = time.perf_counter()
                img, info = P.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, opt)
                t_dev += time.perf_counter() - t0
                t0 = time.perf_counter()
                pose, mask, st, cam = O.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, opt, return_camera=True)
                t_cpu += time.perf_counter() - t0
                got, gf, f = np.r_[img.pose.q, img.pose.t], img.camera.params[0], cam[0]
            same = (info["iterations"] == st["iterations"] and info["refinements"] == st["refinements"] and
                    info["num_inliers"] == st["num_inliers"] and np.array_equal(np.asarray(info["inliers"], dtype=bool), mask))
            exact = np.array_equal(got, pose) and gf == f
            diff = max(float(np.abs(got - pose).max()), abs(gf - f) / max(abs(f), 1e-300))
            tol_ok = exact
            worst = max(worst, 0.0 if exact else diff)
            bitwise += exact
            if not (same and tol_ok):
                bad += 1
                print(f"<!-- DISAGREEMENT {name} k={k} n={n} outliers={outl:.2f} opt={opt} device={info['iterations']},{info['refinements']},"
                      f"{info['num_inliers']} oracle={st['iterations']},{st['refinements']},{st['num_inliers']} diff={diff:.3e} -->")
        rows.append((name, count, bad, bitwise, worst, 1e3 * t_dev / count, 1e3 * t_cpu / count))
    print("# Soak of the focal-length estimators on one MI355X (scripts/soak_focal_gpu.py)\n")
    print("Random problems (7 ... 4000 correspondences, 0 - 65 % outliers, random thresholds / iteration limits / success probabilities, a third")
    print("with PROSAC, pnpf: a camera whose focal length is 30 % off, a third with min_fov 0 / 1 / 20 / 60 degrees) through `pl_estimate_shared_focal_relative_pose` and `pl_estimate_absolute_pose`")
    print("with `estimate_focal_length`, against the oracle's front-ends.\n")
    print("| estimator | problems | disagreements (iterations / refinements / inliers / mask, model beyond its bound) | models bit-identical | worst model difference otherwise | device ms / problem | oracle ms / problem |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]:.1e} | {r[5]:.2f} | {r[6]:.2f} |")
    sys.exit(1 if any(r[2] for r in rows) else 0)


if __name__ == "__main__":
    main()
