"""Times the two focal-length estimators through the C-ABI (device) and the oracle (CPU) on the same problems.
    python scripts/time_focal_estimators.py [reps]
Used for profiles/r03_focal_estimators_*.md (rocprofv3 --kernel-trace --stats of this script)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import ref_lib  # noqa: E402
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    for n in (500, 2000, 5000):
        d = synth.relative_pose_scene(n, 0.4, 7000 + n)
        f, cx, cy = d["camera1"]["params"]
        opt = {"max_error": 2.0, "ransac": {"seed": 1}}
        P.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], opt)  # warm-up
        t0 = time.perf_counter()
        for r in range(reps):
            pair, info = P.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], dict(opt, ransac={"seed": 1 + r}))
        t_dev = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        ref = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], dict(opt, ransac={"seed": reps}))
        t_cpu = time.perf_counter() - t0
        ok = info["iterations"] == ref[3]["iterations"] and pair.camera1.params[0] == ref[1]
        t_ref = float("nan")
        if ref_lib.available():
            t0 = time.perf_counter()
            with ref_lib.reference():
                O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], dict(opt, ransac={"seed": reps}))
            t_ref = time.perf_counter() - t0
        print(f"shared_focal n={n}: device {1e3 * t_dev:.2f} ms/problem, oracle {1e3 * t_cpu:.2f} ms, reference sources {1e3 * t_ref:.2f} ms, iterations {info['iterations']}, "
              f"evaluated {info['iterations_evaluated']}, parity {ok}")
        d = synth.absolute_pose_scene(n, 0.4, 7100 + n)
        opt = {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": 1}}
        P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
        t0 = time.perf_counter()
        for r in range(reps):
            img, info = P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], dict(opt, ransac={"seed": 1 + r}))
        t_dev = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        pose, mask, st, cam = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], dict(opt, ransac={"seed": reps}), return_camera=True)
        t_cpu = time.perf_counter() - t0
        t_ref = float("nan")
        if ref_lib.available():
            t0 = time.perf_counter()
            with ref_lib.reference():
                O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], dict(opt, ransac={"seed": reps}))
            t_ref = time.perf_counter() - t0
        print(f"pnpf n={n}: device {1e3 * t_dev:.2f} ms/problem, oracle {1e3 * t_cpu:.2f} ms, reference sources {1e3 * t_ref:.2f} ms, iterations {info['iterations']}, "
              f"evaluated {info['iterations_evaluated']}, parity {info['iterations'] == st['iterations']}")


if __name__ == "__main__":
    main()
