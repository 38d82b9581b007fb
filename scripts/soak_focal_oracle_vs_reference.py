"""Soak (CPU): the ORACLE's focal-length estimators (solvers of this project's own formulation) against the REFERENCE'S OWN SOURCES
(oracle/_ref: its generated templates) on random problems - how often the two take identical decisions, and how far the final
estimates are apart when they do not.
    python scripts/soak_focal_oracle_vs_reference.py [problems per estimator] > profiles/r03_soak_focal_oracle_vs_reference.md"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import ref_lib  # noqa: E402
from poselib_amd import synth  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(7)
    rows, notes = [], []
    for name in ("shared_focal", "pnpf"):
        same = same_mask = one_lo = 0
        fdiff = []
        for k in range(count):
            n = int(rng.integers(30, 3000))
            outl = float(rng.uniform(0.05, 0.6))
            focal = float(rng.uniform(500, 2500))
            noise = float(rng.uniform(0.1, 1.5))
            ro = {"seed": k, "max_iterations": 5000}
            if name == "shared_focal":
                d = synth.relative_pose_scene(n, outl, 9000 + k, focal=focal, noise_px=noise)
                pp = d["camera1"]["params"][1:3]
                opt = {"max_error": float(rng.uniform(1, 3)), "ransac": ro}
                po, fo, mo, so = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], pp, opt)
                with ref_lib.reference():
                    pr, fr, mr, sr = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], pp, opt)
            else:
                d = synth.absolute_pose_scene(n, outl, 9500 + k, focal=focal, noise_px=noise)
                cam0 = dict(d["camera"], params=[1.2 * focal] + list(d["camera"]["params"][1:]))
                opt = {"max_error": float(rng.uniform(2, 10)), "estimate_focal_length": True, "ransac": ro}
                po, mo, so, co = O.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, opt, return_camera=True)
                with ref_lib.reference():
                    pr, mr, sr, cr = O.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, opt, return_camera=True)
                fo, fr = co[0], cr[0]
            eq = so["iterations"] == sr["iterations"] and so["refinements"] == sr["refinements"] and np.array_equal(mo, mr)
            same += eq
            same_mask += np.array_equal(mo, mr)
            one_lo += (not eq) and so["iterations"] == sr["iterations"] and abs(so["refinements"] - sr["refinements"]) == 1 and np.array_equal(mo, mr)
            fdiff.append(abs(fo - fr) / max(abs(fr), 1e-300))
            if not eq:
                notes.append(f"{name} k={k} n={n} outliers={outl:.2f}: iterations {so['iterations']} / {sr['iterations']}, refinements "
                             f"{so['refinements']} / {sr['refinements']}, inliers {so['num_inliers']} / {sr['num_inliers']}, focal {fo:.9g} / {fr:.9g}")
        rows.append((name, count, same, one_lo, same_mask, float(np.median(fdiff)), float(np.max(fdiff))))
    print("# Oracle against the reference's own sources: the two focal-length estimators (scripts/soak_focal_oracle_vs_reference.py)\n")
    print("Random problems (30 ... 3000 correspondences, 5 - 60 % outliers, random focal lengths / noise / thresholds) through the front-ends of")
    print("the oracle (this project's formulations of P3.5Pf and of the 6-point shared-focal solver) and of `oracle/_ref` (the reference's")
    print("sources with their generated templates, compiled in place).  The solvers differ, everything after them is the same arithmetic.\n")
    print("| estimator | problems | identical iterations, refinements and mask | of the others: same iterations and mask, ONE local optimisation apart | identical masks | focal length, relative difference: median | max |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]:.1e} | {r[6]:.1e} |")
    print("\nThe problems that differ:\n")
    for s in notes:
        print("* " + s)


if __name__ == "__main__":
    main()
