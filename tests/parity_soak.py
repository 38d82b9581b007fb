#!/usr/bin/env python
"""Test infrastructure (uses the oracle).  Parity soak on a GPU box: many random problems per estimator, product (through the C-ABI) vs oracle.
Compares iterations, refinements, inlier count, inlier mask and the model (1e-6).  Prints one summary line per
estimator and every disagreement.   python tests/parity_soak.py [problems per estimator] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402


from golden.make_gauge import DT_LEN_BOUND as REL_DT_LEN_BOUND  # noqa: E402  (|t| gauge: guarded at 1e-3)


def model_diff(kind, a, b):
    """Poses: quaternion and translation; relative poses: R and the DIRECTION of t are held to the tolerance, the LENGTH
    of t - a gauge of the reference's LM (it steps t in its tangent plane and never renormalises; the reference does not
    reproduce |t| across its own builds: tests/golden/make_gauge.py) - only to the guard of 1e-3.
    Returned: the largest of the components, each divided by its bound and scaled back to the 1e-6 tolerance."""
    if kind in ("abs", "rel"):
        qa, qb = np.asarray(a.q), np.asarray(b[:4])
        dq = min(np.abs(qa - qb).max(), np.abs(qa + qb).max())
        ta, tb = np.asarray(a.t), np.asarray(b[4:7])
        if kind == "rel":
            na, nb = np.linalg.norm(ta) + 1e-300, np.linalg.norm(tb) + 1e-300
            d_dir = np.abs(ta / na - tb / nb).max()
            d_len = abs(na - nb)
            return max(dq, d_dir, d_len * (1e-6 / REL_DT_LEN_BOUND))
        return max(dq, np.abs(ta - tb).max())
    A, B = np.asarray(a) / np.linalg.norm(a), np.asarray(b) / np.linalg.norm(b)
    return np.abs(A - B).max()  # sign included


def degrade(rng, d, keys):
    """SOAK_FUZZ3: duplicated correspondences, a block of identical points, a few wildly scaled outliers"""
    n = d[keys[0]].shape[0]
    if n < 12:
        return d
    idx = np.arange(n)
    dup = rng.integers(0, n, size=max(1, n // 5))
    idx[rng.integers(0, n, size=dup.size)] = dup  # duplicates
    blk = rng.integers(0, n, size=max(1, n // 10))
    idx[blk] = idx[blk[0]]  # a block of identical correspondences
    out = dict(d)
    for k in keys:
        out[k] = np.ascontiguousarray(d[k][idx])
    wild = rng.integers(0, n, size=3)
    out[keys[0]] = out[keys[0]].copy()
    out[keys[0]][wild] *= np.array([1e4, -1e3])[: out[keys[0]].shape[1]] if out[keys[0]].shape[1] == 2 else 1.0
    return out


def main(count=100, seed=1):
    rng = np.random.default_rng(seed)
    total_bad = 0
    for kind in ("abs", "rel", "fund", "hom"):
        bad = ref_only = ambiguous = 0
        worst = 0.0
        t0 = time.time()
        for i in range(count):
            n = int(rng.integers(int(os.environ.get("SOAK_NMIN", 12)), int(os.environ.get("SOAK_NMAX", 3000))))
            outl = float(rng.uniform(0.1, 0.7))
            dseed, rseed = int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 30))
            opt = {"ransac": {"seed": rseed}}
            if rng.uniform() < 0.3:
                opt["ransac"].update(max_iterations=3000, min_iterations=int(rng.integers(100, 3000)))
            if rng.uniform() < 0.15:
                opt["ransac"].update(progressive_sampling=True, max_prosac_iterations=int(rng.integers(50, 2000)))
            if os.environ.get("SOAK_FUZZ"):  # odd option values
                opt["ransac"].update(max_iterations=int(rng.choice([0, 1, 7, 300, 5000])),
                                     min_iterations=int(rng.choice([0, 5, 400, 9000])),
                                     success_prob=float(rng.choice([0.5, 0.99, 0.9999, 1.0])),
                                     dyn_num_trials_mult=float(rng.choice([0.5, 3.0, 10.0])))
                opt["max_error"] = float(rng.choice([0.05, 1.0, 12.0, 100.0]))
            cam_fuzz = None
            if os.environ.get("SOAK_FUZZ2"):  # final-refinement options and camera models
                opt["bundle"] = {"loss_type": str(rng.choice(["TRIVIAL", "TRUNCATED", "HUBER", "CAUCHY", "TRUNCATED_LE_ZACH"])),
                                 "loss_scale": float(rng.choice([0.3, 1.0, 5.0])),
                                 "max_iterations": int(rng.choice([0, 3, 100]))}
                which = int(rng.integers(0, 3))
                if which == 1:
                    cam_fuzz = {"model": "PINHOLE", "width": 1000, "height": 1000,
                                "params": [float(rng.uniform(900, 1100)), float(rng.uniform(900, 1100)),
                                           float(rng.uniform(480, 520)), float(rng.uniform(480, 520))]}
                elif which == 2:
                    cam_fuzz = {"model": "OPENCV", "width": 1000, "height": 1000,
                                "params": [float(rng.uniform(900, 1100)), float(rng.uniform(900, 1100)),
                                           float(rng.uniform(480, 520)), float(rng.uniform(480, 520)),
                                           float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.02, 0.02)),
                                           float(rng.uniform(-0.002, 0.002)), float(rng.uniform(-0.002, 0.002))]}
            if kind == "abs":
                d = synth.absolute_pose_scene(n, outl, dseed)
                if os.environ.get("SOAK_FUZZ3"):
                    d = degrade(rng, d, ("p2d", "p3d"))
                cam = cam_fuzz or d["camera"]
                init_p = init_o = None
                if os.environ.get("SOAK_FUZZ4") and rng.uniform() < 0.7:  # warm start (score_initial_model)
                    q = np.asarray(d["q_gt"]) + rng.normal(0, float(rng.choice([1e-4, 1e-2, 0.5])), 4)
                    q /= np.linalg.norm(q)
                    t = np.asarray(d["t_gt"]) + rng.normal(0, 0.01, 3)
                    init_p, init_o = P.CameraPose(q, t), np.r_[q, t]
                oo = dict(opt, ransac=dict(opt["ransac"], score_initial_model=init_o is not None))
                got, info = P.estimate_absolute_pose(d["p2d"], d["p3d"], cam, opt, init_p)
                want, mask, st = O.estimate_absolute_pose(d["p2d"], d["p3d"], cam, oo, init_o)
                got = got.pose
            elif kind == "rel":
                d = synth.relative_pose_scene(n, outl, dseed)
                if os.environ.get("SOAK_FUZZ3"):
                    d = degrade(rng, d, ("x1", "x2"))
                c1, c2 = cam_fuzz or d["camera1"], cam_fuzz or d["camera2"]
                got, info = P.estimate_relative_pose(d["x1"], d["x2"], c1, c2, opt)
                want, mask, st = O.estimate_relative_pose(d["x1"], d["x2"], c1, c2, opt)
            elif kind == "fund":
                d = synth.fundamental_scene(n, outl, dseed)
                if os.environ.get("SOAK_FUZZ3"):
                    d = degrade(rng, d, ("x1", "x2"))
                init_m = None
                if os.environ.get("SOAK_FUZZ4"):
                    opt["real_focal_check"] = bool(rng.uniform() < 0.5)
                    if rng.uniform() < 0.5:
                        init_m = rng.normal(size=(3, 3))
                oo = dict(opt, ransac=dict(opt["ransac"], score_initial_model=init_m is not None))
                got, info = P.estimate_fundamental(d["x1"], d["x2"], opt, init_m)
                want, mask, st = O.estimate_fundamental(d["x1"], d["x2"], oo, init_m)
            else:
                d = synth.homography_scene(n, outl, dseed)
                if os.environ.get("SOAK_FUZZ3"):
                    d = degrade(rng, d, ("x1", "x2"))
                init_m = None
                if os.environ.get("SOAK_FUZZ4") and rng.uniform() < 0.5:
                    init_m = np.eye(3) + rng.normal(0, 0.05, (3, 3))
                oo = dict(opt, ransac=dict(opt["ransac"], score_initial_model=init_m is not None))
                got, info = P.estimate_homography(d["x1"], d["x2"], opt, init_m)
                want, mask, st = O.estimate_homography(d["x1"], d["x2"], oo, init_m)
            if os.environ.get("SOAK_RANSAC"):  # the ransac_* entry points on normalised image points instead
                ro = dict(opt, max_error=float(rng.choice([5e-4, 1e-3, 1.2e-2])))
                if kind == "abs":
                    par = d["camera"]["params"]
                    x = (d["p2d"] - np.array(par[-2:])) / par[0]
                    got, info = P.ransac_pnp(x, d["p3d"], ro)
                    want, mask, st = O.ransac_pnp(x, d["p3d"], ro)
                else:
                    x1, x2 = (d["x1"] - 500.0) / 1000.0, (d["x2"] - 500.0) / 1000.0
                    fn, ofn = {"rel": (P.ransac_relpose, O.ransac_relpose), "fund": (P.ransac_fundamental, O.ransac_fundamental),
                               "hom": (P.ransac_homography, O.ransac_homography)}[kind]
                    got, info = fn(x1, x2, ro)
                    want, mask, st = ofn(x1, x2, ro)
            same = (info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
                    and (np.array(info["inliers"]) == mask).all())
            diff = model_diff(kind, got, want) if st["num_inliers"] > 0 else 0.0
            worst = max(worst, diff if same else 0.0)
            if ((not same or diff > 1e-6) and os.environ.get("SOAK_FUZZ3") and kind == "rel"
                    and info["num_inliers"] == st["num_inliers"] and info["iterations"] == st["iterations"]):
                # duplicated correspondences inside a minimal sample make the 5-point problem rank deficient: the null
                # space basis - and with it the solutions - of two equivalent, not bit-identical solvers differ, and
                # another model with the same support can win
                ambiguous += 1
            elif not same or diff > 1e-6:
                bad += 1
                print(f"  MISMATCH {kind} n={n} outl={outl:.2f} dseed={dseed} rseed={rseed} opt={opt['ransac']}: "
                      f"iterations {info['iterations']}/{st['iterations']} refinements {info['refinements']}/{st['refinements']} "
                      f"inliers {info['num_inliers']}/{st['num_inliers']} model diff {diff:.2e}")
            elif info["refinements"] != st["refinements"]:
                ref_only += 1
        total_bad += bad
        print(f"{kind}: {count} problems, {bad} disagreements, {ref_only} with a different refinement count only, "
              + (f"{ambiguous} ambiguous (same support, another pose), " if ambiguous else "") +
              f"worst model difference among the agreeing {worst:.2e}, {time.time() - t0:.1f} s")
    return total_bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
