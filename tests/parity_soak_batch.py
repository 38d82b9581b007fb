#!/usr/bin/env python
"""Test infrastructure (uses the oracle).  Parity soak of the GROUPED entry point on a GPU box: many random device-resident
problems per estimator through ONE pl_ransac_batch call (lock-step groups), every result against the oracle's ransac_*
run of the same problem: iterations, refinements, hypotheses, inlier count, mask identical, model within 1e-6.
    python tests/parity_soak_batch.py [problems per estimator] [seed] [group size] [groups in flight]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
import poselib_amd as P  # noqa: E402
from parity_soak import model_diff  # noqa: E402
from poselib_amd import synth  # noqa: E402


def main(count=100, seed=1, group=16, in_flight=4):
    rng = np.random.default_rng(seed)
    total_bad = 0
    for kind, name in ((0, "abs"), (1, "rel"), (2, "fund"), (3, "hom")):
        t0 = time.time()
        probs, opts, data = [], [], []
        for i in range(count):
            n = int(rng.integers(int(os.environ.get("SOAK_NMIN", 12)), int(os.environ.get("SOAK_NMAX", 6000))))
            outl = float(rng.uniform(0.1, 0.7))
            dseed, rseed = int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 30))
            ro = {"seed": rseed}
            u = rng.uniform()
            if u < 0.3:  # fixed-length runs of the kind bench.py times (several batches when the arena is small)
                its = int(rng.choice([500, 3000, 20000]))
                ro.update(max_iterations=its, min_iterations=its)
            elif u < 0.45:
                ro.update(max_iterations=int(rng.choice([1, 7, 300, 5000])), min_iterations=int(rng.choice([0, 5, 400])),
                          success_prob=float(rng.choice([0.5, 0.99, 0.9999])), dyn_num_trials_mult=float(rng.choice([0.5, 3.0, 10.0])))
            if kind == 0:
                d = synth.absolute_pose_scene(n, outl, dseed)
                par = d["camera"]["params"]
                a, b = (np.asarray(d["p2d"]) - np.array(par[-2:])) / par[0], np.asarray(d["p3d"])
                thr = float(rng.choice([2e-3, 1.2e-2]))
            else:
                gen = {1: synth.relative_pose_scene, 2: synth.fundamental_scene, 3: synth.homography_scene}[kind]
                d = gen(n, outl, dseed)
                a, b = (np.asarray(d["x1"]) - 500.0) / 1000.0, (np.asarray(d["x2"]) - 500.0) / 1000.0
                thr = float(rng.choice([5e-4, 1e-3, 3e-3]))
            probs.append(P.Problem(kind, a, b))
            opts.append({"max_error": thr, "ransac": ro})
            data.append((a, b))
        got = P.ransac_batch(probs, opts, in_flight, group)
        t_gpu = time.time() - t0
        bad = ref_only = 0
        worst = 0.0
        ofn = {0: O.ransac_pnp, 1: O.ransac_relpose, 2: O.ransac_fundamental, 3: O.ransac_homography}[kind]
        for idx, ((a, b), opt, (m, info)) in enumerate(zip(data, opts, got)):
            want, mask, st = ofn(a, b, opt)
            same = (info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
                    and info["hypotheses"] == st["hypotheses"] and (np.array(info["inliers"]) == mask).all())
            diff = model_diff(name, m, want) if st["num_inliers"] > 0 else 0.0
            worst = max(worst, diff if same else 0.0)
            if not same or diff > 1e-6:
                bad += 1
                print(f"  MISMATCH {name} n={len(a)} opt={opt}: iterations {info['iterations']}/{st['iterations']} "
                      f"inliers {info['num_inliers']}/{st['num_inliers']} hypotheses {info['hypotheses']}/{st['hypotheses']} diff {diff:.2e}")
                sm, sinfo = probs[idx].run(opt)  # the single-problem entry point on the same problem
                print(f"    pl_ransac_run on it: model diff to the batch result {model_diff(name, sm, np.r_[m.q, m.t] if hasattr(m, 'q') else np.ravel(m)):.2e}, "
                      f"to the oracle {model_diff(name, sm, want):.2e}; refinements batch/single/oracle "
                      f"{info['refinements']}/{sinfo['refinements']}/{st['refinements']}; scores {info['model_score']!r} {sinfo['model_score']!r} {st['model_score']!r}")
                if os.environ.get("SOAK_DUMP"):
                    np.savez(os.path.join(os.environ["SOAK_DUMP"], f"mismatch_{name}_{idx}.npz"), a=a, b=b, thr=opt["max_error"],
                             seed=opt["ransac"]["seed"])
            elif info["refinements"] != st["refinements"]:
                ref_only += 1
        for p in probs:
            p.close()
        total_bad += bad
        print(f"{name}: {count} problems through pl_ransac_batch (groups of {group}, {in_flight} in flight), {bad} disagreements, "
              f"{ref_only} with a different refinement count only, worst model difference among the agreeing {worst:.2e}, "
              f"GPU {t_gpu:.1f} s, total {time.time() - t0:.1f} s")
    return total_bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:5])) else 0)
