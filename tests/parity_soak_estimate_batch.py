#!/usr/bin/env python
"""Test infrastructure (uses the oracle).  Soak on a GPU box: pl_estimate_batch with the items round 6 took into the lock-step
groups - PROSAC, warm starts, OPENCV cameras - mixed with plain ones, in calls of random size and worker count, against the ORACLE's
estimate_* front-ends problem by problem: iterations, refinements, inlier count, inlier mask, model (1e-6; sign included).
    python tests/parity_soak_estimate_batch.py [problems=400] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
import poselib_amd as P  # noqa: E402
from parity_soak import model_diff  # noqa: E402
from poselib_amd import synth  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
OCV = {"model": "OPENCV", "width": 1000, "height": 1000, "params": [1000.0, 1005.0, 500.0, 498.0, 0.012, -0.003, 2e-4, -1e-4]}


def distort(p2d, f, cx, cy):
    xn = (np.asarray(p2d) - [cx, cy]) / f
    r2 = (xn ** 2).sum(1)
    k1, k2, p1, p2 = OCV["params"][4:]
    rad = 1 + k1 * r2 + k2 * r2 ** 2
    xd = np.c_[xn[:, 0] * rad + 2 * p1 * xn[:, 0] * xn[:, 1] + p2 * (r2 + 2 * xn[:, 0] ** 2),
               xn[:, 1] * rad + p1 * (r2 + 2 * xn[:, 1] ** 2) + 2 * p2 * xn[:, 0] * xn[:, 1]]
    return xd * OCV["params"][:2] + OCV["params"][2:4]


items, oracle_calls, feats = [], [], []
for i in range(total):
    kind = ["abs", "rel", "fund", "hom"][int(rng.integers(4))]
    n = int(rng.integers(12, 3000))
    outl = float(rng.uniform(0.1, 0.6))
    feat = ["plain", "prosac", "warm", "opencv", "prosac+warm"][int(rng.integers(5))]
    ro = {"seed": int(rng.integers(1 << 30))}
    if "prosac" in feat:
        ro["progressive_sampling"] = True
        if rng.random() < 0.4:
            ro["max_prosac_iterations"] = int(rng.integers(10, 400))
    opt = {"ransac": ro}
    init = None
    if kind == "abs":
        d = synth.absolute_pose_scene(n, outl, 60000 + i)
        order = np.argsort(~d["inlier_gt"], kind="stable") if "prosac" in feat else np.arange(n)
        cam, p2d, p3d = d["camera"], np.asarray(d["p2d"])[order], np.asarray(d["p3d"])[order]
        if feat == "opencv":
            cam, p2d = OCV, distort(p2d, *d["camera"]["params"])
        if "warm" in feat:
            q = np.asarray(d["q_gt"]) + 0.02 * rng.normal(size=4)
            init = np.r_[q / np.linalg.norm(q), np.asarray(d["t_gt"]) + 0.02 * rng.normal(size=3)]
            opt["ransac"]["score_initial_model"] = True
        items.append(("abs", p2d, p3d, cam, dict(opt, **({"initial_model": P.CameraPose(init[:4], init[4:])} if init is not None else {}))))
        oracle_calls.append(lambda p2d=p2d, p3d=p3d, cam=cam, opt=opt, init=init: O.estimate_absolute_pose(p2d, p3d, cam, opt, init_pose=init))
    elif kind == "rel":
        d = synth.relative_pose_scene(n, outl, 60000 + i)
        order = np.argsort(~d["inlier_gt"], kind="stable") if "prosac" in feat else np.arange(n)
        x1, x2 = np.asarray(d["x1"])[order], np.asarray(d["x2"])[order]
        c1 = c2 = d["camera1"]
        if feat == "opencv":
            c1 = c2 = OCV
            x1, x2 = distort(x1, *d["camera1"]["params"]), distort(x2, *d["camera1"]["params"])
        if "warm" in feat:
            q = np.asarray(d["q_gt"]) + 0.01 * rng.normal(size=4)
            init = np.r_[q / np.linalg.norm(q), np.asarray(d["t_gt"])]
            opt["ransac"]["score_initial_model"] = True
        items.append(("rel", x1, x2, c1, c2, dict(opt, **({"initial_model": P.CameraPose(init[:4], init[4:])} if init is not None else {}))))
        oracle_calls.append(lambda x1=x1, x2=x2, c1=c1, c2=c2, opt=opt, init=init: O.estimate_relative_pose(x1, x2, c1, c2, opt, init_pose=init))
    else:
        gen = synth.fundamental_scene if kind == "fund" else synth.homography_scene
        d = gen(n, outl, 60000 + i)
        order = np.argsort(~d["inlier_gt"], kind="stable") if "prosac" in feat else np.arange(n)
        x1, x2 = np.asarray(d["x1"])[order], np.asarray(d["x2"])[order]
        if "warm" in feat:  # a rough model: the oracle's result of a short run
            fn = O.estimate_fundamental if kind == "fund" else O.estimate_homography
            init = fn(x1, x2, {"ransac": {"seed": 7, "max_iterations": 40, "min_iterations": 10}})[0]
            opt["ransac"]["score_initial_model"] = True
        items.append((kind, x1, x2, dict(opt, **({"initial_model": init} if init is not None else {}))))
        fn = O.estimate_fundamental if kind == "fund" else O.estimate_homography
        oracle_calls.append(lambda x1=x1, x2=x2, opt=opt, init=init, fn=fn: fn(x1, x2, opt, init=init))
    feats.append((kind, feat, n))


def strip(it):  # the Python binding sets score_initial_model from "initial_model"; the key is the oracle wrapper's
    opt = dict(it[-1])
    opt["ransac"] = {k: v for k, v in opt["ransac"].items() if k != "score_initial_model"}
    return it[:-1] + (opt,)


results, report = [], {"grouped": 0, "solo": 0, "fallback": 0}
at = 0
while at < total:
    size = int(rng.integers(1, 120))
    res = P.estimate_batch([strip(it) for it in items[at:at + size]], max_in_flight=int(rng.integers(1, 9)))
    rep = P.last_batch_report()
    for k in report:
        report[k] += rep[k]
    results += res
    at += size
bad = 0
worst = 0.0
for i, ((model, info), call, ft) in enumerate(zip(results, oracle_calls, feats)):
    om, omask, ost = call()
    m = model.pose if ft[0] == "abs" else model
    dm = model_diff(ft[0], m, om)
    ok = (info["iterations"] == ost["iterations"] and info["refinements"] == ost["refinements"] and info["num_inliers"] == ost["num_inliers"]
          and (np.array(info["inliers"]) == omask).all() and dm < 1e-6)
    worst = max(worst, dm)
    if not ok:
        bad += 1
        print("DISAGREE", i, ft, info["iterations"], ost["iterations"], info["refinements"], ost["refinements"], info["num_inliers"], ost["num_inliers"], dm)
by = {}
for ft in feats:
    by[ft[1]] = by.get(ft[1], 0) + 1
print(f"pl_estimate_batch vs oracle: {total} problems ({by}), {total - bad} agree in iterations / refinements / inliers / mask / model, worst model difference {worst:.2e}; "
      f"where the items ran: {report}")
