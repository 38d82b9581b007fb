#!/usr/bin/env python
"""Generates tests/golden/golden_focal_v1.json - frozen input/output vectors of the two focal-length estimators:
estimate_absolute_pose with estimate_focal_length (ransac_pnpf) and estimate_shared_focal_relative_pose, plus minimal-solver
vectors of P3.5Pf and of the 6-point shared-focal solver.

PROVENANCE: produced by the CPU ORACLE (oracle/src/solvers_focal.cc: this project's own formulations of the two solvers) and
frozen; tests/test_golden_focal.py pins (a) the oracle against them bit for bit, (b) the REFERENCE's own sources (oracle/_ref) on
the decisions - iterations, refinements, inliers, mask - and the model to 1e-6, (c) the HIP path through the C-ABI.  Inputs are
regenerated from poselib_amd.synth; their SHA-256 is stored.  Re-run:  python tests/golden/make_golden_focal.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from golden.make_golden import digest  # noqa: E402
from poselib_amd import synth  # noqa: E402

CASES = [
    # name, kind, n, outlier ratio, data seed, options
    ("pnpf_200", "pnpf", 200, 0.3, 3001, {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": 1}}),
    ("pnpf_256", "pnpf", 256, 0.5, 3002, {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": 2}}),
    ("pnpf_1500", "pnpf", 1500, 0.5, 3003, {"max_error": 6.0, "estimate_focal_length": True, "ransac": {"seed": 3}}),
    ("pnpf_5000", "pnpf", 5000, 0.6, 3004, {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": 4}}),
    ("shared_focal_100", "shared_focal", 100, 0.2, 3011, {"max_error": 2.0, "ransac": {"seed": 1}}),
    ("shared_focal_800", "shared_focal", 800, 0.4, 3012, {"max_error": 2.0, "ransac": {"seed": 2}}),
    ("shared_focal_2000", "shared_focal", 2000, 0.5, 3013, {"max_error": 1.5, "ransac": {"seed": 3}}),
    ("shared_focal_5000", "shared_focal", 5000, 0.5, 3014, {"max_error": 2.0, "ransac": {"seed": 4}}),
]


def scene(kind, n, outl, seed):
    if kind == "pnpf":
        d = synth.absolute_pose_scene(n, outl, seed)
        d["camera_in"] = dict(d["camera"], params=[1.25 * d["camera"]["params"][0]] + list(d["camera"]["params"][1:]))  # 25 % off
        return d, [d["p2d"], d["p3d"]]
    d = synth.relative_pose_scene(n, outl, seed)
    return d, [d["x1"], d["x2"]]


def run_oracle(kind, d, opt):
    """-> (model = pose7 + focal, mask, stats)"""
    if kind == "pnpf":
        pose, mask, st, cam = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera_in"], opt, return_camera=True)
        return np.r_[pose, cam[0]], mask, st
    pose, focal, mask, st = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], d["camera1"]["params"][1:3], opt)
    return np.r_[pose, focal], mask, st


def minimal_inputs():
    """four P3.5Pf samples (x 4 x 2, X 4 x 3) and four 6-point samples (unit bearings) out of the first scenes"""
    d = synth.absolute_pose_scene(200, 0.0, 3001, noise_px=0.0)
    f, cx, cy = d["camera"]["params"]
    x = np.asarray(d["p2d"]) - [cx, cy]
    p35 = [(x[4 * k:4 * k + 4], np.asarray(d["p3d"])[4 * k:4 * k + 4]) for k in range(4)]
    r = synth.relative_pose_scene(100, 0.0, 3011, noise_px=0.0)
    f, cx, cy = r["camera1"]["params"]

    def unit(p):
        b = np.c_[(np.asarray(p) - [cx, cy]) / 800.0, np.ones(len(p))]
        return b / np.linalg.norm(b, axis=1)[:, None]

    six = [(unit(r["x1"][6 * k:6 * k + 6]), unit(r["x2"][6 * k:6 * k + 6])) for k in range(4)]
    return p35, six


def main():
    out = {"provenance": "oracle-generated (solvers: this project's own formulations); see make_golden_focal.py", "cases": []}
    for name, kind, n, outl, seed, opt in CASES:
        d, arrs = scene(kind, n, outl, seed)
        model, mask, st = run_oracle(kind, d, opt)
        out["cases"].append({
            "name": name, "kind": kind, "n": n, "outlier_ratio": outl, "data_seed": seed, "options": opt,
            "input_sha256": digest(arrs),
            "iterations": st["iterations"], "refinements": st["refinements"], "num_inliers": st["num_inliers"],
            "model_score": repr(float(st["model_score"])),
            "model": [repr(float(v)) for v in model],
            "mask_hex": np.packbits(mask.astype(np.uint8)).tobytes().hex(),
        })
        print(name, st["iterations"], st["refinements"], st["num_inliers"], model[-1])
    p35, six = minimal_inputs()
    out["p35pf"] = [{"poses": [[repr(float(v)) for v in p] for p in O.p35pf(x, X)[0]], "focals": [repr(float(v)) for v in O.p35pf(x, X)[1]]}
                    for x, X in p35]
    out["six_point"] = [{"poses": [[repr(float(v)) for v in p] for p in O.relpose_6pt_shared_focal(a, b)[0]],
                         "focals": [repr(float(v)) for v in O.relpose_6pt_shared_focal(a, b)[1]]} for a, b in six]
    with open(os.path.join(HERE, "golden_focal_v1.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
