#!/usr/bin/env python
"""Generates tests/golden/golden_v1.json — frozen input/output vectors of the hot path.

PROVENANCE: the reference (PoseLib 3.0.0) cannot be compiled or imported in this environment (Eigen3 is
absent, no network) and its own tests hold no golden vectors for this path, so these vectors are produced
by the CPU ORACLE (oracle/, a restatement of the reference) and frozen.  They pin the oracle against
silent regressions and give the GPU tests a second, run-independent target.  Inputs are regenerated from
poselib_amd.synth (bit-reproducible splitmix64 streams); their SHA-256 is stored so that a generator change
is detected.  Re-run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from poselib_amd import synth  # noqa: E402

CASES = [
    # name, kind, n, outlier ratio, data seed, options
    ("cfg0_p3p_200", "abs", 200, 0.5, 1000, {}),
    ("cfg0_p3p_200_seed7", "abs", 200, 0.5, 1000, {"ransac": {"seed": 7}}),
    ("p3p_1500", "abs", 1500, 0.6, 2001, {"ransac": {"seed": 2}}),
    ("cfg1_p3p_5000", "abs", 5000, 0.7, 1001, {}),
    ("rel_800", "rel", 800, 0.4, 2002, {"ransac": {"seed": 1}}),
    ("cfg2_5pt_5000", "rel", 5000, 0.5, 1002, {}),
    ("hom_1000", "hom", 1000, 0.4, 2003, {"ransac": {"seed": 3}}),
    ("cfg3_hom_10000", "hom", 10000, 0.5, 1003, {}),
    ("fund_1000", "fund", 1000, 0.4, 2004, {"ransac": {"seed": 4}}),
    ("cfg3_fund_10000", "fund", 10000, 0.5, 1004, {}),
]


def scene(kind, n, outl, seed):
    if kind == "abs":
        d = synth.absolute_pose_scene(n, outl, seed)
        return d, [d["p2d"], d["p3d"]]
    if kind == "rel":
        d = synth.relative_pose_scene(n, outl, seed)
        return d, [d["x1"], d["x2"]]
    if kind == "hom":
        d = synth.homography_scene(n, outl, seed, noise_px=0.3)
        return d, [d["x1"], d["x2"]]
    d = synth.fundamental_scene(n, outl, seed)
    return d, [d["x1"], d["x2"]]


def digest(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def run_oracle(kind, d, opt):
    if kind == "abs":
        model, mask, st = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
    elif kind == "rel":
        model, mask, st = O.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
    elif kind == "hom":
        model, mask, st = O.estimate_homography(d["x1"], d["x2"], opt)
    else:
        model, mask, st = O.estimate_fundamental(d["x1"], d["x2"], opt)
    return model, mask, st


def main():
    out = {"provenance": "oracle-generated (reference not buildable here: Eigen3 absent); see make_golden.py", "cases": []}
    for name, kind, n, outl, seed, opt in CASES:
        d, arrs = scene(kind, n, outl, seed)
        model, mask, st = run_oracle(kind, d, opt)
        out["cases"].append({
            "name": name, "kind": kind, "n": n, "outlier_ratio": outl, "data_seed": seed, "options": opt,
            "input_sha256": digest(arrs),
            "iterations": st["iterations"], "refinements": st["refinements"], "num_inliers": st["num_inliers"],
            "model_score": repr(float(st["model_score"])),
            "model": [repr(float(v)) for v in np.asarray(model).reshape(-1)],
            "mask_hex": np.packbits(mask.astype(np.uint8)).tobytes().hex(),
        })
        print(name, st["iterations"], st["refinements"], st["num_inliers"])
    # minimal-solver vectors: 12 P3P instances (sample of cfg1 scene)
    d = synth.absolute_pose_scene(5000, 0.7, 1001)
    un = O.unproject(d["camera"], d["p2d"])
    idx, _ = O.sampler_draw(0, 5000, 3, 12)
    sol = []
    for s in idx.astype(np.int64):
        b = np.c_[un[s], np.ones(3)]
        b /= np.sqrt((b * b).sum(1))[:, None]
        sol.append([[repr(float(v)) for v in p] for p in O.p3p(b, d["p3d"][s])])
    out["p3p_cfg1_first12"] = sol
    with open(os.path.join(HERE, "golden_v1.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
