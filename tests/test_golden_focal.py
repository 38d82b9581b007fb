"""Golden fixtures of the two focal-length estimators (tests/golden/golden_focal_v1.json, oracle-generated and frozen - see
make_golden_focal.py for the provenance).  CPU: the oracle keeps reproducing them bit for bit; the REFERENCE'S OWN SOURCES
(oracle/_ref) take the same decisions - iterations, refinements, inliers, mask - and return the same model to 1e-6 (the minimal
solvers are different formulations); the device headers compiled for the host reproduce the solver vectors bit for bit.
GPU: the HIP path through the C-ABI matches them (shared focal bit for bit, pnpf to 1e-9 above 256 correspondences)."""
import json
import os

import numpy as np
import pytest

import hostmath_lib as HM
import oracle_lib as O
import ref_lib
from golden.make_golden import digest
from golden.make_golden_focal import minimal_inputs, run_oracle, scene

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_focal_v1.json")))
CASES = {c["name"]: c for c in G["cases"]}


def unpack_mask(c):
    return np.unpackbits(np.frombuffer(bytes.fromhex(c["mask_hex"]), dtype=np.uint8))[: c["n"]].astype(bool)


def _scene(c):
    d, arrs = scene(c["kind"], c["n"], c["outlier_ratio"], c["data_seed"])
    assert digest(arrs) == c["input_sha256"], "synthetic generator changed: regenerate the fixtures"
    return d


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_focal_golden(name):
    c = CASES[name]
    model, mask, st = run_oracle(c["kind"], _scene(c), c["options"])
    assert (st["iterations"], st["refinements"], st["num_inliers"]) == (c["iterations"], c["refinements"], c["num_inliers"])
    assert (mask == unpack_mask(c)).all()
    assert [repr(float(v)) for v in model] == c["model"]
    assert repr(float(st["model_score"])) == c["model_score"]


@pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built and /root/reference absent")
@pytest.mark.parametrize("name", sorted(CASES))
def test_focal_golden_through_the_reference_sources(name):
    """robust.cc:47-54 -> ransac_pnpf / robust.cc:366-424 -> ransac_shared_focal_relpose of the reference itself"""
    c = CASES[name]
    with ref_lib.reference():
        model, mask, st = run_oracle(c["kind"], _scene(c), c["options"])
    assert (st["iterations"], st["refinements"], st["num_inliers"]) == (c["iterations"], c["refinements"], c["num_inliers"])
    assert (mask == unpack_mask(c)).all()
    want = np.array([float(v) for v in c["model"]])
    assert np.abs(model[:4] - want[:4]).max() < 1e-6 and abs(model[7] - want[7]) < 1e-6 * want[7]
    if c["kind"] == "pnpf":
        assert np.abs(model[4:7] - want[4:7]).max() < 1e-6
    else:  # (|t| of a relative pose is a gauge)
        assert np.abs(model[4:7] / np.linalg.norm(model[4:7]) - want[4:7] / np.linalg.norm(want[4:7])).max() < 1e-6


def test_minimal_solver_vectors_oracle_and_device_headers():
    p35, six = minimal_inputs()
    for (x, X), want in zip(p35, G["p35pf"]):
        for poses, focals in (O.p35pf(x, X), HM.p35pf(x, X, stride=3)):
            assert [[repr(float(v)) for v in p] for p in poses] == want["poses"]
            assert [repr(float(v)) for v in focals] == want["focals"]
    for (a, b), want in zip(six, G["six_point"]):
        for poses, focals in (O.relpose_6pt_shared_focal(a, b), HM.relpose_6pt_shared_focal(a, b, stride=2)):
            assert [[repr(float(v)) for v in p] for p in poses] == want["poses"]
            assert [repr(float(v)) for v in focals] == want["focals"]
    assert sum(len(w["focals"]) for w in G["p35pf"]) >= 8 and sum(len(w["focals"]) for w in G["six_point"]) >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_matches_focal_golden(gpu, name):
    c = CASES[name]
    d = _scene(c)
    if c["kind"] == "pnpf":
        img, info = gpu.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera_in"], c["options"])
        model = np.r_[img.pose.q, img.pose.t, img.camera.params[0]]
    else:
        pair, info = gpu.estimate_shared_focal_relative_pose(d["x1"], d["x2"], d["camera1"]["params"][1:3], c["options"])
        model = np.r_[pair.pose.q, pair.pose.t, pair.camera1.params[0]]
    assert (info["iterations"], info["refinements"], info["num_inliers"]) == (c["iterations"], c["refinements"], c["num_inliers"])
    assert (np.array(info["inliers"]) == unpack_mask(c)).all()
    want = np.array([float(v) for v in c["model"]])
    if c["kind"] == "shared_focal" or c["n"] <= 256:
        assert np.array_equal(model, want), np.abs(model - want).max()
        assert repr(float(info["model_score"])) == c["model_score"]
    else:
        assert np.abs(model[:7] - want[:7]).max() < 1e-9 and abs(model[7] - want[7]) < 1e-9 * want[7]
