"""The committed golden fixtures and the full-size headline configuration against the REFERENCE'S OWN SOURCES
(oracle/_ref/libposelib_ref.so, built in place from /root/reference by oracle/Makefile.ref at the reference's Release
flags -O3; caveat: against oracle/eigen_shim, real Eigen is not in this image).

* tests/golden/golden_v1.json is oracle-generated; here every case is re-run through the reference's front-ends
  (robust.cc:36-126, 242-314, 544-594, 712-757 -> robust/ransac.cc:44-57, 142-154, 248-262, 300-314): iterations,
  refinements, inlier count and mask must be identical, models agree to 1e-8.
* BASELINE configs[1] at 100 000 iterations (the timed configuration of bench.py), two RANSAC seeds: oracle == reference
  in iterations, refinements, inliers and mask.
* The relative-pose gauge, stated and tested: R and the DIRECTION of t are held to 1e-9 everywhere; |t| is a quantity
  the reference does not reproduce across its own builds (tests/golden/make_gauge.py: -O3 SSE2 vs -O3 x86-64-v3, up to
  6.5e-6 on 1000 problems, 2.2 % above 1e-6; 1.4e-4 oracle-vs-reference in one soak problem), so d|t| is only guarded
  (make_gauge.DT_LEN_BOUND = 1e-3) and reported, never claimed.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import ref_lib
from golden.make_gauge import DT_LEN_BOUND, measure, parts, problems
from golden.make_golden import digest, run_oracle, scene
from poselib_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "golden_v1.json")))
GAUGE = json.load(open(os.path.join(HERE, "golden", "relpose_gauge_v1.json")))["measured"]
DIR_BOUND = 1e-9                            # dR, d(t/|t|)

pytestmark = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built and /root/reference absent")


def unpack_mask(c):
    return np.unpackbits(np.frombuffer(bytes.fromhex(c["mask_hex"]), dtype=np.uint8))[: c["n"]].astype(bool)


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: c["name"])
def test_golden_fixture_through_the_reference_sources(case):
    c = case
    d, arrs = scene(c["kind"], c["n"], c["outlier_ratio"], c["data_seed"])
    assert digest(arrs) == c["input_sha256"]
    with ref_lib.reference():
        model, mask, st = run_oracle(c["kind"], d, c["options"])
    assert (st["iterations"], st["refinements"], st["num_inliers"]) == (c["iterations"], c["refinements"], c["num_inliers"])
    assert (np.asarray(mask, dtype=bool) == unpack_mask(c)).all()
    want = np.array([float(v) for v in c["model"]])
    got = np.asarray(model, dtype=np.float64).reshape(-1)
    if c["kind"] == "rel":
        p = parts(got, want)
        assert p["dR"] < DIR_BOUND and p["dt_dir"] < DIR_BOUND, p
        assert p["dt_len"] <= DT_LEN_BOUND, p
    elif c["kind"] == "abs":
        assert np.abs(got - want).max() < 1e-8
    else:
        a, b = got / np.linalg.norm(got), want / np.linalg.norm(want)
        assert np.linalg.norm(a - b) < 1e-8  # sign included


@pytest.mark.parametrize("seed", [0, 1])
def test_headline_configuration_at_full_size_oracle_equals_reference(seed):
    """BASELINE configs[1] exactly as bench.py times it: 5000 correspondences, 70 % outliers, 100 000 iterations"""
    d = synth.absolute_pose_scene(5000, 0.7, 1001)
    x = (np.asarray(d["p2d"]) - 500.0) / 1000.0
    opt = {"max_error": 12.0 / 1000.0, "ransac": {"max_iterations": 100000, "min_iterations": 100000, "seed": seed}}
    mo, ko, so = O.ransac_pnp(x, d["p3d"], opt)
    with ref_lib.reference():
        mr, kr, sr = O.ransac_pnp(x, d["p3d"], opt)
    for k in ("iterations", "refinements", "num_inliers"):
        assert so[k] == sr[k], (k, so, sr)
    assert so["iterations"] == 100000
    assert np.array_equal(ko, kr)
    assert np.abs(np.asarray(mo) - np.asarray(mr)).max() < 1e-9


@pytest.mark.skipif(not ref_lib.available("fma"), reason="oracle/_ref/fma not built and /root/reference absent")
def test_relative_pose_gauge_reference_against_itself():
    """re-measures a slice of make_gauge.py's problems: same outcome, R / direction to 1e-9, |t| inside the frozen spread"""
    m = measure(60)
    assert m["identical_outcome"] == 60
    assert m["max_dR"] < DIR_BOUND and m["max_dt_dir"] < DIR_BOUND
    assert m["max_dt_len"] <= 10.0 * GAUGE["max_dt_len"]  # (this slice is part of the frozen set)
    assert GAUGE["problems"] >= 1000 and GAUGE["identical_outcome"] == GAUGE["problems"]


def test_relative_pose_oracle_against_reference_components():
    """the oracle (which the HIP path equals) against the reference's sources, component by component"""
    worst = {"dR": 0.0, "dt_dir": 0.0, "dt_len": 0.0}
    for i, d, opt in problems(80, first=2000):
        mo, ko, so = O.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
        with ref_lib.reference():
            mr, kr, sr = O.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
        for k in ("iterations", "refinements", "num_inliers"):
            assert so[k] == sr[k], (i, k, so, sr)
        assert np.array_equal(ko, kr), i
        for k, v in parts(mo, mr).items():
            worst[k] = max(worst[k], v)
    assert worst["dR"] < DIR_BOUND and worst["dt_dir"] < DIR_BOUND, worst
    assert worst["dt_len"] <= DT_LEN_BOUND, worst
