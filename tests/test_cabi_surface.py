"""The C-ABI library loads and exports every symbol include/poselib_amd.h declares; option defaults
match the reference's (types.h:39-175); without a GPU the compute entry points fail loudly
(no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import poselib_amd
from poselib_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "poselib_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    poselib_amd.build()
    lib = L.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert set(L.EXPORTED_SYMBOLS) <= set(names)


def test_default_options_match_reference():
    o = L.RobustOptions()
    L.lib().pl_default_robust_options(C.byref(o), 0)
    assert (o.ransac.max_iterations, o.ransac.min_iterations) == (100000, 1000)  # types.h:40-41
    assert (o.ransac.dyn_num_trials_mult, o.ransac.success_prob, o.ransac.seed) == (3.0, 0.9999, 0)
    assert o.ransac.progressive_sampling == 0 and o.ransac.max_prosac_iterations == 100000
    assert o.ransac.score_initial_model == 0
    assert o.bundle.max_iterations == 100 and o.bundle.loss_type == 3 and o.bundle.loss_scale == 1.0  # CAUCHY
    assert (o.bundle.gradient_tol, o.bundle.step_tol, o.bundle.relative_cost_tol) == (1e-12, 1e-8, 1e-10)
    assert (o.bundle.initial_lambda, o.bundle.min_lambda, o.bundle.max_lambda) == (1e-3, 1e-10, 1e10)
    assert o.bundle.lambda_factor == 10.0 and o.bundle.lambda_update == 0 and o.bundle.damping == 0
    assert o.max_error == 12.0  # types.h:113
    for kind in (1, 2, 3):
        L.lib().pl_default_robust_options(C.byref(o), kind)
        assert o.max_error == 1.0  # types.h:134,174


def test_python_surface_mirrors_reference_module():
    for name in ["estimate_absolute_pose", "estimate_relative_pose", "estimate_fundamental", "estimate_homography",
                 "p3p", "relpose_5pt", "essential_matrix_5pt", "relpose_7pt", "homography_4pt", "CameraPose", "Camera",
                 "Image", "RansacOptions", "BundleOptions"]:
        assert hasattr(poselib_amd, name), name
    assert set(poselib_amd.RansacOptions()) == {"max_iterations", "min_iterations", "dyn_num_trials_mult",
                                                "success_prob", "seed", "progressive_sampling",
                                                "max_prosac_iterations"}  # helpers.h:31-40 ('score_initial_model' omitted)
    p = poselib_amd.CameraPose()
    assert p.q.tolist() == [1.0, 0.0, 0.0, 0.0] and p.R.shape == (3, 3) and p.Rt.shape == (3, 4)
    cam = poselib_amd.Camera({"model": "SIMPLE_PINHOLE", "width": 1000, "height": 1000, "params": [1000, 500, 500]})
    assert cam.focal() == 1000.0 and cam.todict()["model"] == "SIMPLE_PINHOLE"


@pytest.mark.skipif(poselib_amd.device_count() > 0, reason="GPU present")
def test_no_gpu_means_loud_failure_not_fallback():
    pts2 = np.zeros((10, 2))
    pts3 = np.zeros((10, 3))
    with pytest.raises(poselib_amd.PoseLibAmdError) as e:
        poselib_amd.estimate_absolute_pose(pts2, pts3, {"model": "SIMPLE_PINHOLE", "params": [1.0, 0.0, 0.0]})
    assert "no HIP device" in str(e.value)
    with pytest.raises(poselib_amd.PoseLibAmdError):
        poselib_amd.p3p(np.eye(3), np.eye(3))
    # the batch entry points, the multi-device one included (round 6): the same loud failure, and a report that says nothing ran
    cam = {"model": "SIMPLE_PINHOLE", "params": [1.0, 0.0, 0.0]}
    for devices in (None, [0], "all"):
        with pytest.raises(poselib_amd.PoseLibAmdError) as e:
            poselib_amd.estimate_batch([("abs", pts2, pts3, cam, {})], devices=devices)
        assert "no HIP device" in str(e.value)
    with pytest.raises(poselib_amd.PoseLibAmdError):
        poselib_amd.p35pf(np.zeros((4, 2)), np.zeros((4, 3)))


def test_unsupported_options_are_rejected():
    o = L.RobustOptions()
    L.lib().pl_default_robust_options(C.byref(o), 0)
    o.tangent_sampson = 1  # the tangent-Sampson camera estimator is outside the accelerated path
    st = L.RansacStats()
    pose = L.CameraPose()
    rc = L.lib().pl_ransac_pnp(None, None, C.c_size_t(0), C.byref(o), C.byref(pose), None, C.byref(st))
    assert rc == L.PL_ERR_UNSUPPORTED
