"""BASELINE config 3 "OPENCV camera model": the homography / 7-point estimators take no camera (robust.h:112-113,
133-134), so pixels of a distorting camera are un-distorted first - Camera::unproject per point
(misc/camera_models.cc:1025-1032, iterative inverse :972-990) - as a stage of its own on the device
(pl_undistort_points), then fed to estimate_homography / estimate_fundamental.  Checked against the oracle's unproject
and the oracle's estimators on the same un-distorted points."""
import numpy as np
import pytest

import oracle_lib as O
from poselib_amd import synth

pytestmark = pytest.mark.gpu

OPENCV = {"model": "OPENCV", "width": 1000, "height": 1000,
          "params": [1000.0, 1010.0, 500.0, 505.0, 0.04, -0.015, 8e-4, -6e-4]}


def _pinhole_pixels(cam, un):
    fx, fy, cx, cy = cam["params"][:4]
    return np.c_[fx * un[:, 0] + cx, fy * un[:, 1] + cy]


@pytest.mark.parametrize("kind", ["hom", "fund"])
def test_undistort_stage_in_front_of_the_two_view_estimators(gpu, kind):
    d = (synth.homography_scene if kind == "hom" else synth.fundamental_scene)(10000, 0.5, 1003 if kind == "hom" else 1004)
    # the scene's pinhole pixels seen through the distorting camera
    pin = {"model": "PINHOLE", "params": [1000.0, 1000.0, 500.0, 500.0]}
    x1d = synth.opencv_distort_pixels(_pinhole_pixels(OPENCV, O.unproject(pin, d["x1"])), OPENCV["params"])
    x2d = synth.opencv_distort_pixels(_pinhole_pixels(OPENCV, O.unproject(pin, d["x2"])), OPENCV["params"])
    u1, u2 = gpu.undistort_points(OPENCV, x1d), gpu.undistort_points(OPENCV, x2d)
    r1, r2 = _pinhole_pixels(OPENCV, O.unproject(OPENCV, x1d)), _pinhole_pixels(OPENCV, O.unproject(OPENCV, x2d))
    err = max(np.abs(u1 - r1).max(), np.abs(u2 - r2).max())
    print(kind, "max |device - oracle| of the un-distorted pixels:", err)
    assert err == 0.0  # the same IEEE operations in the same order (no libm call besides sqrt)
    # the distortion was real and the stage removes it
    assert np.abs(x1d - _pinhole_pixels(OPENCV, O.unproject(pin, d["x1"]))).max() > 1.0
    assert np.abs(u1 - _pinhole_pixels(OPENCV, O.unproject(pin, d["x1"]))).max() < 1e-6
    opt = {"ransac": {"seed": 2}}
    if kind == "hom":
        M, info = gpu.estimate_homography(u1, u2, opt)
        Mo, mask, st = O.estimate_homography(r1, r2, opt)
    else:
        M, info = gpu.estimate_fundamental(u1, u2, opt)
        Mo, mask, st = O.estimate_fundamental(r1, r2, opt)
    assert info["iterations"] == st["iterations"] and info["refinements"] == st["refinements"]
    assert info["num_inliers"] == st["num_inliers"] and (np.array(info["inliers"]) == mask).all()
    assert info["num_inliers"] > 2500
    A, B = M / np.linalg.norm(M), Mo / np.linalg.norm(Mo)
    assert np.linalg.norm(A - B) < 1e-6  # sign included


def test_undistort_rejects_what_it_does_not_cover(gpu):
    import poselib_amd as P

    with pytest.raises(P.PoseLibAmdError):
        P.undistort_points({"model": "NULL", "params": []}, np.zeros((4, 2)))
    assert P.undistort_points(OPENCV, np.zeros((0, 2))).shape == (0, 2)
    p = np.array([[500.0, 505.0], [10.0, 990.0]])
    sp = {"model": "SIMPLE_PINHOLE", "params": [800.0, 400.0, 300.0]}
    assert np.abs(P.undistort_points(sp, p) - p).max() < 1e-12  # a linear camera maps onto itself
