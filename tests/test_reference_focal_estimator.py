"""SURVEY §8 (f4), VERDICT r2 "pins first": the reference's OWN focal-length estimator, runnable here.

`oracle/Makefile.ref` now compiles `solvers/p35pf.cc` (the default solver of `FocalAbsolutePoseEstimator`,
robust/estimators/absolute_pose.h:71-113) with the rest of the reference's sources; what it needed from Eigen -
`householderQr().householderQ()`, `EigenSolver(A, false).eigenvalues()`, linear indexing of a matrix - is in
`oracle/eigen_shim` (Hessenberg reduction + Francis double-shift QR; agrees with LAPACK to 1e-13 on random 10 x 10
matrices).  `ref_p35pf`, `ref_ransac_pnpf`, `estimate_absolute_pose` with `estimate_focal_length` and
`ref_estimate_shared_focal_relative_pose` (solvers/relpose_6pt_focal.cc) run the REFERENCE's code
(robust/ransac.cc:58-75, robust.cc:47-54, estimators/absolute_pose.cc:73-160, bundle with refine_focal_length).

There is NO oracle restatement and NO device path for these estimators (P3.5Pf is a machine-generated elimination
template); `poselib_amd` answers `estimate_focal_length` with PL_ERR_UNSUPPORTED.  These tests fix the reference's behaviour on
synthetic data so that a later restatement has something to be held against."""
import numpy as np
import pytest

import oracle_lib as O
import ref_lib
from poselib_amd import synth

pytestmark = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built and /root/reference absent")


def _centered(d):
    f, cx, cy = d["camera"]["params"]
    return np.asarray(d["p2d"]) - np.array([cx, cy]), f


def _pose_error(pose7, d):  # (q and -q are the same rotation)
    return max(np.abs(synth.quat_to_rotmat(np.asarray(pose7[:4])) - synth.quat_to_rotmat(np.asarray(d["q_gt"]))).max(),
               np.abs(np.asarray(pose7[4:]) - d["t_gt"]).max())


def test_the_oracle_says_that_it_has_no_focal_estimator():
    d = synth.absolute_pose_scene(50, 0.0, 1)
    x, _ = _centered(d)
    with pytest.raises(RuntimeError):
        O.ransac_pnpf(x, d["p3d"])


@pytest.mark.parametrize("seed", range(8))
def test_p35pf_of_the_reference_on_exact_data(seed):
    d = synth.absolute_pose_scene(4, 0.0, 8100 + seed, noise_px=0.0)
    x, f = _centered(d)
    with ref_lib.reference():
        poses, focals = O.p35pf(x, d["p3d"])
    assert 1 <= len(focals) <= 10
    Rgt = synth.quat_to_rotmat(np.asarray(d["q_gt"]))
    err = [max(abs(fo - f) / f, np.abs(synth.quat_to_rotmat(p[:4]) - Rgt).max(), np.abs(p[4:] - d["t_gt"]).max())
           for p, fo in zip(poses, focals)]
    assert min(err) < 1e-7, (focals, min(err))


@pytest.mark.parametrize("seed,outliers", [(0, 0.3), (1, 0.5), (2, 0.6)])
def test_ransac_pnpf_of_the_reference_recovers_pose_and_focal_length(seed, outliers):
    d = synth.absolute_pose_scene(800, outliers, 8200 + seed, noise_px=0.5)
    x, f = _centered(d)
    with ref_lib.reference():
        pose, focal, mask, st = O.ransac_pnpf(x, d["p3d"], {"max_error": 4.0, "ransac": {"seed": seed}})
    assert abs(focal - f) / f < 2e-3
    assert _pose_error(pose, d) < 5e-3
    assert (mask & d["inlier_gt"]).sum() >= 0.97 * d["inlier_gt"].sum() and (mask & ~d["inlier_gt"]).sum() <= 3
    assert st["iterations"] >= 1000 and st["refinements"] >= 1


def test_estimate_absolute_pose_with_estimate_focal_length_through_the_reference():
    """robust.cc:47-54: ransac_pnpf on the un-projected points, then the final bundle with refine_focal_length forced"""
    d = synth.absolute_pose_scene(800, 0.4, 8300, noise_px=0.5)
    f, cx, cy = d["camera"]["params"]
    cam0 = dict(d["camera"], params=[1.3 * f, cx, cy])  # a focal length 30 % off: the estimator must not care
    opt = {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": 4}}
    with ref_lib.reference():
        pose, mask, st, cam = O.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, opt, return_camera=True)
        pose2, mask2, st2, cam2 = O.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, dict(opt, estimate_focal_length=False), return_camera=True)
    assert abs(cam[0] - f) / f < 1e-3 and np.allclose(cam[1:], [cx, cy], rtol=1e-12, atol=0)  # (rescale round trip)
    assert _pose_error(pose, d) < 5e-3
    assert mask.sum() >= 0.97 * d["inlier_gt"].sum()
    assert cam2[0] == pytest.approx(1.3 * f, rel=1e-12) and mask2.sum() < mask.sum()  # without the option the wrong focal stays


@pytest.mark.parametrize("seed,outliers", [(0, 0.3), (1, 0.5)])
def test_shared_focal_relative_pose_of_the_reference(seed, outliers):
    """robust.cc:366-430 estimate_shared_focal_relative_pose -> ransac_shared_focal_relpose (ransac.cc:183-197) ->
    SharedFocalRelativePoseEstimator with relpose_6pt_shared_focal (generated template + Sturm chain of degree 15) and the
    shared-focal refiner: the reference's sources, compiled since round 3 (the shim's HouseholderQR::solve, MatrixBase, ...)"""
    d = synth.relative_pose_scene(1200, outliers, 8400 + seed)
    f, cx, cy = d["camera1"]["params"]
    with ref_lib.reference():
        pose, focal, mask, st = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], {"max_error": 2.0, "ransac": {"seed": seed}})
    assert abs(focal - f) / f < 5e-3
    t_gt = d["t_gt"] / np.linalg.norm(d["t_gt"])
    t = pose[4:] / np.linalg.norm(pose[4:])
    assert np.abs(synth.quat_to_rotmat(pose[:4]) - synth.quat_to_rotmat(np.asarray(d["q_gt"]))).max() < 5e-3
    assert np.abs(t - t_gt).max() < 5e-3
    assert (mask & d["inlier_gt"]).sum() >= 0.97 * d["inlier_gt"].sum() and (mask & ~d["inlier_gt"]).sum() <= 0.02 * len(mask)
    with pytest.raises(RuntimeError):
        O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy])  # (the oracle has no such estimator)
