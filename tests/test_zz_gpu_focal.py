"""GPU parity (-m gpu) of the focal-length estimator - SURVEY §8 (f4): pl_ransac_pnpf and pl_estimate_absolute_pose with
estimate_focal_length (robust.cc:47-54 -> ransac.cc:58-75 -> FocalAbsolutePoseEstimator) through the C-ABI against the oracle.

The chain of evidence: the oracle's ransac_pnpf takes the decisions of the REFERENCE's estimator on the pinned scenes
(tests/test_reference_focal_estimator.py; since round 6 its P3.5Pf restates the reference's template: the same solutions bit for
bit); the device
functions equal the oracle bit for bit on the host (tests/test_hostmath_vs_oracle.py: solver, loop); here the kernels themselves.
Up to 256 correspondences k_lm_cam sums the cost in the reference's order and everything is bit for bit; beyond, its cost is
a tree sum, so a refined model may differ in the last bits: decisions (iterations, refinements, inlier mask) are still demanded
exactly, the returned pose / focal length to 1e-9.

(The file name sorts last on purpose: the path is new in round 3 and a failure here must not hide the rest of the suite behind -x.)
"""
import time

import numpy as np
import pytest

import oracle_lib as O
from poselib_amd import synth

pytestmark = pytest.mark.gpu


def _centered(d):
    f, cx, cy = d["camera"]["params"]
    return np.asarray(d["p2d"]) - np.array([cx, cy]), f


def _check(tag, img, info, ref, exact):
    rpose, rfocal, rmask, rst = ref
    for k in ("iterations", "refinements", "num_inliers", "hypotheses"):
        assert info[k] == rst[k], (tag, k, info[k], rst[k])
    assert np.array_equal(np.asarray(info["inliers"], dtype=bool), rmask), (tag, int((np.asarray(info["inliers"], dtype=bool) != rmask).sum()))
    pose = np.r_[img.pose.q, img.pose.t]
    if exact:
        assert info["model_score"] == rst["model_score"], (tag, info["model_score"], rst["model_score"])
        assert np.array_equal(pose, rpose) and img.camera.params[0] == rfocal, (tag, np.abs(pose - rpose).max(), img.camera.params[0] - rfocal)
    else:
        assert abs(info["model_score"] - rst["model_score"]) <= 1e-9 * abs(rst["model_score"]), (tag, info["model_score"], rst["model_score"])
        assert np.abs(pose - rpose).max() < 1e-9 and abs(img.camera.params[0] - rfocal) < 1e-9 * rfocal, tag


@pytest.mark.parametrize("n", [40, 200, 256])
def test_ransac_pnpf_bit_exact_up_to_256_correspondences(gpu, n):
    for k in range(4):
        d = synth.absolute_pose_scene(n, [0.2, 0.4, 0.5, 0.3][k], 8600 + 10 * n + k, noise_px=0.5)
        x, f = _centered(d)
        opt = {"max_error": 4.0, "ransac": {"seed": 11 + k}}
        ref = O.ransac_pnpf(x, d["p3d"], opt)
        img, info = gpu.ransac_pnpf(x, d["p3d"], opt)
        _check((n, k), img, info, ref, exact=True)
        assert abs(img.camera.params[0] - f) / f < 0.05


@pytest.mark.parametrize("n,outliers", [(800, 0.3), (2000, 0.5), (5000, 0.6)])
def test_ransac_pnpf_larger_problems(gpu, n, outliers):
    d = synth.absolute_pose_scene(n, outliers, 8700 + n, noise_px=0.5)
    x, f = _centered(d)
    opt = {"max_error": 4.0, "ransac": {"seed": n}}
    ref = O.ransac_pnpf(x, d["p3d"], opt)
    img, info = gpu.ransac_pnpf(x, d["p3d"], opt)
    _check(n, img, info, ref, exact=True)  # (round 4: k_lm_cam sums its cost in the reference's order at every n)
    assert abs(img.camera.params[0] - f) / f < 2e-3


def test_ransac_pnpf_iteration_budgets_and_degenerate_sizes(gpu):
    d = synth.absolute_pose_scene(240, 0.1, 8800, noise_px=0.3)
    x, _ = _centered(d)
    for ro in ({"min_iterations": 10, "max_iterations": 5000, "seed": 5}, {"min_iterations": 0, "max_iterations": 37, "seed": 6},
               {"min_iterations": 300, "max_iterations": 300, "seed": 7}, {"min_iterations": 4500, "max_iterations": 9000, "seed": 8}):
        opt = {"max_error": 3.0, "ransac": ro}
        ref = O.ransac_pnpf(x, d["p3d"], opt)
        img, info = gpu.ransac_pnpf(x, d["p3d"], opt)
        _check(ro, img, info, ref, exact=True)
    ref = O.ransac_pnpf(x[:3], d["p3d"][:3], {"max_error": 3.0})
    img, info = gpu.ransac_pnpf(x[:3], d["p3d"][:3], {"max_error": 3.0})
    assert info["iterations"] == 0 and img.camera.params[0] == 1.0 and np.array_equal(np.r_[img.pose.q, img.pose.t], ref[0])
    assert np.array_equal(np.asarray(info["inliers"], dtype=bool), ref[2])
    # PROSAC (sampling.cc:85-136; absolute_pose.h:80 constructs the sampler from opt.ransac): host-drawn samples, the oracle's decisions
    for ro in ({"seed": 2, "progressive_sampling": True}, {"seed": 3, "progressive_sampling": True, "max_prosac_iterations": 300}):
        opt = {"max_error": 3.0, "ransac": ro}
        ref = O.ransac_pnpf(x, d["p3d"], opt)
        img, info = gpu.ransac_pnpf(x, d["p3d"], opt)
        _check(ro, img, info, ref, exact=True)
    # min_fov (types.h:126, absolute_pose.h:78): the bound on the focal length is an option, not a constant; <= 0 disables it
    for fov in (5.0, 40.0, 0.0):
        opt = {"max_error": 3.0, "min_fov": fov, "ransac": {"seed": 11}}
        ref = O.ransac_pnpf(x, d["p3d"], opt)
        img, info = gpu.ransac_pnpf(x, d["p3d"], opt)
        _check(("min_fov", fov), img, info, ref, exact=True)
    with pytest.raises(gpu.PoseLibAmdError):  # the calibrated entry point does not estimate focal lengths
        gpu.ransac_pnp(x, d["p3d"], {"estimate_focal_length": True})


@pytest.mark.parametrize("model", ["SIMPLE_PINHOLE", "PINHOLE"])
def test_estimate_absolute_pose_with_estimate_focal_length(gpu, model):
    """robust.cc:36-126 with opt.estimate_focal_length: the camera comes in with a focal length that is 30 % off, RANSAC estimates
    pose and focal length, the final bundle refines both (refine_focal_length forced)"""
    for k, (n, outl) in enumerate([(220, 0.3), (1500, 0.4)]):
        d = synth.absolute_pose_scene(n, outl, 8900 + k, noise_px=0.5)
        f, cx, cy = d["camera"]["params"]
        cam0 = dict(d["camera"], params=[1.3 * f, cx, cy]) if model == "SIMPLE_PINHOLE" else \
            dict(d["camera"], model="PINHOLE", params=[1.3 * f, 1.3 * f, cx, cy])
        opt = {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": 4 + k}}
        rpose, rmask, rst, rcam = O.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, opt, return_camera=True)
        img, info = gpu.estimate_absolute_pose(d["p2d"], d["p3d"], cam0, opt)
        for key in ("iterations", "refinements", "num_inliers"):
            assert info[key] == rst[key], (model, k, key, info[key], rst[key])
        assert np.array_equal(np.asarray(info["inliers"], dtype=bool), rmask)
        assert np.abs(np.r_[img.pose.q, img.pose.t] - rpose).max() < 1e-8
        assert np.abs(np.asarray(img.camera.params) - rcam).max() < 1e-8 * f
        assert abs(img.camera.focal() - f) / f < 2e-3
        # the same problem inside a batch call: the item runs through the single-problem path, same result
        res = gpu.estimate_batch([("abs", d["p2d"], d["p3d"], cam0, opt), ("abs", d["p2d"], d["p3d"], d["camera"], {"max_error": 4.0})])
        assert np.array_equal(np.r_[res[0][0].pose.q, res[0][0].pose.t], np.r_[img.pose.q, img.pose.t])
        assert np.array_equal(res[0][0].camera.params, img.camera.params) and res[0][1]["inliers"] == info["inliers"]


def test_ransac_pnpf_speed_against_the_oracle(gpu, capsys):
    """not a parity test: wall time per problem, device against the oracle on this host (printed; DESIGN §6 quotes it)"""
    d = synth.absolute_pose_scene(2000, 0.5, 8950, noise_px=0.5)
    x, _ = _centered(d)
    opt = {"max_error": 4.0, "ransac": {"seed": 1, "min_iterations": 10000, "max_iterations": 10000}}
    gpu.ransac_pnpf(x, d["p3d"], opt)
    t0 = time.time()
    img, info = gpu.ransac_pnpf(x, d["p3d"], opt)
    t_gpu = time.time() - t0
    t0 = time.time()
    ref = O.ransac_pnpf(x, d["p3d"], opt)
    t_cpu = time.time() - t0
    _check("speed", img, info, ref, exact=True)
    with capsys.disabled():
        print(f"\n[focal] n=2000, 10000 iterations, {info['hypotheses']} hypotheses: device {t_gpu * 1e3:.1f} ms, oracle {t_cpu * 1e3:.1f} ms "
              f"({t_cpu / t_gpu:.1f}x)")
    assert info["iterations"] == 10000
