"""GPU parity (-m gpu) of the shared-focal relative pose estimator - SURVEY §8 (f4): pl_ransac_shared_focal_relpose,
pl_estimate_shared_focal_relative_pose (robust.cc:366-424 -> ransac.cc:182-203 -> SharedFocalRelativePoseEstimator) and
pl_refine_shared_focal_relpose (bundle.cc:281-297) through the C-ABI against the oracle.

The chain of evidence: the oracle's estimator takes the decisions of the REFERENCE's on the pinned scenes and its refiner equals
the reference's bit for bit (tests/test_reference_focal_estimator.py; since round 6 the 6-point solver restates the reference's
template: the same solutions bit for bit, up to the rounding of the cubes); the
device functions equal the oracle bit for bit on the host (tests/test_hostmath_vs_oracle.py: solver, refiner, loop); here the
kernels themselves.  k_sfocal_score and k_sfocal_lm add every sum correspondence after correspondence, so EVERYTHING is demanded
bit for bit at every size: decisions, score, inlier mask, pose, focal length.

(The file name sorts last on purpose: a failure in this new path must not hide the rest of the suite behind -x.)
"""
import time

import numpy as np
import pytest

import oracle_lib as O
from poselib_amd import synth

pytestmark = pytest.mark.gpu


def _centered(d, s=1.0):
    f, cx, cy = d["camera1"]["params"]
    return (np.asarray(d["x1"]) - [cx, cy]) / s, (np.asarray(d["x2"]) - [cx, cy]) / s, f / s


def _check(tag, pair, info, ref, decisions_only=False):
    rpose, rfocal, rmask, rst = ref
    for k in ("iterations", "refinements", "num_inliers"):
        assert info[k] == rst[k], (tag, k, info[k], rst[k])
    assert np.array_equal(np.asarray(info["inliers"], dtype=bool), rmask), tag
    assert info["model_score"] == rst["model_score"], (tag, info["model_score"], rst["model_score"])
    pose = np.r_[pair.pose.q, pair.pose.t]
    assert np.array_equal(pose, rpose), (tag, np.abs(pose - rpose).max())
    assert pair.camera1.params[0] == rfocal and pair.camera2.params[0] == rfocal, (tag, pair.camera1.params[0] - rfocal)


@pytest.mark.parametrize("n", [30, 200, 256, 700, 2000])
def test_ransac_shared_focal_relpose_bit_exact(gpu, n):
    for k in range(3):
        d = synth.relative_pose_scene(n, [0.2, 0.4, 0.5][k], 8800 + 10 * n + k, noise_px=0.5)
        a, b, f = _centered(d, 500.0)
        opt = {"max_error": 2.0 / 500.0, "ransac": {"seed": 5 + k, "max_iterations": 4000}}
        ref = O.ransac_shared_focal_relpose(a, b, opt)
        pair, info = gpu.ransac_shared_focal_relpose(a, b, opt)
        _check((n, k), pair, info, ref)
        if k < 2 and n >= 200:
            assert abs(pair.camera1.params[0] - f) / f < 0.05


def test_ransac_shared_focal_options_and_initial_model(gpu):
    d = synth.relative_pose_scene(600, 0.3, 8900, noise_px=0.7)
    a, b, f = _centered(d, 400.0)
    for ro in ({"seed": 3, "max_iterations": 700, "min_iterations": 50}, {"seed": 4, "min_iterations": 10, "success_prob": 0.9, "dyn_num_trials_mult": 1.0},
               {"seed": 5, "max_iterations": 2000, "min_iterations": 1500}):
        opt = {"max_error": 1.5 / 400.0, "ransac": ro}
        ref = O.ransac_shared_focal_relpose(a, b, opt)
        pair, info = gpu.ransac_shared_focal_relpose(a, b, opt)
        _check(("opt", ro["seed"]), pair, info, ref)
    # score_initial_model: a model near the truth, the focal length 10 % off
    init = gpu.ImagePair(gpu.CameraPose(d["q_gt"], d["t_gt"]), gpu.Camera("SIMPLE_PINHOLE", [1.1 * f, 0.0, 0.0]))
    opt = {"max_error": 1.5 / 400.0, "ransac": {"seed": 8, "max_iterations": 1200, "score_initial_model": True}}
    ref = O.ransac_shared_focal_relpose(a, b, opt, init_pose=np.r_[d["q_gt"], d["t_gt"]], init_focal=1.1 * f)
    pair, info = gpu.ransac_shared_focal_relpose(a, b, {"max_error": 1.5 / 400.0, "ransac": {"seed": 8, "max_iterations": 1200}}, initial_pair=init)
    _check("initial", pair, info, ref)
    # fewer correspondences than a sample: nothing runs (ransac_impl.h:87-91)
    pair, info = gpu.ransac_shared_focal_relpose(a[:5], b[:5], {"max_error": 0.01})
    ref = O.ransac_shared_focal_relpose(a[:5], b[:5], {"max_error": 0.01})
    assert info["iterations"] == ref[3]["iterations"] == 0 and info["num_inliers"] == 0


@pytest.mark.parametrize("seed,outliers,n", [(0, 0.3, 1200), (1, 0.5, 1200), (2, 0.2, 300), (3, 0.4, 3000)])
def test_estimate_shared_focal_relative_pose_front_end(gpu, seed, outliers, n):
    """robust.cc:366-424: principal point removed, shared scale normalisation, RANSAC, final refinement on the inliers, focal
    length back in pixels"""
    d = synth.relative_pose_scene(n, outliers, 8400 + seed)
    f, cx, cy = d["camera1"]["params"]
    opt = {"max_error": 2.0, "ransac": {"seed": seed}}
    t0 = time.perf_counter()
    pair, info = gpu.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], opt)
    t_dev = time.perf_counter() - t0
    ref = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], opt)
    _check(("front", seed), pair, info, ref)
    assert pair.camera1.params[1:] == [cx, cy]
    assert abs(pair.camera1.params[0] - f) / f < 1e-2
    print(f"\n[shared focal] n={n}, {info['iterations']} iterations, {info['hypotheses']} hypotheses: device {1e3 * t_dev:.1f} ms, "
          f"oracle {1e3 * ref[3]['seconds']:.1f} ms")
    # the LOSS options of the final bundle reach the refiner
    for loss in ("HUBER", "TRUNCATED_LE_ZACH"):
        o2 = dict(opt, bundle={"loss_type": loss, "loss_scale": 1.5, "max_iterations": 40})
        pair, info = gpu.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], o2)
        _check(("front", seed, loss), pair, info, O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], o2))


@pytest.mark.parametrize("n", [20, 256, 1500])
def test_refine_shared_focal_relpose_bit_exact(gpu, n):
    d = synth.relative_pose_scene(n, 0.15, 9100 + n, focal=900.0)
    a, b, f = _centered(d, 700.0)
    q = np.r_[d["q_gt"], d["t_gt"]] + 0.01 * np.random.default_rng(n).normal(size=7)
    q[:4] /= np.linalg.norm(q[:4])
    for loss in ("TRIVIAL", "TRUNCATED", "HUBER", "CAUCHY", "TRUNCATED_CAUCHY", "TRUNCATED_LE_ZACH"):
        bo = {"loss_type": loss, "loss_scale": 0.003, "max_iterations": 30}
        po, fo, so = O.refine_shared_focal_relpose(a, b, q, 1.1 * f, bo)
        pair, its = gpu.refine_shared_focal_relpose(a, b, gpu.ImagePair(gpu.CameraPose(q[:4], q[4:]), gpu.Camera("SIMPLE_PINHOLE", [1.1 * f, 0, 0])), bo)
        assert its == so.iterations, (n, loss, its, so.iterations)
        assert np.array_equal(np.r_[pair.pose.q, pair.pose.t], po) and pair.camera1.params[0] == fo, (n, loss)


def test_shared_focal_rejections(gpu):
    d = synth.relative_pose_scene(100, 0.2, 9200)
    a, b, f = _centered(d, 500.0)
    # PROSAC is served (host-drawn samples, relative_pose.h:155): the oracle's run bit for bit
    opt = {"max_error": 0.01, "ransac": {"progressive_sampling": True, "seed": 4, "max_iterations": 2000}}
    po, fo, mo, so = O.ransac_shared_focal_relpose(a, b, opt)
    pair, info = gpu.ransac_shared_focal_relpose(a, b, opt)
    assert (info["iterations"], info["refinements"], info["num_inliers"]) == (so["iterations"], so["refinements"], so["num_inliers"])
    assert np.array_equal(np.r_[pair.pose.q, pair.pose.t], po) and pair.camera1.params[0] == fo
    assert np.array_equal(np.asarray(info["inliers"], dtype=bool), mo)
    with pytest.raises(gpu.PoseLibAmdError):
        gpu.ransac_shared_focal_relpose(a, b, {"max_error": 0.01, "tangent_sampson": True})


def test_front_end_with_an_initial_pair_and_minimal_inputs(gpu):
    """robust.cc:392-397: with score_initial_model the pair's focal length is divided by the normalisation scale before the
    RANSAC and the initial model competes; 6, 7 and 10 correspondences (a sample is 6)"""
    d = synth.relative_pose_scene(800, 0.3, 9300, noise_px=0.6)
    f, cx, cy = d["camera1"]["params"]
    init = gpu.ImagePair(gpu.CameraPose(d["q_gt"], d["t_gt"]), gpu.Camera("SIMPLE_PINHOLE", [1.1 * f, cx, cy]))
    ro = {"seed": 21, "max_iterations": 1500}
    ref = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], {"max_error": 2.0, "ransac": dict(ro, score_initial_model=True)},
                                                init_pose=np.r_[d["q_gt"], d["t_gt"]], init_focal=1.1 * f)
    pair, info = gpu.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], {"max_error": 2.0, "ransac": ro}, initial_pair=init)
    _check("front initial", pair, info, ref)
    for n in (6, 7, 10):
        opt = {"max_error": 2.0, "ransac": {"seed": n, "max_iterations": 300, "min_iterations": 20}}
        pair, info = gpu.estimate_shared_focal_relative_pose(d["x1"][:n], d["x2"][:n], [cx, cy], opt)
        _check(("front", n), pair, info, O.estimate_shared_focal_relative_pose(d["x1"][:n], d["x2"][:n], [cx, cy], opt))


def test_hostile_inputs_return(gpu):
    """NaN / infinite coordinates, every correspondence the same point, a planar scene: the calls return (bounded loops everywhere:
    a hang would cost the GPU) and take the oracle's decisions"""
    d = synth.relative_pose_scene(300, 0.2, 9400)
    f, cx, cy = d["camera1"]["params"]
    a, b = (np.asarray(d["x1"]) - [cx, cy]) / 500.0, (np.asarray(d["x2"]) - [cx, cy]) / 500.0
    opt = {"max_error": 0.004, "ransac": {"seed": 2, "max_iterations": 400, "min_iterations": 50}}
    hostile = {"nan": (np.where(np.arange(300)[:, None] % 7 == 0, np.nan, a), b), "inf": (a, np.where(np.arange(300)[:, None] % 11 == 0, np.inf, b)),
               "one point": (np.tile(a[:1], (300, 1)), np.tile(b[:1], (300, 1))), "identical views": (a, a)}
    h = synth.homography_scene(300, 0.1, 9401)
    hostile["planar"] = ((np.asarray(h["x1"]) - 500.0) / 500.0, (np.asarray(h["x2"]) - 500.0) / 500.0)
    with np.errstate(all="ignore"):
        for name, (x1, x2) in hostile.items():
            pair, info = gpu.ransac_shared_focal_relpose(x1, x2, opt)
            ref = O.ransac_shared_focal_relpose(x1, x2, opt)
            assert info["iterations"] == ref[3]["iterations"] and info["num_inliers"] == ref[3]["num_inliers"], (name, info["iterations"], ref[3])
            assert np.array_equal(np.asarray(info["inliers"], dtype=bool), ref[2]), name


def test_batched_front_end_serves_both_focal_estimators(gpu):
    """pl_estimate_batch: shared-focal pairs (item kind 4) and absolute-pose items with estimate_focal_length next to ordinary
    items - every result equals the single-problem call's bit for bit (they are served by the pool one problem at a time)"""
    probs, singles = [], []
    for k in range(6):
        d = synth.relative_pose_scene(400 + 100 * k, 0.3, 9500 + k)
        f, cx, cy = d["camera1"]["params"]
        opt = {"max_error": 2.0, "ransac": {"seed": k}}
        probs.append(("shared_focal", d["x1"], d["x2"], [cx, cy], opt))
        singles.append(gpu.estimate_shared_focal_relative_pose(d["x1"], d["x2"], [cx, cy], opt))
        da = synth.absolute_pose_scene(300 + 100 * k, 0.3, 9600 + k)
        oa = {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": k}}
        probs.append(("abs", da["p2d"], da["p3d"], da["camera"], oa))
        singles.append(gpu.estimate_absolute_pose(da["p2d"], da["p3d"], da["camera"], oa))
        probs.append(("rel", d["x1"], d["x2"], d["camera1"], d["camera2"], {"max_error": 2.0, "ransac": {"seed": k}}))
        singles.append(gpu.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], {"max_error": 2.0, "ransac": {"seed": k}}))
    out = gpu.estimate_batch(probs, max_in_flight=4)
    for (model, info), (smodel, sinfo), pr in zip(out, singles, probs):
        for key in ("iterations", "refinements", "num_inliers", "model_score", "inliers"):
            assert info[key] == sinfo[key], (pr[0], key)
        if pr[0] == "shared_focal":
            assert np.array_equal(np.r_[model.pose.q, model.pose.t], np.r_[smodel.pose.q, smodel.pose.t])
            assert model.camera1.params == smodel.camera1.params and model.camera2.params == smodel.camera2.params
        elif pr[0] == "abs":
            assert np.array_equal(np.r_[model.pose.q, model.pose.t], np.r_[smodel.pose.q, smodel.pose.t]) and model.camera.params == smodel.camera.params
        else:
            assert np.array_equal(np.r_[model.q, model.t], np.r_[smodel.q, smodel.t])


def test_both_focal_solvers_on_the_device_equal_the_oracle_bit_for_bit(gpu):
    """pl_solve_focal_batch / pl_p35pf / pl_relpose_6pt_shared_focal: the generator kernels' solvers on explicit minimal problems -
    every pose, every focal length and their order equal the oracle's, whose solvers are the reference's bit for bit
    (tests/test_reference_focal_estimator.py; the six-point cubes correctly rounded on both sides); the golden solver vectors -
    generated by the reference's own sources - included"""
    import json
    import os

    from golden.make_golden_focal import minimal_inputs

    rng = np.random.default_rng(17)
    # 6-point: random two-view geometry at normalised focal lengths
    from scipy.spatial.transform import Rotation

    six = []
    for _ in range(600):
        f = rng.uniform(0.3, 3.0)
        X = rng.uniform(-1, 1, (6, 3)) * [2, 2, 1] + [0, 0, 5]
        R = Rotation.from_rotvec(rng.normal(size=3) * 0.2).as_matrix()
        t = rng.normal(size=3)
        X2 = X @ R.T + t / np.linalg.norm(t)
        b1 = np.c_[f * X[:, :2] / X[:, 2:], np.ones(6)]
        b2 = np.c_[f * X2[:, :2] / X2[:, 2:], np.ones(6)]
        six.append(np.r_[(b1 / np.linalg.norm(b1, axis=1)[:, None]).reshape(-1), (b2 / np.linalg.norm(b2, axis=1)[:, None]).reshape(-1)])
    six = np.array(six)
    models, counts = gpu.solve_focal_batch("relpose_6pt_shared_focal", six)
    total = 0
    for k in range(len(six)):
        po, fo = O.relpose_6pt_shared_focal(six[k, :18].reshape(6, 3), six[k, 18:].reshape(6, 3))
        assert counts[k] == len(fo), k
        assert np.array_equal(models[k, : counts[k], :7], po, equal_nan=True) and np.array_equal(models[k, : counts[k], 7], fo, equal_nan=True), k
        total += len(fo)
    assert total > 600
    # P3.5Pf: samples of noisy absolute-pose scenes
    p35 = []
    for k in range(150):
        d = synth.absolute_pose_scene(4 * 4, 0.0, 9700 + k, noise_px=[0.0, 1.0][k % 2], focal=float(rng.uniform(500, 2500)))
        fcam, cx, cy = d["camera"]["params"]
        x = np.asarray(d["p2d"]) - [cx, cy]
        for j in range(4):
            p35.append(np.r_[x[4 * j:4 * j + 4].reshape(-1), np.asarray(d["p3d"])[4 * j:4 * j + 4].reshape(-1)])
    p35 = np.array(p35)
    models, counts = gpu.solve_focal_batch("p35pf", p35)
    total = 0
    for k in range(len(p35)):
        po, fo = O.p35pf(p35[k, :8].reshape(4, 2), p35[k, 8:].reshape(4, 3))
        assert counts[k] == len(fo), k
        assert np.array_equal(models[k, : counts[k], :7], po, equal_nan=True) and np.array_equal(models[k, : counts[k], 7], fo, equal_nan=True), k
        total += len(fo)
    assert total > 600
    # large launches (several workgroups per CU, every LDS region of a workgroup reused by the next): the same problems, the p35 samples
    # re-drawn from their scenes, 4608 of each in one call - against the oracle again
    big6 = np.concatenate([six] * 8)[:4608]
    models, counts = gpu.solve_focal_batch("relpose_6pt_shared_focal", big6)
    ref6 = [O.relpose_6pt_shared_focal(six[k, :18].reshape(6, 3), six[k, 18:].reshape(6, 3)) for k in range(len(six))]
    for k in range(len(big6)):
        po, fo = ref6[k % len(six)]
        assert counts[k] == len(fo), k
        assert np.array_equal(models[k, : counts[k], :7], po, equal_nan=True) and np.array_equal(models[k, : counts[k], 7], fo, equal_nan=True), k
    big35 = []
    for k in range(144):
        d = synth.absolute_pose_scene(16, 0.0, 9700 + k, noise_px=[0.0, 1.0][k % 2], focal=float(rng.uniform(500, 2500)))
        fcam, cx, cy = d["camera"]["params"]
        x = np.asarray(d["p2d"]) - [cx, cy]
        for j in range(32):
            idx = rng.choice(16, 4, replace=False)
            big35.append(np.r_[x[idx].reshape(-1), np.asarray(d["p3d"])[idx].reshape(-1)])
    big35 = np.array(big35)
    assert len(big35) == 4608
    models, counts = gpu.solve_focal_batch("p35pf", big35)
    for k in range(len(big35)):
        po, fo = O.p35pf(big35[k, :8].reshape(4, 2), big35[k, 8:].reshape(4, 3))
        assert counts[k] == len(fo), k
        assert np.array_equal(models[k, : counts[k], :7], po, equal_nan=True) and np.array_equal(models[k, : counts[k], 7], fo, equal_nan=True), k
    # single-problem entry points on the golden vectors
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_focal_v1.json")))
    g35, g6 = minimal_inputs()
    for (x, X), want in zip(g35, G["p35pf"]):
        sols = gpu.p35pf(x, X)
        assert [[repr(float(v)) for v in np.r_[p.q, p.t]] for p, _ in sols] == want["poses"] and [repr(f) for _, f in sols] == want["focals"]
    for (a, b), want in zip(g6, G["six_point"]):
        sols = gpu.relpose_6pt_shared_focal(a, b)
        assert [[repr(float(v)) for v in np.r_[p.q, p.t]] for p, _ in sols] == want["poses"] and [repr(f) for _, f in sols] == want["focals"]
