"""GPU parity (-m gpu) of the absolute-pose bundle adjustment that refines camera intrinsics (BundleOptions
refine_focal_length / refine_principal_point / refine_extra_params): k_lm_cam through the C-ABI against the oracle,
which is bit-identical with the reference sources on this path (tests/test_oracle_vs_reference.py).

k_lm_cam sums every entry of the normal equations correspondence after correspondence, like the reference, for every n;
the robust cost is summed in the reference's order as well (at every n since round 4; by a tree beyond 256 before).  So: bit for bit at
every size (round 3: pose / camera to 1e-9 beyond 256), where a stop rule met at rounding level may
fire an LM iteration earlier or later.
"""
import numpy as np
import pytest

import oracle_lib as O
from poselib_amd import synth

pytestmark = pytest.mark.gpu

FLAGS = [{"refine_focal_length": True},
         {"refine_principal_point": True},
         {"refine_focal_length": True, "refine_principal_point": True},
         {"refine_focal_length": True, "refine_extra_params": True},
         {"refine_focal_length": True, "refine_principal_point": True, "refine_extra_params": True}]


def cameras(d):
    f, cx, cy = d["camera"]["params"]
    pix = np.asarray(d["p2d"])
    par = [f, f, cx, cy, -0.05, 0.01, 1e-3, -5e-4]
    return [(d["camera"], pix), (dict(d["camera"], model="PINHOLE", params=[f, f, cx, cy]), pix),
            (dict(d["camera"], model="OPENCV", params=par), synth.opencv_distort_pixels(pix, par))]


def off_calibration(cam, rs, rel, pp):
    par = np.array(cam["params"], dtype=np.float64)
    nf = 1 if cam["model"] == "SIMPLE_PINHOLE" else 2
    par[:nf] *= 1.0 + rel * rs.randn(nf)
    par[nf:nf + 2] += pp * rs.randn(2)
    return dict(cam, params=[float(v) for v in par])


def start_pose(d, rs, s):
    q = d["q_gt"] + s * rs.randn(4)
    return np.r_[q / np.linalg.norm(q), d["t_gt"] + s * rs.randn(3)]


@pytest.mark.parametrize("n", [7, 64, 200, 256])
def test_bundle_adjust_with_intrinsics_bit_exact_up_to_256_correspondences(gpu, n):
    rs = np.random.RandomState(100 + n)
    d = synth.absolute_pose_scene(n, 0.0, 5000 + n)
    p0 = start_pose(d, rs, 0.003)
    runs = 0
    for cam, pix in cameras(d):
        cam0 = off_calibration(cam, rs, 0.02, 3.0)
        pr = gpu.Problem(gpu.KIND_ABS, pix, d["p3d"])
        for flags in FLAGS:
            for bo in ({"loss_type": "CAUCHY", "loss_scale": 1.0}, {"loss_type": "HUBER", "loss_scale": 2.0, "max_iterations": 30}):
                bo = dict(bo, **flags)
                rp, rc, st = O.bundle_adjust_camera(pix, d["p3d"], cam0, p0, bo)
                pose, camera, it = pr.bundle_adjust(gpu.CameraPose(p0[:4], p0[4:]), cam0, bo)
                assert it == st.iterations, (cam["model"], flags, it, st.iterations)
                assert np.array_equal(np.r_[pose.q, pose.t], rp), (cam["model"], flags, np.abs(np.r_[pose.q, pose.t] - rp).max())
                assert np.array_equal(np.asarray(camera.params), rc), (cam["model"], flags)
                runs += 1
        pr.close()
    assert runs == 30


@pytest.mark.parametrize("n", [257, 1500, 6000])
def test_bundle_adjust_with_intrinsics_larger_problems_and_masks(gpu, n):
    rs = np.random.RandomState(200 + n)
    d = synth.absolute_pose_scene(n, 0.3, 5100 + n)
    p0 = start_pose(d, rs, 0.002)
    m = d["inlier_gt"]
    for cam, pix in cameras(d):
        cam0 = off_calibration(cam, rs, 0.02, 3.0)
        pr = gpu.Problem(gpu.KIND_ABS, pix, d["p3d"])
        for flags in FLAGS[2:]:
            for bo, mask in (({"loss_type": "CAUCHY", "loss_scale": 1.0}, m), ({"loss_type": "TRUNCATED", "loss_scale": 8.0, "max_iterations": 25}, None)):
                bo = dict(bo, **flags)
                sel = slice(None) if mask is None else mask
                rp, rc, st = O.bundle_adjust_camera(pix[sel], d["p3d"][sel], cam0, p0, bo)
                pose, camera, it = pr.bundle_adjust(gpu.CameraPose(p0[:4], p0[4:]), cam0, bo, mask=mask)
                # (round 4: the robust cost is summed in the reference's order at every n like the normal equations - bit for bit;
                # up to round 3 it was a tree sum beyond 256 correspondences and this test allowed 1e-9 and +- 6 iterations)
                assert it == st.iterations, (cam["model"], flags, it, st.iterations)
                assert np.array_equal(np.r_[pose.q, pose.t], rp), (cam["model"], flags, np.abs(np.r_[pose.q, pose.t] - rp).max())
                assert np.array_equal(np.asarray(camera.params), rc), (cam["model"], flags)
        pr.close()


def test_flags_that_select_no_parameter_are_the_plain_bundle(gpu):
    """refine_extra_params on a model without extra parameters: K = 6, the pose-only kernel, camera unchanged"""
    d = synth.absolute_pose_scene(300, 0.0, 5300)
    rs = np.random.RandomState(3)
    p0 = start_pose(d, rs, 0.003)
    pr = gpu.Problem(gpu.KIND_ABS, d["p2d"], d["p3d"])
    bo = {"refine_extra_params": True}
    rp, rc, st = O.bundle_adjust_camera(d["p2d"], d["p3d"], d["camera"], p0, bo)
    pose, camera, it = pr.bundle_adjust(gpu.CameraPose(p0[:4], p0[4:]), d["camera"], bo)
    plain, it2 = pr.refine(gpu.CameraPose(p0[:4], p0[4:]), {}, camera=d["camera"])
    assert it == st.iterations == it2 and np.array_equal(camera.params, d["camera"]["params"])
    assert np.abs(np.r_[pose.q, pose.t] - rp).max() < 1e-12 and np.array_equal(np.r_[pose.q, pose.t], np.r_[plain.q, plain.t])
    with pytest.raises(gpu.PoseLibAmdError):  # pl_refine_model cannot return a camera
        pr.refine(gpu.CameraPose(p0[:4], p0[4:]), {"refine_focal_length": True}, camera=d["camera"])
    pr.close()


def _front_end_cases():
    rs = np.random.RandomState(41)
    out = []
    for k, (n, outl) in enumerate([(1200, 0.4), (200, 0.2), (3000, 0.5)]):
        d = synth.absolute_pose_scene(n, outl, 5400 + k)
        for cam, pix in cameras(d):
            cam0 = off_calibration(cam, rs, 0.01, 2.0)
            opt = {"max_error": 8.0, "ransac": {"seed": 3 + k, "max_iterations": 2000}, "bundle": dict(FLAGS[2 + (k + len(out)) % 3])}
            out.append((pix, d["p3d"], cam0, opt, d["camera"]["params"][0]))
    return out


def test_estimate_absolute_pose_refining_intrinsics_matches_the_oracle(gpu):
    """robust.cc:36-126 with opt.bundle.refine_*: RANSAC + LO at the given calibration (identical decisions), the final
    bundle on the inliers moves pose and camera"""
    for pix, X, cam0, opt, f_true in _front_end_cases():
        rp, rmask, rst, rcam = O.estimate_absolute_pose(pix, X, cam0, opt, return_camera=True)
        img, info = gpu.estimate_absolute_pose(pix, X, cam0, opt)
        for k in ("iterations", "refinements", "num_inliers"):
            assert info[k] == rst[k], (k, info[k], rst[k])
        assert np.array_equal(np.asarray(info["inliers"], dtype=bool), rmask)
        assert np.abs(np.r_[img.pose.q, img.pose.t] - rp).max() < 1e-8
        assert np.abs(np.asarray(img.camera.params) - rcam).max() < 1e-8 * max(1.0, np.abs(rcam).max())
        assert not np.array_equal(np.asarray(img.camera.params), np.asarray(cam0["params"]))  # the camera did move
        assert abs(img.camera.params[0] - f_true) < abs(cam0["params"][0] - f_true)


def test_batched_front_end_with_and_without_intrinsics_equals_the_single_calls(gpu):
    cases = _front_end_cases()
    probs = []
    for i, (pix, X, cam0, opt, _) in enumerate(cases):
        probs.append(("abs", pix, X, cam0, opt))
        probs.append(("abs", pix, X, cam0, dict(opt, bundle={})))  # the same problem, pose only: same group, other kernel
    singles = [gpu.estimate_absolute_pose(pr[1], pr[2], pr[3], pr[4]) for pr in probs]
    res = gpu.estimate_batch(probs, max_in_flight=2)
    for (img, info), (simg, sinfo), pr in zip(res, singles, probs):
        assert np.array_equal(np.r_[img.pose.q, img.pose.t], np.r_[simg.pose.q, simg.pose.t])
        assert np.array_equal(img.camera.params, simg.camera.params)
        assert info["iterations"] == sinfo["iterations"] and info["inliers"] == sinfo["inliers"]
        if not pr[4]["bundle"]:
            # (camera.rescale(1/f) ... rescale(f) round trip of the reference: equal up to that rounding)
            assert np.abs(np.asarray(img.camera.params) - np.asarray(pr[3]["params"])).max() < 1e-9


def test_pose_refiners_bit_exact_on_rough_starts_and_all_lm_options(gpu):
    """k_lm (pose only) in the regime that exposed the sincos difference: rough starting poses, few
    correspondences (<= 256: sums in the reference's order), Marquardt damping, both lambda updates - bit for bit against the oracle"""
    rs = np.random.RandomState(78)
    for k in range(30):
        n = int(rs.choice([8, 15, 60, 250]))
        lu, dm = int(rs.randint(2)), int(rs.randint(2))
        loss = ["TRIVIAL", "HUBER", "CAUCHY"][int(rs.randint(3))]
        d = synth.absolute_pose_scene(n, 0.0, 6200 + k)
        un = O.unproject(d["camera"], d["p2d"])
        q = d["q_gt"] + 0.15 * rs.randn(4)
        p0 = np.r_[q / np.linalg.norm(q), d["t_gt"] + 0.3 * rs.randn(3)]
        bo = dict(loss_type=loss, loss_scale=0.01, max_iterations=60, lambda_update=lu, damping=dm)
        ref, st = O.bundle_adjust(un, d["p3d"], {"model": "NULL", "params": []}, p0, bo)
        pr = gpu.Problem(gpu.KIND_ABS, un, d["p3d"])
        pose, it = pr.refine(gpu.CameraPose(p0[:4], p0[4:]), bo)
        pr.close()
        assert it == st.iterations and np.array_equal(np.r_[pose.q, pose.t], ref, equal_nan=True), ("abs", k, n, lu, dm, loss)
        dr = synth.relative_pose_scene(n, 0.0, 6300 + k)
        a, b = O.unproject(dr["camera1"], dr["x1"]), O.unproject(dr["camera2"], dr["x2"])
        q = dr["q_gt"] + 0.1 * rs.randn(4)
        p0 = np.r_[q / np.linalg.norm(q), dr["t_gt"] / np.linalg.norm(dr["t_gt"]) + 0.2 * rs.randn(3)]
        bo = dict(loss_type=loss, loss_scale=1e-3, max_iterations=60, lambda_update=lu, damping=dm)
        ref, st = O.refine("relpose", a, b, p0, bo)
        pr = gpu.Problem(gpu.KIND_REL, a, b)
        pose, it = pr.refine(gpu.CameraPose(p0[:4], p0[4:]), bo)
        pr.close()
        assert it == st.iterations and np.array_equal(np.r_[pose.q, pose.t], ref, equal_nan=True), ("rel", k, n, lu, dm, loss)
