"""GPU parity at the sizes that are benchmarked (-m gpu), and adversarial tests of the scoring pre-filters ON THE DEVICE.

* The four bench workloads at full size (BASELINE configs 1-3: 100 000 iterations, min = max) against the oracle:
  iterations, refinements, hypotheses, inlier count and mask identical, model within 1e-6 - the configuration bench.py
  times is the configuration that is checked (bench.py repeats the check on RANSAC seeds 0..7 inside its own run).
* pl_debug_score_stream pushes arbitrary models through the streaming scorers of the main loop - the fp16 / matrix-core
  filter (k_shadow16 + k_score_mfma) and the fp32 filters (k_score_queue) - so the "may only reject a proven outlier"
  property is checked on the device with inputs chosen to break it: correspondences planted at thr (1 +- 1e-9 .. 1e-3),
  coordinates and translations at and beyond the fp16 range, thresholds at 1.0 and just above (where the matrix-core
  path hands over to the fp32 one), NaN / inf models, subnormal scales.  Ground truth: the oracle's exact fp64 counts
  (PoseLib/robust/utils.cc:36-65, 158-239, 300-329).
"""
import threading

import numpy as np
import pytest

import oracle_lib as O
from poselib_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-6
ITER = 100000
FOCAL = 1000.0
# (name, kind, N, outlier ratio, max_error px, data seed)  == bench.py WORKLOADS
FULL = [("p3p_5000", 0, 5000, 0.7, 12.0, 1001), ("relpose_5000", 1, 5000, 0.5, 1.0, 1002),
        ("fund_10000", 2, 10000, 0.5, 1.0, 1004), ("hom_10000", 3, 10000, 0.5, 1.0, 1003)]


def _workload_points(kind, n, outl, dseed):
    if kind == 0:
        d = synth.absolute_pose_scene(n, outl, dseed)
        return (d["p2d"] - 500.0) / FOCAL, d["p3d"]
    gen = {1: synth.relative_pose_scene, 2: synth.fundamental_scene, 3: synth.homography_scene}[kind]
    d = gen(n, outl, dseed)
    return (d["x1"] - 500.0) / FOCAL, (d["x2"] - 500.0) / FOCAL


def _model_diff(kind, got, ref):
    if kind in (0, 1):
        g = np.r_[got.q, got.t]
        dr = np.linalg.norm(synth.quat_to_rotmat(g[:4]) - synth.quat_to_rotmat(ref[:4]))
        return max(dr, np.linalg.norm(g[4:] - ref[4:]))
    a, b = np.ravel(got) / np.linalg.norm(got), np.ravel(ref) / np.linalg.norm(ref)
    return np.linalg.norm(a - b)  # sign included


@pytest.mark.parametrize("name,kind,n,outl,err,dseed", FULL, ids=[f[0] for f in FULL])
@pytest.mark.parametrize("rseed", [0, 5])
def test_full_size_run_matches_the_oracle(gpu, name, kind, n, outl, err, dseed, rseed):
    """ransac_impl.h:157-201 at the benchmarked size: one batch of 100 000 iterations on the device, every decision of
    the sequential loop replayed - against the oracle's plain loop"""
    A, B = _workload_points(kind, n, outl, dseed)
    opt = {"max_error": err / FOCAL, "ransac": {"max_iterations": ITER, "min_iterations": ITER, "seed": rseed}}
    prob = gpu.Problem(kind, A, B)
    model, info = prob.run(opt)
    prob.close()
    ofn = {0: O.ransac_pnp, 1: O.ransac_relpose, 2: O.ransac_fundamental, 3: O.ransac_homography}[kind]
    ref, mask, st = ofn(A, B, opt)
    print(name, "seed", rseed, "hypotheses", info["hypotheses"], st["hypotheses"], "refinements", info["refinements"],
          st["refinements"], "inliers", info["num_inliers"], st["num_inliers"], "nan", info["nan_hypotheses"],
          "gpu s", info["seconds"], "cpu s", st["seconds"])
    assert info["iterations"] == st["iterations"] == ITER
    assert info["hypotheses"] == st["hypotheses"]
    assert info["refinements"] == st["refinements"]
    assert info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    assert abs(info["model_score"] - st["model_score"]) <= 1e-9 * abs(st["model_score"])
    d = _model_diff(kind, model, ref)
    assert d <= POSE_TOL, d


# ------------------------------------------------------------------------------------------ adversarial: absolute pose
def _rot(q):
    return synth.quat_to_rotmat(np.asarray(q, dtype=np.float64))


def _plant_at_threshold(q, t, X, thr, rs):
    """2-D points whose reprojection error under (q, t) is thr (1 +- 1e-9 .. 1e-3)"""
    Z = X @ _rot(q).T + t
    ang = rs.uniform(0, 2 * np.pi, len(X))
    eps = 10.0 ** rs.uniform(-9, -3, len(X)) * rs.choice([-1.0, 1.0], len(X))
    rad = thr * (1 + eps)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = Z[:, :2] / Z[:, 2:3] + np.c_[rad * np.cos(ang), rad * np.sin(ang)]
    bad = ~np.isfinite(x).all(1)
    x[bad] = 0.0
    return x


def _abs_models(d, scale, off, rs, n_random=6):
    q, t = np.asarray(d["q_gt"], float), np.asarray(d["t_gt"], float) * scale
    R = _rot(q)
    t = t - R @ off  # the scene is expressed in a shifted world frame
    models = [np.r_[q, t]]
    for k in range(1, 6):  # increasingly perturbed
        qq = q + 10.0 ** (-k) * rs.randn(4)
        qq /= np.linalg.norm(qq)
        models.append(np.r_[qq, t + 10.0 ** (-k) * scale * rs.randn(3)])
    for _ in range(n_random):
        qq = rs.randn(4)
        qq /= np.linalg.norm(qq)
        models.append(np.r_[qq, rs.randn(3) * scale * 3 - _rot(qq) @ off])
    return models


def test_matrix_core_filter_never_drops_an_inlier_on_the_device(gpu):
    rs = np.random.RandomState(77)
    pairs = dropped = 0
    paths = {0: 0, 1: 0, 2: 0}
    #           scene scale, world shift, thresholds
    scenes = [(1.0, 0.0), (1.0, 0.0), (50.0, 0.0), (1e-3, 0.0), (1e-6, 0.0), (1e4, 0.0), (1.0, 1e3), (1.0, 2.9e4),
              (1.0, 3.1e4), (1.0, 1e5), (300.0, 0.0), (0.25, 0.0)]
    for si, (scale, shift) in enumerate(scenes):
        d = synth.absolute_pose_scene(4000, 0.5, 5000 + si)
        x0 = (np.asarray(d["p2d"]) - 500.0) / FOCAL
        off = np.array([shift, -shift, 0.5 * shift])
        X = np.asarray(d["p3d"]) * scale + off
        models = _abs_models(d, scale, off, rs)
        # models with |t| at / beyond the fp16 range, non-finite entries
        q = np.asarray(d["q_gt"], float)
        for tt in ([2.9e4, 0, 1.0], [0, -3.1e4, 2.0], [1e6, 1e6, 1e6], [np.nan, 0, 1], [0, np.inf, 1], [0, 0, 0]):
            models.append(np.r_[q, np.array(tt, float)])
        models.append(np.r_[np.nan, q[1:], 0.1, 0.2, 0.3])
        models.append(np.r_[2.0 * q, models[0][4:]])  # non-unit quaternion: |R_ij| up to 4 (rows not unit-bounded)
        M = np.array(models)
        for thr in (1e-5, 1e-3, 0.012, 0.5, 1.0, 1.0000001, 3.0):
            x = x0.copy()
            # half of the correspondences sit exactly at the decision boundary of the first model
            planted = _plant_at_threshold(M[0, :4], M[0, 4:], X, thr, rs)
            sel = rs.rand(len(X)) < 0.5
            x[sel] = planted[sel]
            if si == 1:  # image points far outside the field of view as well
                x[::7] *= 40.0
            prob = gpu.Problem(gpu.KIND_ABS, x, X)
            cnt, sc, path = prob.score_stream(M, thr)
            prob.close()
            paths[path] += 1
            if thr <= 0.9999 and np.isfinite(x).all():
                assert path == 2, (scale, shift, thr, path)  # the matrix-core filter is on the path
            if thr > 1.0:
                assert path == 1, (thr, path)  # hands over to the fp32 filter
            for k in range(len(M)):
                osc, ocnt = O.score("reproj", M[k], x, X, thr * thr)
                pairs += len(X)
                if cnt[k] != ocnt:
                    dropped += abs(int(cnt[k]) - int(ocnt))
                    print("MISMATCH scene", si, "scale", scale, "shift", shift, "thr", thr, "model", k, cnt[k], ocnt)
                else:
                    assert abs(sc[k] - osc) <= 1e-9 * abs(osc) + 1e-300
    print(f"absolute pose: {pairs} (model, correspondence) pairs through the device filters, paths {paths}, "
          f"count differences {dropped}")
    assert pairs >= 4_000_000
    assert dropped == 0


def test_matrix_core_filter_wide_field_of_view_and_far_world_frames_on_the_device(gpu):
    """Round 6: image coordinates up to +-4 with inliers planted at the threshold in the corners, world frames whose origin is
    10 .. 2e4 away from the camera while the scene sits next to it: the row's largest error is then the product rn16(t_2) rn16(p),
    which the sixteenth k slot of the three-half-plane form pays for (pl_prefilter.h "Round 6").  Counts through the streaming
    scorer must equal the oracle's exact evaluation for the planted model and for models next to it."""
    rs = np.random.RandomState(177)
    pairs = dropped = 0
    for trial in range(20):
        n = 4000
        q = rs.randn(4)
        q /= np.linalg.norm(q)
        R = _rot(q)
        fov = [0.5, 1.5, 4.0, 4.0][trial % 4]
        xy = rs.uniform(-fov, fov, (n, 2))
        if trial % 2:
            xy[: n // 2] = np.sign(xy[: n // 2]) * fov * (1 - 1e-3 * rs.rand(n // 2, 2))  # corners
        depth = 10.0 ** rs.uniform(-1, 1.5, n)
        Zc = np.c_[xy * depth[:, None], depth]
        tlen = [0.0, 10.0, 300.0, 5e3, 2e4][trial % 5]
        t = rs.randn(3)
        t *= tlen / max(np.linalg.norm(t), 1e-9)
        X = (Zc - t) @ R  # R X + t = Zc
        models = [np.r_[q, t]]
        for k in (6, 5, 4, 3):
            qq = q + 10.0 ** (-k) * rs.randn(4)
            qq /= np.linalg.norm(qq)
            models.append(np.r_[qq, t * (1 + 10.0 ** (-k) * rs.randn())])
        M = np.array(models)
        for thr in (1e-3, 0.012, 0.3, 0.99):
            x = _plant_at_threshold(q, t, X, thr, rs)
            prob = gpu.Problem(gpu.KIND_ABS, x, X)
            cnt, sc, path = prob.score_stream(M, thr)
            prob.close()
            assert path == 2, (trial, thr, path)
            for k in range(len(M)):
                osc, ocnt = O.score("reproj", M[k], x, X, thr * thr)
                pairs += n
                if cnt[k] != ocnt:
                    dropped += abs(int(cnt[k]) - int(ocnt))
                    print("MISMATCH trial", trial, "fov", fov, "|t|", tlen, "thr", thr, "model", k, cnt[k], ocnt)
                else:
                    assert abs(sc[k] - osc) <= 1e-9 * abs(osc) + 1e-300
    print(f"absolute pose, wide field of view: {pairs} pairs through the device filter, count differences {dropped}")
    assert pairs >= 1_500_000
    assert dropped == 0


# ------------------------------------------------------------------------------------------ adversarial: two-view
def _essential(q, t):
    R = _rot(q)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return tx @ R


@pytest.mark.parametrize("kind", [1, 2, 3])
def test_fp32_filters_never_drop_an_inlier_on_the_device(gpu, kind):
    rs = np.random.RandomState(80 + kind)
    pairs = diff = 0
    okind = {1: "sampson_pose", 2: "sampson_F", 3: "homography"}[kind]
    for trial in range(8):
        pixel = kind != 1 and trial % 2 == 0  # F / H also on raw pixel coordinates (entries spanning 1e-6 .. 1)
        # (fewer than 1024 normalised correspondences, or pixel coordinates: the fp32 form; otherwise Sampson scores take
        # the matrix-core form, which has its own test below)
        npts = 1000 if (kind != 3 and not pixel and trial % 4 == 1) else 3000
        if kind == 3:
            d = synth.homography_scene(npts, 0.5, 6000 + trial)
        else:
            d = synth.relative_pose_scene(npts, 0.5, 6100 + trial)
        x1, x2 = np.asarray(d["x1"], float), np.asarray(d["x2"], float)
        if not pixel:
            x1, x2 = (x1 - 500.0) / FOCAL, (x2 - 500.0) / FOCAL
        if kind == 3:  # a homography close to the truth from a handful of inliers
            inl = np.flatnonzero(d["inlier_gt"])[:40]
            rows = []
            for (u0, v0), (u1, v1) in zip(x1[inl], x2[inl]):
                p = np.array([u0, v0, 1.0])
                rows.append(np.r_[p, 0, 0, 0, -u1 * p])
                rows.append(np.r_[0, 0, 0, p, -v1 * p])
            gt = np.linalg.svd(np.array(rows))[2][-1].reshape(3, 3)
        else:
            q, t = np.asarray(d["q_gt"], float), np.asarray(d["t_gt"], float)
            E = _essential(q, t)
            if pixel:
                Kinv = np.array([[1e-3, 0, -0.5], [0, 1e-3, -0.5], [0, 0, 1.0]])
                gt = Kinv.T @ E @ Kinv
            else:
                gt = E
        models = []
        if kind == 1:
            for k in range(12):
                if k == 0:
                    qq, tt = q, t
                elif k < 6:
                    qq = q + 10.0 ** (-k) * rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = (t + 10.0 ** (-k) * rs.randn(3)) * rs.choice([1.0, 1e-3, 1e3])
                elif k < 10:
                    qq = rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = rs.randn(3)
                elif k == 10:
                    qq, tt = q, np.array([np.nan, 0.0, 1.0])
                else:
                    qq, tt = q, np.zeros(3)
                models.append(np.r_[qq, tt])
            M = np.array(models)
        else:
            for k in range(12):
                if k == 0:
                    Mk = gt
                elif k < 6:
                    Mk = gt + 10.0 ** (-2 * k) * np.abs(gt).max() * rs.randn(3, 3)
                elif k < 9:
                    Mk = rs.randn(3, 3) * ((1e-3 if pixel else 1.0) ** rs.randint(0, 3, (3, 3)))
                elif k == 9:
                    Mk = gt.copy()
                    Mk[2] = [1e-3, -1e-3, 1e-9] if pixel else [1.0, -1.0, 1e-9]  # denominators around zero (H)
                elif k == 10:
                    Mk = gt.copy()
                    Mk[1, 1] = np.nan
                else:
                    Mk = np.zeros((3, 3))
                models.append(Mk * 10.0 ** rs.choice([-12, -3, 0, 0, 4, 15]) * rs.choice([-1.0, 1.0]))
            M = np.array(models)
        for thr in ((0.3, 1.0, 3.0, 30.0) if pixel else (1e-5, 1e-3, 3e-3, 0.1)):
            prob = gpu.Problem(kind, x1, x2)
            cnt, sc, path = prob.score_stream(M, thr)
            prob.close()
            # (round 3: the homography scorer has a matrix-core form as well - normalised coordinates, >= 1024 correspondences)
            assert path == (2 if (not pixel and npts >= 1024) else 1), (kind, pixel, npts, thr, path)
            for k in range(len(M)):
                osc, ocnt = O.score(okind, M[k], x1, x2, thr * thr)
                pairs += len(x1)
                if cnt[k] != ocnt:
                    diff += abs(int(cnt[k]) - int(ocnt))
                    print("MISMATCH kind", kind, "trial", trial, "thr", thr, "model", k, cnt[k], ocnt)
                elif np.isfinite(osc):
                    assert abs(sc[k] - osc) <= 1e-9 * abs(osc) + 1e-300
    print(f"kind {kind}: {pairs} pairs through the device filter, count differences {diff}")
    assert pairs >= 900_000 and diff == 0


@pytest.mark.parametrize("kind", [1, 2])
def test_matrix_core_sampson_filter_never_drops_an_inlier_on_the_device(gpu, kind):
    """k_score_mfma2: fp16 high / low operands, two quadratic forms per pair out of the matrix pipe (pl_prefilter.h).
    Correspondences planted around the decision boundary, coordinates up to the operand bound of 8, matrices rescaled
    over 27 decades, NaN / zero / rank-deficient models; counts must equal the oracle's exact evaluation."""
    rs = np.random.RandomState(90 + kind)
    pairs = diff = 0
    okind = {1: "sampson_pose", 2: "sampson_F"}[kind]
    for trial in range(10):
        d = synth.relative_pose_scene(4000, 0.5, 6300 + trial)
        sc = 1.0 if kind == 1 else [1.0, 1.0, 7.5, 3.0, 0.05][trial % 5]
        x1, x2 = (np.asarray(d["x1"], float) - 500.0) / FOCAL * sc, (np.asarray(d["x2"], float) - 500.0) / FOCAL * sc
        q, t = np.asarray(d["q_gt"], float), np.asarray(d["t_gt"], float)
        S = np.diag([1 / sc, 1 / sc, 1.0])
        gt = S @ _essential(q, t) @ S
        models = []
        for k in range(14):
            if kind == 1:
                if k == 0:
                    qq, tt = q, t
                elif k < 7:
                    qq = q + 10.0 ** (-k) * rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = (t + 10.0 ** (-k) * rs.randn(3)) * rs.choice([1.0, 1e-3, 1e3])
                elif k < 12:
                    qq = rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = rs.randn(3)
                elif k == 12:
                    qq, tt = q, np.array([np.nan, 0.0, 1.0])
                else:
                    qq, tt = q, np.zeros(3)
                models.append(np.r_[qq, tt])
            else:
                if k == 0:
                    Mk = gt
                elif k < 7:
                    Mk = gt + 10.0 ** (-2 * k) * np.abs(gt).max() * rs.randn(3, 3)
                elif k < 10:
                    Mk = rs.randn(3, 3) * (1e-3 ** rs.randint(0, 3, (3, 3)))
                elif k == 10:
                    Mk = np.outer(rs.randn(3), rs.randn(3))  # rank one
                elif k == 11:
                    Mk = gt.copy()
                    Mk[1, 1] = np.nan
                elif k == 12:
                    Mk = np.zeros((3, 3))
                else:
                    Mk = np.diag([0.0, 0.0, 1.0])  # Cx + Cy = 0 for every correspondence
                models.append(Mk * 10.0 ** rs.choice([-12, -3, 0, 0, 4, 15]) * rs.choice([-1.0, 1.0]))
        M = np.array(models)
        # half of the second-image points moved onto the decision boundary of the first model: along the normal of the
        # epipolar line, to k thr (1 +- 1e-9 .. 1e-3) for a few k around the Sampson / point-line ratio
        a = np.c_[x1, np.ones(len(x1))]
        l = a @ gt.T
        nrm = np.linalg.norm(l[:, :2], axis=1)
        dist = np.einsum("ij,ij->i", np.c_[x2, np.ones(len(x2))], l) / nrm
        foot = x2 - dist[:, None] * l[:, :2] / nrm[:, None]
        for thr in (1e-5 * sc, 1e-3 * sc, 3e-3 * sc, 0.1 * sc):
            eps = 10.0 ** rs.uniform(-9, -3, len(x2)) * rs.choice([-1, 1], len(x2))
            kk = rs.choice([0.9, 1.0, 1.2, 1.41, 1.42, 1.6], len(x2))
            planted = foot + (kk * thr * (1 + eps))[:, None] * l[:, :2] / nrm[:, None]
            b = x2.copy()
            sel = rs.rand(len(x2)) < 0.5
            b[sel] = planted[sel]
            if trial == 1:
                b[::9] *= 30.0  # correspondences beyond the operand bound: the problem falls back to the fp32 form
            if trial == 2 and kind == 2:
                b[::11] = np.clip(b[::11] * 1.05, -7.99, 7.99)
            prob = gpu.Problem(kind, x1, b)
            cnt, scv, path = prob.score_stream(M, thr)
            prob.close()
            in_range = max(np.abs(x1).max(), np.abs(b).max()) <= 8.0 and 1e-12 <= thr * thr <= 1e4
            assert path == (2 if in_range else 1), (kind, trial, thr, path)
            for k in range(len(M)):
                osc, ocnt = O.score(okind, M[k], x1, b, thr * thr)
                pairs += len(x1)
                if cnt[k] != ocnt:
                    diff += abs(int(cnt[k]) - int(ocnt))
                    print("MISMATCH kind", kind, "trial", trial, "thr", thr, "model", k, cnt[k], ocnt)
                elif np.isfinite(osc):
                    assert abs(scv[k] - osc) <= 1e-9 * abs(osc) + 1e-300
    print(f"kind {kind}: {pairs} pairs through the matrix-core Sampson filter, count differences {diff}")
    assert pairs >= 2_000_000 and diff == 0


def test_matrix_core_homography_filter_never_drops_an_inlier_on_the_device(gpu):
    """k_score_mfmah (round 3): four linear forms per pair out of the matrix pipe, verdict max(|V_0|, |V_1|) > |U| + S
    (pl_prefilter.h).  Correspondences planted around the decision boundary, coordinate scales up to the operand bound of 8,
    matrices rescaled over 27 decades and negated, vanishing lines THROUGH the correspondences (the sign of h_2 changes among
    them), denominators of exactly zero, NaN / zero / inf models; counts must equal the oracle's exact evaluation."""
    rs = np.random.RandomState(97)
    pairs = diff = 0
    for trial in range(10):
        d = synth.homography_scene(4000, 0.5, 6500 + trial)
        sc = [1.0, 1.0, 7.5, 3.0, 0.05][trial % 5]
        x1, x2 = (np.asarray(d["x1"], float) - 500.0) / FOCAL * sc, (np.asarray(d["x2"], float) - 500.0) / FOCAL * sc
        inl = np.flatnonzero(d["inlier_gt"])[:40]
        rows = []
        for (u0, v0), (u1, v1) in zip(x1[inl], x2[inl]):
            p = np.array([u0, v0, 1.0])
            rows.append(np.r_[p, 0, 0, 0, -u1 * p])
            rows.append(np.r_[0, 0, 0, p, -v1 * p])
        gt = np.linalg.svd(np.array(rows))[2][-1].reshape(3, 3)
        models = []
        for k in range(14):
            if k == 0:
                Mk = gt
            elif k < 6:
                Mk = gt + 10.0 ** (-2 * k) * np.abs(gt).max() * rs.randn(3, 3)
            elif k < 9:
                Mk = rs.randn(3, 3) * (1e-3 ** rs.randint(0, 3, (3, 3)))
            elif k == 9:  # the vanishing line runs through the correspondences
                Mk = gt.copy()
                Mk[2] = np.array([1.0, -1.0, 1e-9]) * np.abs(gt).max()
            elif k == 10:  # ... through correspondence `trial` exactly: a denominator of 0
                Mk = gt.copy()
                Mk[2] = [1.0, 0.0, -x1[trial, 0]]
            elif k == 11:
                Mk = gt.copy()
                Mk[1, 1] = np.nan
            elif k == 12:
                Mk = np.zeros((3, 3))
            else:
                Mk = np.diag([1.0, 1.0, np.inf])
            models.append(Mk * 10.0 ** rs.choice([-12, -3, 0, 0, 4, 15]) * rs.choice([-1.0, 1.0]))
        M = np.array(models)
        a = np.c_[x1, np.ones(len(x1))]
        h = a @ gt.T
        proj = h[:, :2] / h[:, 2:3]
        for thr in (1e-5 * sc, 1e-3 * sc, 3e-3 * sc, 0.1 * sc, 1.0, 8.0):
            eps = 10.0 ** rs.uniform(-9, -3, len(x2)) * rs.choice([-1, 1], len(x2))
            ang = rs.uniform(0, 2 * np.pi, len(x2))
            planted = proj + (thr * (1 + eps))[:, None] * np.c_[np.cos(ang), np.sin(ang)]
            b = x2.copy()
            sel = (rs.rand(len(x2)) < 0.5) & (np.abs(planted).max(1) < 7.99)
            b[sel] = planted[sel]
            if trial == 1:
                b[::9] *= 30.0  # correspondences beyond the operand bound: the problem falls back to the fp32 form
            prob = gpu.Problem(3, x1, b)
            cnt, scv, path = prob.score_stream(M, thr)
            prob.close()
            in_range = max(np.abs(x1).max(), np.abs(b).max()) <= 8.0
            assert path == (2 if in_range else 1), (trial, thr, path)
            for k in range(len(M)):
                osc, ocnt = O.score("homography", M[k], x1, b, thr * thr)
                pairs += len(x1)
                if cnt[k] != ocnt:
                    diff += abs(int(cnt[k]) - int(ocnt))
                    print("MISMATCH trial", trial, "thr", thr, "model", k, cnt[k], ocnt)
                elif np.isfinite(osc):
                    assert abs(scv[k] - osc) <= 1e-9 * abs(osc) + 1e-300
    print(f"homography: {pairs} pairs through the matrix-core filter, count differences {diff}")
    assert pairs >= 3_000_000 and diff == 0


# ------------------------------------------------------------------------------------------ concurrency of the batch entry
def test_two_host_threads_may_call_estimate_batch_at_once(gpu):
    """include/poselib_amd.h promises re-entrancy; the internal pool serves one batch at a time, a second caller waits
    (it used to overwrite the first batch's job state).  Both batches must equal the single calls."""
    sets = []
    for base in (0, 100):
        probs, singles = [], []
        for i in range(10):
            n = 400 + 97 * i
            opt = {"ransac": {"seed": base + i}}
            if i % 2 == 0:
                d = synth.absolute_pose_scene(n, 0.4, 7000 + base + i)
                probs.append(("abs", d["p2d"], d["p3d"], d["camera"], opt))
                img, info = gpu.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
                singles.append((np.r_[img.pose.q, img.pose.t], info))
            else:
                d = synth.homography_scene(n, 0.4, 7000 + base + i)
                probs.append(("hom", d["x1"], d["x2"], opt))
                H, info = gpu.estimate_homography(d["x1"], d["x2"], opt)
                singles.append((H.reshape(-1), info))
        sets.append((probs, singles))
    out, errs = [None, None], []

    def work(j):
        try:
            for _ in range(3):
                out[j] = gpu.estimate_batch(sets[j][0], max_in_flight=4)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(j,)) for j in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    for j in range(2):
        for (model, info), (ref_model, ref_info), pr in zip(out[j], sets[j][1], sets[j][0]):
            flat = np.r_[model.pose.q, model.pose.t] if pr[0] == "abs" else model.reshape(-1)
            assert np.array_equal(flat, ref_model)
            for k in ("iterations", "refinements", "num_inliers", "model_score"):
                assert info[k] == ref_info[k]
            assert info["inliers"] == ref_info["inliers"]
