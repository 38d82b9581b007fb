"""The scoring kernels' conservative fp32 pre-filter (poselib_amd/csrc/pl_prefilter.h) may only exclude
correspondences that are NOT inliers under the reference's exact fp64 test.  The kernels and this test run the very
same inline functions (tests/hostmath compiles the device headers for the host), so the property is checked here
on the CPU against exact inlier flags that tests/test_hostmath_vs_oracle.py pins bit-for-bit to the oracle:

    for every (model, correspondence):   proven_outlier  =>  not inlier

over good, perturbed and random models; normalised and pixel-scale coordinates; thresholds from 1e-9 to 1e+2;
correspondences planted exactly at the threshold; rescaled matrices (Sampson / homography are scale invariant);
models with NaN / inf / out-of-range entries.  It also reports how selective the filter is.
"""
import numpy as np
import pytest

import hostmath_lib as HM
from poselib_amd import synth


def _rot(rs, s=1.0):
    w = rs.randn(3) * s
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


def _check(est, rec, cols, thr2, xy_absmax=0.0, stats=None):
    _, _, inl, _ = HM.score(est, rec, cols, thr2)
    en, out = HM.prefilter(est, rec, cols, thr2, xy_absmax)
    bad = inl & out
    assert not bad.any(), (est, thr2, np.flatnonzero(bad)[:5])
    if stats is not None and en:
        stats[0] += int((~inl).sum())
        stats[1] += int((~inl & ~out).sum())
    return en


def test_absolute_pose_filter_never_drops_an_inlier():
    rs = np.random.RandomState(1)
    stats = [0, 0]
    for trial in range(60):
        scale = [1.0, 1.0, 50.0, 1e-3, 1e4][trial % 5]  # scene scale
        d = synth.absolute_pose_scene(1500, 0.5, 100 + trial)
        par = d["camera"]["params"]
        x = (np.asarray(d["p2d"]) - par[-2:]) / par[0]
        X = np.asarray(d["p3d"]) * scale
        q, t = np.asarray(d["q_gt"], float), np.asarray(d["t_gt"], float) * scale
        cols = [x[:, 0], x[:, 1], X[:, 0], X[:, 1], X[:, 2]]
        xy = float(np.abs(x).max())
        for thr in (1e-4, 2e-3, 1.2e-2, 0.3, 5.0):
            for model in range(4):
                if model == 0:
                    qq, tt = q, t
                elif model == 1:
                    qq = q + 1e-3 * rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = t + 1e-3 * scale * rs.randn(3)
                elif model == 2:
                    qq = rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = rs.randn(3) * scale * 3
                else:  # camera centre next to a scene point: depths around zero
                    qq = q
                    R = np.array(HM.pose_record(q, t)[HM.MAT:HM.MAT + 9]).reshape(3, 3)
                    tt = -R @ X[trial] + 1e-6 * scale * rs.randn(3)
                _check("abs", HM.pose_record(qq, tt), cols, thr * thr, xy, stats)
    print(f"abs: {stats[1]}/{stats[0]} non-inliers pass the filter ({100.0 * stats[1] / stats[0]:.3f} %)")
    assert stats[1] < 0.02 * stats[0]


def _plant_at_threshold_abs(rec, X, thr, rs):
    """2-D points whose reprojection error is thr * (1 +- 1e-7 ... 1e-3)"""
    R = rec[HM.MAT:HM.MAT + 9].reshape(3, 3)
    Z = X @ R.T + rec[4:7]
    ok = Z[:, 2] > 1e-3
    ang = rs.uniform(0, 2 * np.pi, len(X))
    eps = 10.0 ** rs.uniform(-9, -3, len(X)) * rs.choice([-1, 1], len(X))
    rad = thr * (1 + eps)
    x = Z[:, :2] / Z[:, 2:3] + np.c_[rad * np.cos(ang), rad * np.sin(ang)]
    return x[ok], X[ok]


def test_absolute_pose_filter_at_the_threshold():
    rs = np.random.RandomState(2)
    for trial in range(20):
        d = synth.absolute_pose_scene(2000, 0.0, 300 + trial)
        rec = HM.pose_record(d["q_gt"], d["t_gt"])
        for thr in (1e-5, 1e-3, 1.2e-2, 0.5):
            x, X = _plant_at_threshold_abs(rec, np.asarray(d["p3d"]), thr, rs)
            cols = [x[:, 0], x[:, 1], X[:, 0], X[:, 1], X[:, 2]]
            _check("abs", rec, cols, thr * thr, float(np.abs(x).max()))


def _two_view_cols(d, pixel):
    x1, x2 = np.asarray(d["x1"], float), np.asarray(d["x2"], float)
    if not pixel:
        x1, x2 = (x1 - 500.0) / 1000.0, (x2 - 500.0) / 1000.0
    return [x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1]]


def _essential(q, t):
    return HM.pose_record(q, t, essential=True)


@pytest.mark.parametrize("est", ["fund", "rel"])
def test_sampson_filter_never_drops_an_inlier(est):
    rs = np.random.RandomState(3)
    stats = [0, 0]
    for trial in range(40):
        d = synth.relative_pose_scene(1500, 0.5, 400 + trial)
        q, t = np.asarray(d["q_gt"], float), np.asarray(d["t_gt"], float)
        pixel = est == "fund" and trial % 2 == 0
        cols = _two_view_cols(d, pixel)
        recE = _essential(q, t)
        E = recE[HM.MAT:HM.MAT + 9].reshape(3, 3)
        if pixel:
            Kinv = np.array([[1e-3, 0, -0.5], [0, 1e-3, -0.5], [0, 0, 1.0]])
            Fgt = Kinv.T @ E @ Kinv
        else:
            Fgt = E
        for thr in ((0.3, 1.0, 3.0, 20.0) if pixel else (1e-5, 1e-3, 3e-3, 0.1)):
            for model in range(5):
                if est == "rel":
                    if model == 0:
                        rec = recE
                    elif model < 3:
                        qq = q + 10.0 ** -(model + 1) * rs.randn(4)
                        qq /= np.linalg.norm(qq)
                        tt = t + 10.0 ** -(model + 1) * rs.randn(3)
                        rec = _essential(qq, tt * rs.choice([1.0, 1e-3, 1e3]))
                    else:
                        qq = rs.randn(4)
                        rec = _essential(qq / np.linalg.norm(qq), rs.randn(3))
                else:
                    if model == 0:
                        F = Fgt
                    elif model < 3:
                        F = Fgt + 10.0 ** -(2 * model) * np.abs(Fgt).max() * rs.randn(3, 3)
                    else:
                        F = rs.randn(3, 3) * (1e-3 if pixel else 1.0) ** rs.randint(0, 3, (3, 3))
                    F = F * 10.0 ** rs.choice([-12, -3, 0, 0, 4, 15])  # Sampson is scale invariant
                    rec = HM.matrix_record(F)
                _check(est, rec, cols, thr * thr, 0.0, stats)
    print(f"{est}: {stats[1]}/{stats[0]} non-inliers pass the filter ({100.0 * stats[1] / stats[0]:.3f} %)")
    # rel: correspondences that satisfy the Sampson test but fail the cheirality test are non-inliers the Sampson
    # filter cannot (and must not) exclude
    # fund: half of the trials are raw pixel coordinates with matrices rescaled by 1e-12 .. 1e15, where the rounding slack
    # e_C carries weight; the one-comparison form of the filter (pl_prefilter.h) trades 3 points of selectivity there
    # for 4 fewer operations per pair everywhere
    assert stats[1] < (0.15 if est == "rel" else 0.08) * stats[0]


def test_homography_filter_never_drops_an_inlier():
    rs = np.random.RandomState(4)
    stats = [0, 0]
    for trial in range(40):
        pixel = trial % 2 == 0
        d = synth.homography_scene(1500, 0.5, 500 + trial)
        cols = _two_view_cols(d, pixel)
        # a homography close to the truth: fit to a handful of inlier correspondences
        inl = np.flatnonzero(d["inlier_gt"])[:40]
        a = np.c_[cols[0][inl], cols[1][inl], np.ones(len(inl))]
        b = np.c_[cols[2][inl], cols[3][inl]]
        A = []
        for p, (u, v) in zip(a, b):
            A.append(np.r_[p, 0, 0, 0, -u * p])
            A.append(np.r_[0, 0, 0, p, -v * p])
        Hgt = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
        for thr in ((0.3, 1.0, 3.0, 30.0) if pixel else (1e-5, 1e-3, 3e-3, 0.1)):
            for model in range(5):
                if model == 0:
                    Hm = Hgt
                elif model < 3:
                    Hm = Hgt + 10.0 ** -(2 * model) * np.abs(Hgt).max() * rs.randn(3, 3)
                elif model == 3:
                    Hm = rs.randn(3, 3)
                else:  # third row nearly orthogonal to many points: denominators around zero
                    Hm = Hgt.copy()
                    Hm[2] = [1.0, -1.0, 1e-9] if not pixel else [1e-3, -1e-3, 1e-9]
                Hm = Hm * 10.0 ** rs.choice([-12, -3, 0, 0, 4, 15]) * rs.choice([-1, 1])
                _check("hom", HM.matrix_record(Hm), cols, thr * thr, 0.0, stats)
    print(f"hom: {stats[1]}/{stats[0]} non-inliers pass the filter ({100.0 * stats[1] / stats[0]:.3f} %)")
    assert stats[1] < 0.10 * stats[0]


def test_degenerate_models():
    d = synth.homography_scene(500, 0.3, 7)
    cols = _two_view_cols(d, True)
    for est in ("fund", "hom"):
        for M in (np.zeros((3, 3)), np.full((3, 3), np.nan), np.eye(3) * 1e-30, np.eye(3) * 1e30,
                  np.array([[1, 0, 0], [0, np.inf, 0], [0, 0, 1.0]]), np.array([[1, 0, 0], [0, 1, 0], [np.nan, 0, 1.0]])):
            rec = HM.matrix_record(M)
            _, _, inl, _ = HM.score(est, rec, cols, 4.0)
            en, out = HM.prefilter(est, rec, cols, 4.0)
            assert not (inl & out).any()
            if np.isnan(M).any():
                assert not inl.any() and out.all()  # the NaN shortcut is exact
    a = synth.absolute_pose_scene(500, 0.3, 8)
    x = (np.asarray(a["p2d"]) - 500.0) / 1000.0
    X = np.asarray(a["p3d"])
    cols = [x[:, 0], x[:, 1], X[:, 0], X[:, 1], X[:, 2]]
    for t in ([np.nan, 0, 0], [0, 0, np.inf], [1e30, 0, 1e30], [0, 0, 0]):
        rec = HM.pose_record(a["q_gt"], np.array(t, float))
        _, _, inl, _ = HM.score("abs", rec, cols, 1e-4)
        en, out = HM.prefilter("abs", rec, cols, 1e-4, float(np.abs(x).max()))
        assert not (inl & out).any()
    # thresholds outside the range fp32 can carry switch the filter off
    assert not HM.prefilter("fund", HM.matrix_record(np.eye(3)), _two_view_cols(d, True), 1e-40)[0]


# ---- Sampson, fp16 / matrix-core form (k_score_mfma2) ---------------------------------------------------------------
def _check16(est, rec, cols, thr2, uv, stats=None, order=0):
    _, _, inl, _ = HM.score(est, rec, cols, thr2)
    en, out = HM.prefilter16(est, rec, cols, thr2, uv, order)
    bad = inl & out
    assert not bad.any(), (est, thr2, np.flatnonzero(bad)[:5])
    if stats is not None and en:
        stats[0] += int((~inl).sum())
        stats[1] += int((~inl & ~out).sum())
    return en


def test_fp16_conversions_match_ieee_half():
    rs = np.random.RandomState(11)
    v = np.concatenate([rs.randn(400000) * 10.0 ** rs.uniform(-9, 5, 400000),
                        [0, -0.0, 65504, 65519.9, 65520, 1e9, -1e9, np.inf, -np.inf, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8,
                         2.99e-8, 1e-10, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11]]).astype(np.float32)
    bits, back = HM.half_rn(v)
    with np.errstate(over="ignore"):
        want = v.astype(np.float16)
    assert (bits == want.view(np.uint16)).all()
    assert (back == want.astype(np.float32)).all()


@pytest.mark.parametrize("est", ["fund", "rel"])
def test_sampson_fp16_form_never_drops_an_inlier(est):
    rs = np.random.RandomState(5)
    stats = [0, 0]
    enabled = 0
    for trial in range(40):
        d = synth.relative_pose_scene(1500, 0.5, 600 + trial)
        q, t = np.asarray(d["q_gt"], float), np.asarray(d["t_gt"], float)
        cols = _two_view_cols(d, False)
        # coordinate scales the operands must carry: normalised image points, points near the bound of 8, tiny ones
        sc = [1.0, 1.0, 7.9, 1e-3, 3.0][trial % 5]
        cols = [c * sc for c in cols]
        uv = float(max(np.abs(c).max() for c in cols))
        recE = _essential(q, t)
        E = recE[HM.MAT:HM.MAT + 9].reshape(3, 3)
        for thr in (1e-5 * sc, 1e-3 * sc, 3e-3 * sc, 0.1 * sc):
            for model in range(5):
                if est == "rel":
                    if model == 0:
                        rec = recE
                    elif model < 3:
                        qq = q + 10.0 ** -(model + 1) * rs.randn(4)
                        qq /= np.linalg.norm(qq)
                        tt = t + 10.0 ** -(model + 1) * rs.randn(3)
                        rec = _essential(qq, tt * rs.choice([1.0, 1e-3, 1e3]))
                    else:
                        qq = rs.randn(4)
                        rec = _essential(qq / np.linalg.norm(qq), rs.randn(3))
                else:
                    S = np.diag([1 / sc, 1 / sc, 1.0])
                    Fgt = S @ E @ S
                    if model == 0:
                        F = Fgt
                    elif model < 3:
                        F = Fgt + 10.0 ** -(2 * model) * np.abs(Fgt).max() * rs.randn(3, 3)
                    else:
                        F = rs.randn(3, 3) * 1e-3 ** rs.randint(0, 3, (3, 3))
                    F = F * 10.0 ** rs.choice([-12, -3, 0, 0, 4, 15])
                    rec = HM.matrix_record(F)
                if est == "rel" and sc != 1.0:
                    continue  # (an essential matrix lives on normalised coordinates)
                # selectivity is reported for the regime the form is meant for (coordinates of order one, thresholds of
                # a pixel at focal lengths of 10^2..10^4); the never-drops property is checked everywhere
                typical = sc in (1.0, 3.0) and thr >= 1e-3 * sc
                enabled += _check16(est, rec, cols, thr * thr, uv, stats if typical else None, order=trial % 3)
    assert enabled > 100
    print(f"{est} (fp16 form): {stats[1]}/{stats[0]} non-inliers pass the filter ({100.0 * stats[1] / stats[0]:.3f} %)")
    assert stats[1] < (0.15 if est == "rel" else 0.05) * stats[0]


def test_sampson_fp16_form_at_the_threshold_and_degenerate():
    rs = np.random.RandomState(6)
    for trial in range(12):
        d = synth.relative_pose_scene(3000, 0.0, 700 + trial)
        cols = _two_view_cols(d, False)
        rec = _essential(d["q_gt"], d["t_gt"])
        E = rec[HM.MAT:HM.MAT + 9].reshape(3, 3)
        # move the second point along the epipolar normal until the Sampson error is thr (1 +- eps)
        a = np.c_[cols[0], cols[1], np.ones(len(cols[0]))]
        l = a @ E.T  # epipolar lines in image 2
        nrm = np.linalg.norm(l[:, :2], axis=1)
        b = np.c_[cols[2], cols[3]]
        dist = (np.einsum("ij,ij->i", np.c_[b, np.ones(len(b))], l)) / nrm
        foot = b - (dist / nrm)[:, None] * l[:, :2]
        for thr in (1e-4, 1e-3, 1e-2):
            eps = 10.0 ** rs.uniform(-9, -3, len(b)) * rs.choice([-1, 1], len(b))
            # (first order: the Sampson distance is close to the point-line distance / sqrt(2) here; scan a band around it)
            for k in (0.9, 1.0, 1.2, 1.41, 1.6):
                bb = foot + (k * thr * (1 + eps) / nrm)[:, None] * l[:, :2]
                c2 = [cols[0], cols[1], bb[:, 0], bb[:, 1]]
                uv = float(max(np.abs(c).max() for c in c2))
                _check16("fund", rec, c2, thr * thr, uv)
                _check16("rel", rec, c2, thr * thr, uv)
    d = synth.homography_scene(500, 0.3, 7)
    cols = _two_view_cols(d, False)
    uv = float(max(np.abs(c).max() for c in cols))
    for M in (np.zeros((3, 3)), np.full((3, 3), np.nan), np.eye(3) * 1e-30, np.eye(3) * 1e30, np.eye(3) * 1e-17,
              np.array([[1, 0, 0], [0, np.inf, 0], [0, 0, 1.0]]), np.array([[1, 0, 0], [0, 1, 0], [np.nan, 0, 1.0]]),
              np.array([[0, 0, 0], [0, 0, 0], [0, 0, 1.0]]), np.array([[1e-9, 0, 0], [0, 1e-9, 0], [0, 0, 1.0]])):
        rec = HM.matrix_record(M)
        _, _, inl, _ = HM.score("fund", rec, cols, 1e-4)
        en, out = HM.prefilter16("fund", rec, cols, 1e-4, uv)
        assert en and not (inl & out).any()
        if np.isnan(M).any():
            assert out.all()
    # points with NaN / huge coordinates are never excluded; the form is off beyond the coordinate bound
    bad = [c.copy() for c in cols]
    bad[0][:5] = np.nan
    bad[2][5:10] = 7.99
    en, out = HM.prefilter16("fund", HM.matrix_record(np.eye(3)), bad, 1e-4, 7.99)
    assert en and not out[:5].any()
    assert not HM.prefilter16("fund", HM.matrix_record(np.eye(3)), cols, 1e-4, 8.5)[0]
    assert not HM.prefilter16("fund", HM.matrix_record(np.eye(3)), cols, 1e-14, uv)[0]


# ---- reprojection, fp16 / matrix-core form (k_shadow16 + k_score_mfma) ------------------------------------------------
def _check16_abs(rec, cols, thr2, xy, stats=None, order=0):
    _, _, inl, _ = HM.score("abs", rec, cols, thr2)
    en, out = HM.prefilter16_abs(rec, cols, thr2, xy, order)
    bad = inl & out
    assert not bad.any(), (thr2, np.flatnonzero(bad)[:5])
    if stats is not None and en:
        stats[0] += int((~inl).sum())
        stats[1] += int((~inl & ~out).sum())
    return en


def test_absolute_pose_fp16_form_never_drops_an_inlier():
    """The four half-plane forms of the headline kernel's filter, built by the very operand functions the kernels call
    (pf16_abs_model / pf16_abs_point) and accumulated in fp32 in several orders: scenes scaled by 1e-6 .. 300, world frames
    shifted up to and beyond the fp16 range, camera centres next to scene points, thresholds 1e-5 .. 1, correspondences
    planted at the decision boundary, NaN / inf / huge translations, non-unit quaternions."""
    rs = np.random.RandomState(21)
    stats = [0, 0]
    enabled = 0
    for trial in range(48):
        scale, shift = [(1.0, 0.0), (1.0, 0.0), (50.0, 0.0), (1e-3, 0.0), (1e-6, 0.0), (300.0, 0.0), (1.0, 1e3), (1.0, 2.9e4),
                        (1.0, 3.1e4), (0.25, 0.0), (1.0, 0.0), (1.0, 1e5)][trial % 12]
        d = synth.absolute_pose_scene(1500, 0.5, 800 + trial)
        par = d["camera"]["params"]
        x = (np.asarray(d["p2d"]) - par[-2:]) / par[0]
        off = np.array([shift, -shift, 0.5 * shift])
        X = np.asarray(d["p3d"]) * scale + off
        q = np.asarray(d["q_gt"], float)
        R = np.array(HM.pose_record(q, np.zeros(3))[HM.MAT:HM.MAT + 9]).reshape(3, 3)
        t = np.asarray(d["t_gt"], float) * scale - R @ off
        if trial % 12 == 10:
            x = x.copy()
            x[::7] *= 40.0  # image points far outside the field of view
        for thr in (1e-5, 1e-3, 1.2e-2, 0.3, 1.0):
            rec0 = HM.pose_record(q, t)
            xp, Xp = _plant_at_threshold_abs(rec0, X, thr, rs)
            if len(xp) < 100:  # (tiny scenes: depths below the helper's cut) the scene as it is
                xx, Xp = x, X
            else:
                sel = rs.rand(len(xp)) < 0.5
                xx = np.where(sel[:, None], xp, x[:len(xp)])
            cols = [xx[:, 0], xx[:, 1], Xp[:, 0], Xp[:, 1], Xp[:, 2]]
            xy = float(np.abs(xx).max())
            for model in range(6):
                if model == 0:
                    qq, tt = q, t
                elif model == 1:
                    qq = q + 1e-3 * rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = t + 1e-3 * scale * rs.randn(3)
                elif model == 2:
                    qq = rs.randn(4)
                    qq /= np.linalg.norm(qq)
                    tt = rs.randn(3) * scale * 3 - np.array(HM.pose_record(qq, np.zeros(3))[HM.MAT:HM.MAT + 9]).reshape(3, 3) @ off
                elif model == 3:  # camera centre next to a scene point: depths around zero
                    qq, tt = q, -R @ X[trial] + 1e-6 * scale * rs.randn(3)
                elif model == 4:
                    qq, tt = q, np.array([[2.9e4, 0, 1.0], [0, -3.1e4, 2.0], [np.nan, 0, 1], [0, np.inf, 1]][trial % 4])
                else:
                    qq, tt = 2.0 * q, t  # rows not unit-bounded
                # (selectivity: random models at the thresholds the form is meant for; planted boundary points and tiny
                # thresholds - below the fp16 slack - pass the filter by design)
                typical = model == 2 and scale == 1.0 and shift == 0.0 and thr >= 1.2e-2
                en = _check16_abs(HM.pose_record(qq, tt), cols, thr * thr, xy, stats if typical else None, order=trial % 3)
                enabled += en
    assert enabled > 500
    print(f"abs (fp16 form): {stats[1]}/{stats[0]} non-inliers pass the filter ({100.0 * stats[1] / stats[0]:.3f} %)")
    assert stats[1] < 0.05 * stats[0]
    # above thr = 1 the form hands over to the fp32 one
    assert not HM.prefilter16_abs(HM.pose_record(q, t), cols, 1.0000001 ** 2, 0.5)[0]


def test_absolute_pose_fp16_form_wide_field_of_view_and_large_translations():
    """Round 6: image coordinates up to +-4 (a field of view far beyond 90 degrees) with inliers planted AT the threshold in the
    corners, and world frames whose origin is far from the camera (|t| = 10 .. 2e4 with the scene next to the camera): the product
    rn16(t_2) rn16(p) of the fp16 form then carries the largest error of the row (2^-10 |t_2| |p|), which the sixteenth k slot
    (Tm |p|) pays for - rounds 2 - 5 charged it to a per-hypothesis slack that only covers it while 2 |p| <= 1 + max|x|,|y| + thr."""
    rs = np.random.RandomState(77)
    enabled = 0
    for trial in range(40):
        n = 1500
        q = rs.randn(4)
        q /= np.linalg.norm(q)
        R = np.array(HM.pose_record(q, np.zeros(3))[HM.MAT:HM.MAT + 9]).reshape(3, 3)
        fov = [0.5, 1.5, 4.0, 4.0][trial % 4]
        xy = rs.uniform(-fov, fov, (n, 2))
        if trial % 2:
            xy[: n // 2] = np.sign(xy[: n // 2]) * fov * (1 - 1e-3 * rs.rand(n // 2, 2))  # corners
        depth = 10.0 ** rs.uniform(-1, 1.5, n)
        Zc = np.c_[xy * depth[:, None], depth]  # camera frame
        tlen = [0.0, 10.0, 300.0, 5e3, 2e4][trial % 5]
        t = rs.randn(3)
        t *= tlen / max(np.linalg.norm(t), 1e-9)
        X = (Zc - t) @ R  # world points: R X + t = Zc
        keep = (np.abs(X).sum(1) < 2.5e4) & (np.abs(xy).sum(1) * np.abs(X).sum(1) < 2.9e4)
        X, xy = X[keep], xy[keep]
        if len(X) < 50:
            continue
        for thr in (1e-3, 1.2e-2, 0.3, 1.0):
            rec = HM.pose_record(q, t)
            xp, Xp = _plant_at_threshold_abs(rec, X, thr, rs)
            cols = [xp[:, 0], xp[:, 1], Xp[:, 0], Xp[:, 1], Xp[:, 2]]
            for order in range(3):
                enabled += _check16_abs(rec, cols, thr * thr, float(np.abs(xp).max()), order=order)
            # a second model next to the first: its inliers are a subset of the planted points
            q2 = q + 1e-4 * rs.randn(4)
            q2 /= np.linalg.norm(q2)
            enabled += _check16_abs(HM.pose_record(q2, t * (1 + 1e-5)), cols, thr * thr, float(np.abs(xp).max()), order=trial % 3)
    assert enabled > 300


def _check16_hom(rec, cols, thr2, uv, stats=None, order=0):
    _, _, inl, _ = HM.score("hom", rec, cols, thr2)
    en, out = HM.prefilter16_hom(rec, cols, thr2, uv, order)
    bad = inl & out
    assert not bad.any(), (thr2, np.flatnonzero(bad)[:5])
    if stats is not None and en:
        stats[0] += int((~inl).sum())
        stats[1] += int((~inl & ~out).sum())
    return en


def _fit_homography(cols, inl):
    a = np.c_[cols[0][inl], cols[1][inl], np.ones(len(inl))]
    b = np.c_[cols[2][inl], cols[3][inl]]
    A = []
    for p, (u, v) in zip(a, b):
        A.append(np.r_[p, 0, 0, 0, -u * p])
        A.append(np.r_[0, 0, 0, p, -v * p])
    return np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)


def test_homography_fp16_form_never_drops_an_inlier():
    """The four linear forms of k_score_mfmah's filter (round 3), built by the very operand functions the kernels call
    (pf16_hom_model / pf16_hom_point) and accumulated in fp32 in several orders: coordinate scales 1e-3 .. 7.9, thresholds
    1e-5 .. 8, matrices rescaled by 1e-12 .. 1e15 and negated, third rows whose form h_2 changes sign inside the image
    (|h_2| has no fixed sign: the inlier region is two opposite cones), denominators around zero, correspondences planted at
    the decision boundary thr (1 +- 1e-9 .. 1e-3), NaN / inf / zero matrices, NaN points, coordinates beyond the bound."""
    rs = np.random.RandomState(31)
    stats = [0, 0]
    enabled = 0
    for trial in range(40):
        d = synth.homography_scene(1500, 0.5, 900 + trial)
        cols = _two_view_cols(d, False)
        sc = [1.0, 1.0, 7.9, 1e-3, 3.0][trial % 5]
        cols = [np.asarray(c, float) * sc for c in cols]
        uv = float(max(np.abs(c).max() for c in cols))
        Hgt = _fit_homography(cols, np.flatnonzero(d["inlier_gt"])[:40])
        for thr in (1e-5 * sc, 1e-3 * sc, 3e-3 * sc, 0.1 * sc, 1.0, 8.0):
            # half of the correspondences planted at the decision boundary of the first model: b = H a + thr (1 +- eps) dir
            a = np.c_[cols[0], cols[1], np.ones(len(cols[0]))]
            h = a @ Hgt.T
            proj = h[:, :2] / h[:, 2:3]
            ang = rs.uniform(0, 2 * np.pi, len(a))
            eps = rs.choice([1e-9, 1e-6, 1e-3], len(a)) * rs.choice([-1, 1], len(a))
            planted = proj + (thr * (1 + eps))[:, None] * np.c_[np.cos(ang), np.sin(ang)]
            sel = (rs.rand(len(a)) < 0.5) & (np.abs(planted).max(1) < 7.99)
            cc = [cols[0], cols[1], np.where(sel, planted[:, 0], cols[2]), np.where(sel, planted[:, 1], cols[3])]
            uvc = float(max(np.abs(c).max() for c in cc))
            for model in range(7):
                if model == 0:
                    Hm = Hgt
                elif model < 3:
                    Hm = Hgt + 10.0 ** -(2 * model) * np.abs(Hgt).max() * rs.randn(3, 3)
                elif model == 3:
                    Hm = rs.randn(3, 3)
                elif model == 4:  # the vanishing line runs through the correspondences: h_2 changes sign among them
                    Hm = Hgt.copy()
                    Hm[2] = np.array([1.0, -1.0, 1e-9]) * np.abs(Hgt).max()
                elif model == 5:  # ... and through a planted correspondence exactly: denominators of 0, 1e-300, 1e-17
                    Hm = Hgt.copy()
                    Hm[2] = [1.0, 0.0, -cols[0][trial] + [0.0, 1e-300, 1e-17][trial % 3]]
                else:
                    Hm = np.array([[np.nan, 0, 0], [0, 1, 0], [0, 0, 1.0]]) if trial % 3 == 0 else (
                        np.zeros((3, 3)) if trial % 3 == 1 else np.diag([1.0, 1.0, np.inf]))
                Hm = Hm * 10.0 ** rs.choice([-12, -3, 0, 0, 4, 15]) * rs.choice([-1, 1])
                typical = model == 3 and sc in (1.0, 3.0) and 1e-3 * sc <= thr <= 0.1 * sc
                enabled += _check16_hom(HM.matrix_record(Hm), cc, thr * thr, uvc, stats if typical else None, order=trial % 3)
    assert enabled > 500
    print(f"hom (fp16 form): {stats[1]}/{stats[0]} non-inliers pass the filter ({100.0 * stats[1] / stats[0]:.3f} %)")
    assert stats[1] < 0.05 * stats[0]
    # NaN points and coordinates beyond the bound are never excluded; thresholds above 8 / coordinates above 8 hand over
    cols2 = [c.copy() for c in cols]
    cols2[0][::5] = np.nan
    cols2[3][1::5] = 8.5
    en, out = HM.prefilter16_hom(HM.matrix_record(Hgt), cols2, 1e-6, 7.0)
    assert en and not out[::5].any() and not out[1::5].any()
    assert not HM.prefilter16_hom(HM.matrix_record(Hgt), cols, 8.0000001 ** 2, 1.0)[0]
    assert not HM.prefilter16_hom(HM.matrix_record(Hgt), cols, 1e-6, 8.5)[0]


def test_absolute_pose_fp16_form_narrow_field_of_view_tiny_threshold():
    """with a narrow field of view (|x|, |y| <= 1e-3) and thresholds <= 1e-4 the relative part of the fp16 form's
    slack, G = 2^-11 (1 + 2^-6) (1 + max|x|,|y| + thr), is all that covers the rounding of coefficients just above 1 times
    |X|_1 ~ 1e4 - the regime where a missing margin would show.  Correspondences planted at thr (1 +- 1e-9 .. 1e-3); rotations
    whose third row is close to a coordinate axis (coefficients thr R_2 -+ R_a next to 1), all accumulation orders."""
    rs = np.random.RandomState(77)
    enabled = 0
    for trial in range(60):
        n = 1500
        depth = rs.uniform(3e3, 2.5e4, n)
        X = np.c_[rs.uniform(-1e-3, 1e-3, n) * depth, rs.uniform(-1e-3, 1e-3, n) * depth, depth]
        # camera looking down +Z, rotated by a small angle so that R_2 ~ (eps, eps, 1) and the rows R_0, R_1 stay near axes
        w = rs.randn(3) * [1e-4, 1e-4, 0.7][trial % 3]
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        q = _quat(R)
        t = np.r_[rs.uniform(-1, 1, 2), rs.uniform(-100, 100)]
        Xw = (X - t) @ R  # world points such that R Xw + t = X
        rec = HM.pose_record(q, t)
        for thr in (1e-6, 1e-5, 1e-4):
            xp, Xp = _plant_at_threshold_abs(rec, Xw, thr, rs)
            cols = [xp[:, 0], xp[:, 1], Xp[:, 0], Xp[:, 1], Xp[:, 2]]
            for order in range(3):
                enabled += _check16_abs(rec, cols, thr * thr, float(np.abs(xp).max()), None, order=order)
    assert enabled > 400
