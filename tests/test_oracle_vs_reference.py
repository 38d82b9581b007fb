"""Pins the oracle's restatement against the REFERENCE'S OWN SOURCES (oracle/_ref/libposelib_ref.so, built in
place from /root/reference by oracle/Makefile.ref against the Eigen-API shim).  Everything on the `ref` side is
PoseLib's code — front-ends (robust.cc), ransac_* (robust/ransac.cc), estimator classes, loop template, sampler,
minimal solvers, scoring and masks, refiners + LM (robust/bundle.cc, robust/optim/*.h), camera models — only Eigen
is not the real one (see oracle/ref_shim/ref_api.cc).  Skipped where neither /root/reference nor a prebuilt
library is present.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as R
from poselib_amd import synth

pytestmark = pytest.mark.skipif(not R.available(), reason="reference build (oracle/_ref) not available")


def both(fn, *a, **k):
    mine = getattr(O, fn)(*a, **k)
    with R.reference() as ref:
        theirs = getattr(ref, fn)(*a, **k)
    return mine, theirs


@pytest.mark.parametrize("seed,N,K", [(0, 10, 3), (1, 5000, 3), (7, 5000, 5), (123456789, 10000, 7), (42, 10000, 4),
                                      (2**63 + 11, 37, 5), (3, 7, 7)])
def test_sampler_indices_and_state(seed, N, K):
    (a, sa), (b, sb) = both("sampler_draw", seed, N, K, 4000)
    assert np.array_equal(a, b) and sa == sb


@pytest.mark.parametrize("seed,N,K", [(0, 100, 3), (5, 5000, 5), (9, 300, 4)])
def test_prosac_sampler(seed, N, K):
    (a, sa), (b, sb) = both("sampler_draw", seed, N, K, 3000, prosac=True, max_prosac=1000)
    assert np.array_equal(a, b) and sa == sb


def test_loop_control_functions():
    with R.reference():
        ref = O._lib
        rp = [[ref.orc_all_inlier_probability(i, n, k) for i in range(0, n + 1, max(1, n // 17))]
              for n, k in ((10, 5), (5000, 3), (10000, 7))]
        rd = [ref.orc_dynamic_max_iter(i, 5000, k, np.log(1 - p), m, 1000, 100000)
              for i in range(0, 5001, 53) for k in (3, 5) for p, m in ((0.9999, 3.0), (0.99, 1.0))]
    lib = O.lib()
    mp = [[lib.orc_all_inlier_probability(i, n, k) for i in range(0, n + 1, max(1, n // 17))]
          for n, k in ((10, 5), (5000, 3), (10000, 7))]
    md = [lib.orc_dynamic_max_iter(i, 5000, k, np.log(1 - p), m, 1000, 100000)
          for i in range(0, 5001, 53) for k in (3, 5) for p, m in ((0.9999, 3.0), (0.99, 1.0))]
    assert mp == rp and md == rd


@pytest.mark.parametrize("n,k,c,opt", [
    (100, 5, 50, dict(min_iterations=10, max_iterations=1000, success_prob=0.99, dyn_num_trials_mult=1.0)),
    (100, 5, 100, dict(min_iterations=50, max_iterations=10000)),
    (100, 5, 5, dict(min_iterations=10, max_iterations=50)),
    (100, 5, 95, dict(min_iterations=10, max_iterations=10000, dyn_num_trials_mult=3.0)),
    (5000, 3, 2500, dict()),
])
def test_ransac_loop_on_mock_estimator(n, k, c, opt):  # the loop of robust/ransac_impl.h itself
    a, b = both("mock_ransac", n, k, c, opt)
    a.pop("seconds"), b.pop("seconds")
    assert a == b


def test_cubic_solvers_bit_identical():
    rs = np.random.RandomState(5)
    co = rs.randn(2000, 3) * np.array([3.0, 10.0, 30.0])
    out = []
    for ctx in (None, R.reference):
        cm = ctx() if ctx else None
        if cm:
            cm.__enter__()
        lib = O._lib if cm else O.lib()
        res = []
        for c2, c1, c0 in co:
            r1 = C.c_double(0)
            ok = lib.orc_solve_cubic_single_real(c2, c1, c0, C.byref(r1))
            r3 = (C.c_double * 3)()
            n = lib.orc_solve_cubic_real(c2, c1, c0, r3)
            res.append((ok, r1.value if ok else 0.0, n, tuple(r3[:n])))
        if cm:
            cm.__exit__(None, None, None)
        out.append(res)
    assert out[0] == out[1]


def _bearing(rs, n):
    v = np.c_[rs.uniform(-0.6, 0.6, (n, 2)), np.ones(n)]
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _rot(rs):
    q = rs.randn(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_p3p_matches_reference():
    rs = np.random.RandomState(11)
    worst, exact, total = 0.0, 0, 0
    for _ in range(400):
        Rm, t = _rot(rs), rs.randn(3)
        x = _bearing(rs, 3)
        X = (Rm.T @ ((x * rs.uniform(2, 10, (3, 1))).T - t[:, None])).T
        a, b = both("p3p", x, X)
        assert a.shape == b.shape
        if len(a):
            worst = max(worst, np.abs(a - b).max())
            exact += int(np.array_equal(a, b))
            total += 1
    assert total > 350 and worst < 1e-9
    print(f"p3p: {exact}/{total} problems bit-identical, worst |diff| {worst:.2e}")


def test_p3p_on_outlier_contaminated_samples_incl_nan_poses():
    """samples drawn from a 70 %-outlier scene (BASELINE config 1): inconsistent triplets make p3p.cc emit NaN poses
    (about 13 % of all hypotheses); the oracle must emit exactly the same ones, the scorers rely on it"""
    sc = synth.absolute_pose_scene(5000, 0.7, 1001)
    x = (np.asarray(sc["p2d"]) - 500.0) / 1000.0
    X = np.asarray(sc["p3d"])
    idx, _ = O.sampler_draw(0, 5000, 3, 1500)
    nan = total = 0
    for s in idx.astype(int):
        xb = np.c_[x[s], np.ones(3)]
        xb /= np.linalg.norm(xb, axis=1, keepdims=True)
        a, b = both("p3p", xb, X[s])
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
        total += len(a)
        nan += int(np.isnan(a).any(axis=1).sum()) if len(a) else 0
    assert 0.05 < nan / total < 0.25


def _two_view(rs, n):
    Rm, t = _rot(rs) if rs.rand() < 0.3 else synth_small_rot(rs), rs.randn(3)
    x1 = _bearing(rs, n)
    X = x1 * rs.uniform(2, 10, (n, 1))
    x2 = (Rm @ X.T).T + t
    x2 /= np.linalg.norm(x2, axis=1, keepdims=True)
    return x1, x2


def synth_small_rot(rs):
    w = rs.randn(3) * 0.1
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _match_sets(A, B, tol):
    """each solution of A has a partner in B (sign-normalised), and vice versa"""
    if len(A) != len(B):
        return False
    if len(A) == 0:
        return True
    A = np.array([np.ravel(a) for a in A])
    B = np.array([np.ravel(b) for b in B])
    d = np.abs(A[:, None, :] - B[None, :, :]).max(axis=2)
    return d.min(axis=1).max() < tol and d.min(axis=0).max() < tol


def test_five_point_matches_reference():
    rs = np.random.RandomState(12)
    okE = okP = same_order = 0
    trials = 300
    for _ in range(trials):
        x1, x2 = _two_view(rs, 5)
        a, b = both("essential_5pt", x1, x2)
        okE += _match_sets(a, b, 1e-6)
        same_order += len(a) == len(b) and all(np.abs(p - q).max() < 1e-6 for p, q in zip(a, b))
        pa, pb = both("relpose_5pt", x1, x2)
        okP += _match_sets(list(pa), list(pb), 1e-6)
    # the null-space basis on the ref side comes from the shim's QR, not Eigen's: the solution SET agrees except
    # where a (near-)double root splits differently
    assert okE >= trials - 6 and okP >= trials - 6 and same_order >= trials - 6
    print(f"5pt: E sets {okE}/{trials}, pose sets {okP}/{trials}, same order {same_order}/{trials}")


def test_seven_point_matches_reference():
    rs = np.random.RandomState(13)
    ok = 0
    trials = 300
    for _ in range(trials):
        x1, x2 = _two_view(rs, 7)
        a, b = both("relpose_7pt", x1, x2)
        ok += len(a) == len(b) and all(np.array_equal(p, q) for p, q in zip(a, b))
    # BIT for bit since round 5: the cubic's coefficients are summed in the order relpose_7pt.cc:22-37 writes them, because
    # the last bits of a minimal F decide the SIGN of the F its refinement returns (optim_utils.h:57-72, next test)
    assert ok == trials
    print(f"7pt bit-identical: {ok}/{trials}")


def test_fundamental_matrices_come_back_with_the_reference_sources_sign():
    """FactorizedFundamentalMatrix(F) (optim_utils.h:57-72) runs JacobiSVD and negates U / V by their determinants: with a
    rank-2 F the sign of the refined matrix is the sign of a rounding-level singular value, i.e. a function of every bit of
    the input and of the SVD's operation order.  Oracle and reference sources share one Eigen-ordered two-sided Jacobi
    routine (oracle/eigen_shim/Eigen/src/JacobiSVD3x3.h) and the oracle's 7-point solver is bit-identical to the
    reference's, so whole runs now agree INCLUDING the sign (the oracle returned -F in 20 - 37 % of the runs).
    The 600-problem form of this test is tests/soak_fundamental_sign.py -> profiles/r05_soak_fundamental_sign.md."""
    rs = np.random.RandomState(2025)
    exact = 0
    runs = 0
    for i in range(20):
        n = int(rs.randint(8, 2500))
        d = synth.fundamental_scene(n, float(rs.uniform(0.05, 0.7)), 61000 + i)
        for fn in ("ransac_fundamental", "estimate_fundamental"):
            opt = {"max_error": float(rs.uniform(0.5, 3.0)), "ransac": {"seed": int(rs.randint(0, 1 << 30)), "max_iterations": 1500}}
            (Fa, ka, sa), (Fb, kb, sb) = both(fn, d["x1"], d["x2"], opt)
            assert (sa["iterations"], sa["refinements"], sa["num_inliers"]) == (sb["iterations"], sb["refinements"], sb["num_inliers"])
            assert np.array_equal(ka, kb)
            assert np.abs(np.asarray(Fa) - np.asarray(Fb)).max() <= 1e-12 * np.abs(Fb).max(), (i, fn)  # sign-sensitive
            exact += bool(np.array_equal(Fa, Fb))
            runs += 1
    print(f"fundamental runs bit-identical to the reference sources: {exact}/{runs}")
    assert exact >= runs - 2


def test_homography_4pt_matches_reference():
    rs = np.random.RandomState(14)
    worst = 0.0
    for i in range(300):
        x1 = _bearing(rs, 4)
        H = np.eye(3) + 0.3 * rs.randn(3, 3)
        x2 = (H @ x1.T).T
        x2 /= np.linalg.norm(x2, axis=1, keepdims=True)
        if i % 5 == 0:
            x2[0] = -x2[0]  # fails the cheirality pre-check (homography_4pt.cc:38-46)
        (na, a), (nb, b) = both("homography_4pt", x1, x2)
        assert na == nb
        if na:
            worst = max(worst, np.abs(a - b).max() / np.abs(a).max())
    assert worst < 1e-9


KINDS = [("reproj", "abs"), ("sampson_pose", "rel"), ("sampson_F", "fund"), ("homography", "hom")]


def _scene(tag, n, seed):
    """normalised (abs / rel) or pixel (fund / hom) correspondences plus one good model for the scoring tests"""
    quick = {"max_error": 1.5, "ransac": {"seed": 1, "max_iterations": 300, "min_iterations": 100}}
    if tag == "abs":
        d = synth.absolute_pose_scene(n, 0.4, seed)
        par = d["camera"]["params"]
        return (np.asarray(d["p2d"]) - par[-2:]) / par[0], np.asarray(d["p3d"]), np.r_[d["q_gt"], d["t_gt"]]
    if tag == "rel":
        d = synth.relative_pose_scene(n, 0.4, seed)
        p1, p2 = d["camera1"]["params"], d["camera2"]["params"]
        t = np.asarray(d["t_gt"]) / np.linalg.norm(d["t_gt"])
        return (np.asarray(d["x1"]) - p1[-2:]) / p1[0], (np.asarray(d["x2"]) - p2[-2:]) / p2[0], np.r_[d["q_gt"], t]
    if tag == "fund":
        d = synth.fundamental_scene(n, 0.4, seed)
        return d["x1"], d["x2"], O.ransac_fundamental(d["x1"], d["x2"], quick)[0]
    d = synth.homography_scene(n, 0.4, seed)
    return d["x1"], d["x2"], O.ransac_homography(d["x1"], d["x2"], quick)[0]


@pytest.mark.parametrize("kind,tag", KINDS)
def test_scores_and_masks(kind, tag):
    rs = np.random.RandomState(21)
    for seed in range(3):
        a, b, gt = _scene(tag, 3000, 50 + seed)
        thr2 = 4e-6 if tag in ("abs", "rel") else 4.0
        for pert in (0.0, 1e-4, 1e-2):
            m = np.asarray(gt, dtype=np.float64) + pert * rs.randn(*np.shape(gt))
            if kind in ("reproj", "sampson_pose"):
                m[:4] /= np.linalg.norm(m[:4])
            (sa, ca), (sb, cb) = both("score", kind, m, a, b, thr2)
            assert ca == cb and sa == pytest.approx(sb, rel=1e-12, abs=1e-300)
            ma, mb = both("inliers", kind, m, a, b, thr2)
            assert np.array_equal(ma, mb)


def test_normalize_points():
    x1, x2, _ = _scene("hom", 2000, 77)
    for flags in ((True, True, True), (True, True, False), (True, False, True), (False, True, True)):
        a, b = both("normalize_points", x1, x2, *flags)
        for p, q in zip(a, b):
            assert np.allclose(p, q, rtol=1e-13, atol=1e-13)


def _rel_parts(a, b):
    """relative pose, component by component: dR, d(t/|t|) and d|t|.  |t| is a gauge freedom (the reference's LM steps t
    along its tangent plane without renormalising, optim/relative.h:94-152): the reference does not reproduce it
    across its own builds (tests/golden/make_gauge.py), so it gets its own, wider bound - stated, not hidden."""
    from golden.make_gauge import parts

    return parts(a, b)


from golden.make_gauge import DT_LEN_BOUND as REL_DT_LEN_BOUND  # noqa: E402  (|t| gauge: guarded at 1e-3, see make_gauge.py)


def _cmp_run(fn, a, b, opt, tol=1e-8):
    (ma, ka, sa), (mb, kb, sb) = both(fn, a, b, opt)
    for k in ("iterations", "refinements", "num_inliers"):
        assert sa[k] == sb[k], (k, sa, sb)
    assert np.array_equal(ka, kb)
    assert sa["model_score"] == pytest.approx(sb["model_score"], rel=1e-9)
    if fn == "ransac_relpose":
        p = _rel_parts(ma, mb)
        assert p["dR"] < tol and p["dt_dir"] < tol, p
        assert p["dt_len"] <= REL_DT_LEN_BOUND, p
        return sa
    sc = max(1.0, np.abs(ma).max())
    assert np.abs(ma - mb).max() / sc < tol
    return sa


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_full_lo_ransac_absolute_pose(seed):
    x, X, _ = _scene("abs", 2000, 300 + seed)
    st = _cmp_run("ransac_pnp", x, X, {"max_error": 1e-3, "ransac": {"seed": seed, "max_iterations": 3000}})
    assert st["num_inliers"] > 800


@pytest.mark.parametrize("seed", [0, 1])
def test_full_lo_ransac_relative_pose(seed):
    x1, x2, _ = _scene("rel", 1500, 400 + seed)
    st = _cmp_run("ransac_relpose", x1, x2, {"max_error": 1e-3, "ransac": {"seed": seed, "max_iterations": 2000}})
    assert st["num_inliers"] > 500


@pytest.mark.parametrize("seed,rfc", [(0, False), (1, True)])
def test_full_lo_ransac_fundamental(seed, rfc):
    x1, x2, _ = _scene("fund", 1500, 500 + seed)
    st = _cmp_run("ransac_fundamental", x1, x2,
                  {"max_error": 1.5, "real_focal_check": rfc, "ransac": {"seed": seed, "max_iterations": 2000}}, 1e-6)
    assert st["num_inliers"] > 500


@pytest.mark.parametrize("seed", [0, 1])
def test_full_lo_ransac_homography(seed):
    x1, x2, _ = _scene("hom", 1500, 600 + seed)
    st = _cmp_run("ransac_homography", x1, x2, {"max_error": 1.5, "ransac": {"seed": seed, "max_iterations": 2000}})
    assert st["num_inliers"] > 500


# ---------------------------------------------------------------------------------------------------- LM refiners
# ref side: the reference's refiner classes + NormalAccumulator + robust losses + lm_impl loop (robust/optim/*.h)
BOPTS = [
    {"loss_type": "TRUNCATED", "max_iterations": 25},                       # the LO settings of every estimator
    {"loss_type": "CAUCHY"},                                                # BundleOptions defaults
    {"loss_type": "TRIVIAL", "max_iterations": 10},
    {"loss_type": "HUBER", "damping": 1},
    {"loss_type": "TRUNCATED_CAUCHY", "lambda_update": 1, "lambda_factor": 5.0},
    {"loss_type": "TRUNCATED_LE_ZACH", "max_iterations": 30},
]


REFINE_FLAGS = [{"refine_focal_length": True},
                {"refine_principal_point": True},
                {"refine_focal_length": True, "refine_principal_point": True},
                {"refine_focal_length": True, "refine_extra_params": True},
                {"refine_focal_length": True, "refine_principal_point": True, "refine_extra_params": True}]


def _camera_cases(d):
    f, cx, cy = d["camera"]["params"]
    pix = np.asarray(d["p2d"])
    par = [f, f, cx, cy, -0.05, 0.01, 1e-3, -5e-4]
    return [(d["camera"], pix),
            (dict(d["camera"], model="PINHOLE", params=[f, f, cx, cy]), pix),
            (dict(d["camera"], model="OPENCV", params=par), synth.opencv_distort_pixels(pix, par))]


def _off_calibration(cam, rs, rel=0.03, pp=4.0):
    """start the refinement from intrinsics that are a few per cent off"""
    par = np.array(cam["params"], dtype=np.float64)
    nf = {"SIMPLE_PINHOLE": 1, "PINHOLE": 2, "OPENCV": 2}[cam["model"]]
    par[:nf] *= 1.0 + rel * rs.randn(nf)
    par[nf:nf + 2] += pp * rs.randn(2)
    return dict(cam, params=[float(v) for v in par])


# ------------------------------------------------------------------------------- front-ends (the real robust.cc)
def _opencv_scene(n, seed):
    d = synth.absolute_pose_scene(n, 0.4, seed)
    f, cx, cy = d["camera"]["params"]
    par = [f, f, cx, cy, -0.08, 0.02, 1e-3, -5e-4]
    cam = dict(d["camera"], model="OPENCV", params=par)
    return dict(d, camera=cam, p2d=synth.opencv_distort_pixels(np.asarray(d["p2d"]), par))


@pytest.mark.parametrize("model", ["SIMPLE_PINHOLE", "PINHOLE", "OPENCV"])
def test_unproject(model):
    rs = np.random.RandomState(3)
    pix = rs.uniform(50, 950, (500, 2))
    cam = {"SIMPLE_PINHOLE": {"model": "SIMPLE_PINHOLE", "params": [800.0, 500.0, 480.0]},
           "PINHOLE": {"model": "PINHOLE", "params": [800.0, 790.0, 500.0, 480.0]},
           "OPENCV": {"model": "OPENCV", "params": [800.0, 790.0, 500.0, 480.0, -0.08, 0.02, 1e-3, -5e-4]}}[model]
    a, b = both("unproject", cam, pix)
    assert np.array_equal(a, b)


def _cmp_frontend(fn, args, opt, tol=1e-8):
    (ma, ka, sa), (mb, kb, sb) = both(fn, *args, opt)
    for k in ("iterations", "refinements", "num_inliers"):
        assert sa[k] == sb[k], (k, sa, sb)
    assert np.array_equal(ka, kb)
    assert sa["inlier_ratio"] == sb["inlier_ratio"]
    assert sa["model_score"] == pytest.approx(sb["model_score"], rel=1e-9)
    if fn == "estimate_relative_pose":
        p = _rel_parts(ma, mb)
        assert p["dR"] < tol and p["dt_dir"] < tol, p
        assert p["dt_len"] <= REL_DT_LEN_BOUND, p
    else:
        assert np.abs(ma - mb).max() / max(1.0, np.abs(ma).max()) < tol
    return sa, bool(np.array_equal(ma, mb))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_estimate_absolute_pose(seed):  # robust.cc estimate_absolute_pose, BASELINE config 0 / 1 shape
    d = synth.absolute_pose_scene(1500, 0.5, 800 + seed) if seed < 3 else _opencv_scene(1500, 803)
    opt = {"max_error": 4.0, "ransac": {"seed": seed, "max_iterations": 4000}}
    st, exact = _cmp_frontend("estimate_absolute_pose", (d["p2d"], d["p3d"], d["camera"]), opt)
    assert st["num_inliers"] > 600
    print("estimate_absolute_pose bit-identical:", exact)


@pytest.mark.parametrize("flags", REFINE_FLAGS[2:], ids=lambda f: "+".join(sorted(k[7:] for k in f)))
def test_estimate_absolute_pose_refining_intrinsics(flags):
    """robust.cc:36-126 with opt.bundle.refine_*: RANSAC + LO at the given calibration, the final bundle moves the camera"""
    rs = np.random.RandomState(37)
    d = synth.absolute_pose_scene(1200, 0.4, 811)
    for cam, pix in _camera_cases(d):
        cam0 = _off_calibration(cam, rs, 0.01, 2.0)
        opt = {"max_error": 8.0, "ransac": {"seed": 3, "max_iterations": 2000}, "bundle": dict(flags)}
        (ma, ka, sa, ca), (mb, kb, sb, cb) = both("estimate_absolute_pose", pix, d["p3d"], cam0, opt, return_camera=True)
        for k in ("iterations", "refinements", "num_inliers"):
            assert sa[k] == sb[k], (k, sa, sb)
        assert np.array_equal(ka, kb) and sa["num_inliers"] > 300
        assert np.abs(ma - mb).max() < 1e-8 and np.abs(ca - cb).max() < 1e-6 * max(1.0, np.abs(ca).max())
        assert abs(ca[0] - d["camera"]["params"][0]) < abs(cam0["params"][0] - d["camera"]["params"][0])  # focal moved towards the truth


@pytest.mark.parametrize("seed", [0, 1])
def test_estimate_relative_pose(seed):
    d = synth.relative_pose_scene(1500, 0.4, 810 + seed)
    opt = {"max_error": 1.5, "ransac": {"seed": seed, "max_iterations": 2000}}
    st, _ = _cmp_frontend("estimate_relative_pose", (d["x1"], d["x2"], d["camera1"], d["camera2"]), opt)
    assert st["num_inliers"] > 600


@pytest.mark.parametrize("seed,rfc", [(0, False), (1, True)])
def test_estimate_fundamental(seed, rfc):
    d = synth.fundamental_scene(1500, 0.4, 820 + seed)
    opt = {"max_error": 1.5, "real_focal_check": rfc, "ransac": {"seed": seed, "max_iterations": 2000}}
    st, _ = _cmp_frontend("estimate_fundamental", (d["x1"], d["x2"]), opt, 1e-6)
    assert st["num_inliers"] > 600


@pytest.mark.parametrize("seed", [0, 1])
def test_estimate_homography(seed):
    d = synth.homography_scene(1500, 0.4, 830 + seed)
    opt = {"max_error": 1.5, "ransac": {"seed": seed, "max_iterations": 2000}}
    st, _ = _cmp_frontend("estimate_homography", (d["x1"], d["x2"]), opt)
    assert st["num_inliers"] > 600


def test_estimate_with_prosac_and_initial_model():
    d = synth.absolute_pose_scene(1200, 0.5, 840)
    opt = {"max_error": 4.0, "ransac": {"seed": 5, "max_iterations": 3000, "progressive_sampling": True,
                                        "max_prosac_iterations": 500}}
    _cmp_frontend("estimate_absolute_pose", (d["p2d"], d["p3d"], d["camera"]), opt)
    opt = {"max_error": 4.0, "ransac": {"seed": 6, "max_iterations": 3000, "score_initial_model": True}}
    init = np.r_[d["q_gt"], d["t_gt"]]
    (ma, ka, sa), (mb, kb, sb) = both("estimate_absolute_pose", d["p2d"], d["p3d"], d["camera"], opt, init)
    assert sa["iterations"] == sb["iterations"] and sa["num_inliers"] == sb["num_inliers"] and np.array_equal(ka, kb)
    assert np.abs(ma - mb).max() < 1e-9


@pytest.mark.parametrize("ransac", [{"success_prob": 1.0, "min_iterations": 5, "max_iterations": 300},
                                    {"success_prob": 1.0, "min_iterations": 400, "max_iterations": 5000, "dyn_num_trials_mult": 0.5},
                                    {"min_iterations": 9000, "max_iterations": 300}, {"max_iterations": 0}, {"max_iterations": 1},
                                    {"success_prob": 0.5, "dyn_num_trials_mult": 10.0, "min_iterations": 0}])
def test_odd_loop_control_options(ransac):
    """Loop control outside the usual range.  success_prob = 1 makes ransac_impl.h:70-71 cast +inf to size_t - undefined
    behaviour that this toolchain (gcc, x86-64) resolves to 0, i.e. the run stops right after min_iterations: the
    oracle (same literal code, same compiler) and the reference's sources must agree, and the product spells the
    conversion out (driver.cc dynamic_max_iter; GPU soak with SOAK_FUZZ)."""
    d = synth.homography_scene(400, 0.3, 850)
    opt = {"max_error": 1.5, "ransac": dict(ransac, seed=3)}
    st, _ = _cmp_frontend("estimate_homography", (d["x1"], d["x2"]), opt)
    if ransac.get("success_prob") == 1.0 and st["num_inliers"] > 0:
        assert st["iterations"] <= ransac["min_iterations"] + 2 < ransac["max_iterations"]  # stops right after min_iterations


EXACT = []  # per LM comparison: parameters and final cost bit-identical?


def _bstats(st):
    return {k: getattr(st, k) for k, _ in st._fields_}


def _cmp_lm(sa, sb, ma, mb, tol=1e-9):
    a, b = _bstats(sa), _bstats(sb)
    assert a["iterations"] == b["iterations"] and a["invalid_steps"] == b["invalid_steps"], (a, b)
    assert a["initial_cost"] == pytest.approx(b["initial_cost"], rel=1e-12)
    assert a["cost"] == pytest.approx(b["cost"], rel=1e-9, abs=1e-18)
    assert np.abs(np.asarray(ma) - np.asarray(mb)).max() / max(1.0, np.abs(ma).max()) < tol
    EXACT.append(bool(np.array_equal(ma, mb) and a["cost"] == b["cost"]))


def _perturb_pose(p, rs, s):
    p = np.asarray(p, dtype=np.float64) + s * rs.randn(7)
    p[:4] /= np.linalg.norm(p[:4])
    return p


@pytest.mark.parametrize("bo", BOPTS)
def test_lm_absolute_pose_calibrated(bo):
    rs = np.random.RandomState(31)
    x, X, gt = _scene("abs", 800, 700)
    bo = dict(bo, loss_scale=2e-3)
    start = _perturb_pose(gt, rs, 0.01)
    (pa, sa), (pb, sb) = both("bundle_adjust", x, X, {"model": -1}, start, bo)
    _cmp_lm(sa, sb, pa, pb)


@pytest.mark.parametrize("model", ["SIMPLE_PINHOLE", "PINHOLE", "OPENCV"])
def test_lm_absolute_pose_through_camera_models(model):  # the final polish of estimate_absolute_pose (pixels)
    rs = np.random.RandomState(32)
    d = synth.absolute_pose_scene(600, 0.3, 710)
    f, cx, cy = d["camera"]["params"]
    pix = np.asarray(d["p2d"])
    if model == "SIMPLE_PINHOLE":
        cam = d["camera"]
    elif model == "PINHOLE":
        cam = dict(d["camera"], model="PINHOLE", params=[f, f, cx, cy])
    else:
        par = [f, f, cx, cy, -0.05, 0.01, 1e-3, -5e-4]
        cam = dict(d["camera"], model="OPENCV", params=par)
        pix = synth.opencv_distort_pixels(pix, par)
    start = _perturb_pose(np.r_[d["q_gt"], d["t_gt"]], rs, 0.005)
    for bo in ({"loss_type": "TRUNCATED", "loss_scale": 2.0, "max_iterations": 25}, {"loss_type": "CAUCHY", "loss_scale": 1.0}):
        (pa, sa), (pb, sb) = both("bundle_adjust", pix, d["p3d"], cam, start, bo)
        _cmp_lm(sa, sb, pa, pb)


@pytest.mark.parametrize("flags", REFINE_FLAGS, ids=lambda f: "+".join(sorted(k[7:] for k in f)))
def test_lm_absolute_pose_with_intrinsics(flags):
    """bundle_adjust with refine_focal_length / refine_principal_point / refine_extra_params (robust/bundle.cc:93-118
    -> AbsolutePoseRefiner with Camera::get_param_refinement_idx, camera_models.cc: project_with_jac's parameter block)"""
    rs = np.random.RandomState(36)
    d = synth.absolute_pose_scene(500, 0.2, 745)
    start = _perturb_pose(np.r_[d["q_gt"], d["t_gt"]], rs, 0.003)
    for cam, pix in _camera_cases(d):
        cam0 = _off_calibration(cam, rs)
        for bo in ({"loss_type": "CAUCHY", "loss_scale": 1.0}, {"loss_type": "TRUNCATED", "loss_scale": 6.0, "max_iterations": 25}):
            (pa, ca, sa), (pb, cb, sb) = both("bundle_adjust_camera", pix, d["p3d"], cam0, start, dict(bo, **flags))
            _cmp_lm(sa, sb, np.r_[pa, ca / 1000.0], np.r_[pb, cb / 1000.0])
            assert ca.shape == (len(cam["params"]),)
            if not flags.get("refine_extra_params") and cam["model"] == "OPENCV":
                assert np.array_equal(ca[4:], np.asarray(cam0["params"])[4:])   # untouched parameters pass through
            if not flags.get("refine_principal_point"):
                npp = {"SIMPLE_PINHOLE": 1, "PINHOLE": 2, "OPENCV": 2}[cam["model"]]
                assert np.array_equal(ca[npp:npp + 2], np.asarray(cam0["params"])[npp:npp + 2])


@pytest.mark.parametrize("bo", BOPTS)
def test_lm_relative_pose(bo):
    rs = np.random.RandomState(33)
    x1, x2, gt = _scene("rel", 800, 720)
    keep = O.inliers("sampson_pose", gt, x1, x2, 2e-5)
    start = _perturb_pose(gt, rs, 0.01)
    (pa, sa), (pb, sb) = both("refine", "relpose", x1[keep], x2[keep], start, dict(bo, loss_scale=1e-3))
    _cmp_lm(sa, sb, pa, pb)


@pytest.mark.parametrize("bo", BOPTS)
def test_lm_fundamental(bo):
    rs = np.random.RandomState(34)
    x1, x2, F = _scene("fund", 800, 730)
    start = F + 1e-3 * np.abs(F).max() * rs.randn(3, 3)
    (Fa, sa), (Fb, sb) = both("refine", "fundamental", x1, x2, start, dict(bo, loss_scale=1.5))
    Fa, Fb = Fa / np.linalg.norm(Fa), Fb / np.linalg.norm(Fb)
    _cmp_lm(sa, sb, Fa, Fb, 1e-8)


@pytest.mark.parametrize("bo", BOPTS)
def test_lm_homography(bo):
    rs = np.random.RandomState(35)
    x1, x2, H = _scene("hom", 800, 740)
    start = H + 1e-3 * np.abs(H).max() * rs.randn(3, 3)
    (Ha, sa), (Hb, sb) = both("refine", "homography", x1, x2, start, dict(bo, loss_scale=1.5))
    _cmp_lm(sa, sb, Ha, Hb, 1e-8)


def test_lm_bit_identity_summary():
    """runs after the LM comparisons (file order): with the Cholesky written in one operation order on both
    sides, the reference's refiner classes + LM loop and the oracle's restatement agree to the bit"""
    if not EXACT:
        pytest.skip("LM comparisons deselected")
    print(f"LM comparisons bit-identical: {sum(EXACT)}/{len(EXACT)}")
    assert sum(EXACT) >= 0.9 * len(EXACT)
