"""The product's device math (poselib_amd/csrc/pl_*.h), compiled for the HOST by tests/hostmath (test-only
build of the very same headers hipcc compiles into the kernels), against the oracle — on CPU, so that
kernel arithmetic is validated without a GPU.  With the same libm both sides must agree to the bit for
everything except the 5-point solver (different, but equivalent, summation order of the constraint rows)."""
import numpy as np
import pytest

import hostmath_lib as HM
import oracle_lib as O
from poselib_amd import synth


def bear(p):
    b = np.c_[p, np.ones(len(p))]
    return b / np.sqrt((b * b).sum(1))[:, None]


def test_sampler_bit_exact_and_positions():
    for N, K in [(5000, 3), (200, 3), (10000, 7), (10000, 4), (5000, 5), (7, 7)]:
        idx, pos, total = HM.draw_samples(3, N, K, 3000)
        ref, state = O.sampler_draw(3, N, K, 3000)
        assert (idx.astype(np.uint64) == ref).all()
        assert pos[0] == 0 and (np.diff(pos.astype(np.int64)) >= K).all()
        G = 0x9E3779B97F4A7C15
        assert total == ((state - 3) * pow(G, -1, 2**64)) % 2**64


def test_unproject_bit_exact():
    rs = np.random.RandomState(0)
    pix = rs.uniform(0, 1000, (500, 2))
    cams = [("SIMPLE_PINHOLE", 0, [1000.0, 500.0, 480.0]), ("PINHOLE", 1, [900.0, 950.0, 500.0, 480.0]),
            ("OPENCV", 4, [1000.0, 1010.0, 500.0, 480.0, -0.1, 0.02, 0.001, -0.002])]
    for name, mid, params in cams:
        a = O.unproject({"model": name, "params": params}, pix)
        b = HM.unproject(HM.camera_params(mid, params), pix)
        assert (a == b).all(), name


def test_p3p_and_reprojection_score_bit_exact():
    d = synth.absolute_pose_scene(3000, 0.6, 1001)
    un = O.unproject(d["camera"], d["p2d"])
    cols = [un[:, 0], un[:, 1], d["p3d"][:, 0], d["p3d"][:, 1], d["p3d"][:, 2]]
    idx, _, _ = HM.draw_samples(0, 3000, 3, 1500)
    thr2 = (12 / 1000.0) ** 2
    nmodels = 0
    for it in range(1500):
        s = idx[it]
        xb = bear(un[s])
        recs = HM.solve("abs", xb, d["p3d"][s])
        ref = O.p3p(xb, d["p3d"][s])
        assert len(recs) == len(ref)
        for r, o in zip(recs, ref):
            # (a degenerate sample makes BOTH sides return a NaN model; it scores zero inliers on both)
            assert np.array_equal(r[:7], o, equal_nan=True)
            nmodels += 1
            if it < 150:
                sc, cnt, flags, r2 = HM.score("abs", r, cols, thr2)
                osc, ocnt = O.score("reproj", o, un, d["p3d"], thr2)
                assert cnt == ocnt
                assert sc == osc  # same summation order: the same bits
                # final-mask arithmetic (division form) may differ from the scoring form only at the threshold
                m = HM.mask_abs(r, cols, thr2)
                assert (m == O.inliers("reproj", o, un, d["p3d"], thr2)).all()
    assert nmodels > 1500


def test_two_view_solvers_and_scores():
    d = synth.relative_pose_scene(2000, 0.4, 1002)
    a = O.unproject(d["camera1"], d["x1"])
    b = O.unproject(d["camera2"], d["x2"])
    cols = [a[:, 0], a[:, 1], b[:, 0], b[:, 1]]
    thr2 = 1e-6
    # 7-point: bit exact
    idx, _, _ = HM.draw_samples(1, 2000, 7, 400)
    for it in range(400):
        s = idx[it]
        Fs = O.relpose_7pt(bear(a[s]), bear(b[s]))
        recs = HM.solve("fund", bear(a[s]), bear(b[s]))
        assert len(Fs) == len(recs)
        for F, r in zip(Fs, recs):
            assert (r[7:16].reshape(3, 3) == F).all()
            if it < 40:
                sc, cnt, _, _ = HM.score("fund", r, cols, thr2)
                osc, ocnt = O.score("sampson_F", F, a, b, thr2)
                assert cnt == ocnt and sc == osc  # (r^2 or thr^2 per correspondence, in order: the same bits)
    # homography: bit exact
    idx, _, _ = HM.draw_samples(2, 2000, 4, 400)
    for it in range(400):
        s = idx[it]
        n, H = O.homography_4pt(bear(a[s]), bear(b[s]))
        recs = HM.solve("hom", bear(a[s]), bear(b[s]))
        assert n == len(recs)
        if n:
            assert (recs[0][7:16].reshape(3, 3) == H).all()
            sc, cnt, _, _ = HM.score("hom", recs[0], cols, thr2)
            osc, ocnt = O.score("homography", H, a, b, thr2)
            assert cnt == ocnt and sc == osc
    # 5-point: the constraint polynomials are accumulated in the oracle's association order (pl_solver_rel.h
    # mac_lin_lin / mac_quad_lin), everything downstream is the same sequence of operations: bit-identical poses
    idx, _, _ = HM.draw_samples(0, 2000, 5, 600)
    total = 0
    for it in range(600):
        s = idx[it]
        ref = O.relpose_5pt(bear(a[s]), bear(b[s]))
        recs = HM.solve("rel", bear(a[s]), bear(b[s]))
        assert len(ref) == len(recs), it
        for o, r in zip(ref, recs):
            assert (o == r[:7]).all(), (it, o, r[:7])
        total += len(ref)
    assert total > 300
    pose, mask, st = O.ransac_relpose(a, b, dict(max_error=1e-3))
    rec = HM.pose_record(pose[:4], pose[4:], True)
    sc, cnt, flags, _ = HM.score("rel", rec, cols, thr2)
    osc, ocnt = O.score("sampson_pose", pose, a, b, thr2)
    assert cnt == ocnt and sc == osc
    assert (flags == O.inliers("sampson_pose", pose, a, b, thr2)).all()


def test_lm_refiners_bit_exact():
    d = synth.absolute_pose_scene(1500, 0.5, 1000)
    un = O.unproject(d["camera"], d["p2d"])
    cols = [un[:, 0], un[:, 1], d["p3d"][:, 0], d["p3d"][:, 1], d["p3d"][:, 2]]
    q = d["q_gt"] + 0.01 * np.array([0.3, -0.2, 0.5, 0.1])
    q /= np.linalg.norm(q)
    p0 = np.r_[q, d["t_gt"] + np.array([0.01, -0.02, 0.015])]
    for loss, scale, mi in [(1, 0.012, 25), (3, 0.001, 100), (2, 0.002, 50), (0, 1.0, 10), (4, 0.01, 30), (5, 0.01, 30)]:
        ref, st = O.bundle_adjust(un, d["p3d"], {"model": "NULL", "params": []}, p0,
                                  dict(loss_type=loss, loss_scale=scale, max_iterations=mi))
        got, it, _ = HM.lm("abs", cols, p0, HM.lm_options(mi, loss, scale))
        assert it == st.iterations and (got[:7] == ref).all(), loss
    # final bundle: pixel points * scale, rescaled pinhole camera, inlier mask (robust.cc:103-123)
    scale = 1.0 / 1000
    camp = [1000 * scale, 500 * scale, 500 * scale]
    m = d["inlier_gt"]
    ref, st = O.bundle_adjust(d["p2d"][m] * scale, d["p3d"][m], {"model": "SIMPLE_PINHOLE", "params": camp}, p0,
                              dict(loss_type=3, loss_scale=scale))
    colsp = [d["p2d"][:, 0], d["p2d"][:, 1]] + cols[2:]
    got, it, _ = HM.lm("abs", colsp, p0, HM.lm_options(100, 3, scale), HM.camera_params(0, camp), point_scale=scale,
                       mask=m)
    assert it == st.iterations and (got[:7] == ref).all()

    dh = synth.homography_scene(1200, 0.3, 1003, noise_px=0.2)
    s, a, b, _, _ = O.normalize_points(dh["x1"], dh["x2"])
    H0, _, _ = O.ransac_homography(a, b, dict(max_error=1.0 / s))
    Hp = H0 + 1e-3 * np.arange(9).reshape(3, 3) / 9
    ref, st = O.refine("homography", a, b, Hp, dict(loss_type=1, loss_scale=1.0 / s, max_iterations=25))
    got, it, _ = HM.lm("hom", [a[:, 0], a[:, 1], b[:, 0], b[:, 1]], Hp.reshape(9), HM.lm_options(25, 1, 1.0 / s))
    assert it == st.iterations and (got[:9] == ref.reshape(9)).all()

    dr = synth.relative_pose_scene(1500, 0.4, 1002)
    a = O.unproject(dr["camera1"], dr["x1"])
    b = O.unproject(dr["camera2"], dr["x2"])
    thr = 1e-3
    pose, _, _ = O.ransac_relpose(a, b, dict(max_error=thr))
    q = pose[:4] + 0.002 * np.array([0.3, -0.2, 0.5, 0.1])
    q /= np.linalg.norm(q)
    p0 = np.r_[q, pose[4:] + np.array([0.01, -0.005, 0.004])]
    m5 = O.inliers("sampson_pose", p0, a, b, 5 * thr * thr)
    ref, st = O.refine("relpose", a[m5], b[m5], p0, dict(loss_type=1, loss_scale=thr, max_iterations=25))
    got, it, skipped = HM.lm("rel", [a[:, 0], a[:, 1], b[:, 0], b[:, 1]], p0, HM.lm_options(25, 1, thr),
                             prefilter_thr2=5 * thr * thr)
    assert not skipped and it == st.iterations and (got[:7] == ref).all()


def test_lm_with_intrinsics_bit_exact():
    """k_lm_cam's arithmetic (pl_refine_cam.h, serial on the host) against the oracle's bundle_adjust with refine_* flags
    (itself bit-identical with the reference sources: test_oracle_vs_reference.py): pose, camera, iterations, cost."""
    rs = np.random.RandomState(5)
    d = synth.absolute_pose_scene(700, 0.3, 1010)
    f, cx, cy = d["camera"]["params"]
    pix = np.asarray(d["p2d"])
    par = [f, f, cx, cy, -0.05, 0.01, 1e-3, -5e-4]
    cases = [(0, "SIMPLE_PINHOLE", [f, cx, cy], pix), (1, "PINHOLE", [f, f, cx, cy], pix),
             (4, "OPENCV", par, synth.opencv_distort_pixels(pix, par))]
    q = d["q_gt"] + 0.003 * rs.randn(4)
    p0 = np.r_[q / np.linalg.norm(q), d["t_gt"] + 0.003 * rs.randn(3)]
    m = d["inlier_gt"]
    checked = 0
    for mid, name, params, px in cases:
        nf = 1 if mid == 0 else 2
        off = np.array(params, dtype=np.float64)
        off[:nf] *= 1.0 + 0.02 * rs.randn(nf)
        off[nf:nf + 2] += 3.0 * rs.randn(2)
        cols = [px[:, 0], px[:, 1], d["p3d"][:, 0], d["p3d"][:, 1], d["p3d"][:, 2]]
        for flags in (1, 2, 3, 5, 7):
            for loss, lscale, mi in ((3, 1.0, 100), (1, 6.0, 25), (2, 2.0, 40)):
                bo = dict(loss_type=loss, loss_scale=lscale, max_iterations=mi, refine_focal_length=bool(flags & 1),
                          refine_principal_point=bool(flags & 2), refine_extra_params=bool(flags & 4))
                rp, rc, st = O.bundle_adjust_camera(px[m], d["p3d"][m], {"model": name, "params": list(off)}, p0, bo)
                gp, gc, it, costs = HM.lm_cam(cols, p0, HM.lm_options(mi, loss, lscale), HM.camera_params(mid, list(off)), flags,
                                              mask=m)
                assert it == st.iterations, (name, flags, loss)
                assert (gp[:7] == rp).all() and (gc == rc).all(), (name, flags, loss, np.abs(gp[:7] - rp).max(), np.abs(gc - rc).max())
                assert costs[1] == st.cost and costs[0] == st.initial_cost
                checked += 1
    assert checked == 45
    # the final bundle's form (robust.cc:103-123): pixels * 1/f, camera rescaled, all flags
    scale = 1.0 / f
    camp = [f * scale, cx * scale, cy * scale]
    bo = dict(loss_type=3, loss_scale=scale, refine_focal_length=True, refine_principal_point=True)
    rp, rc, st = O.bundle_adjust_camera(pix[m] * scale, d["p3d"][m], {"model": "SIMPLE_PINHOLE", "params": camp}, p0, bo)
    cols = [pix[:, 0], pix[:, 1], d["p3d"][:, 0], d["p3d"][:, 1], d["p3d"][:, 2]]
    gp, gc, it, _ = HM.lm_cam(cols, p0, HM.lm_options(100, 3, scale), HM.camera_params(0, camp), 3, point_scale=scale, mask=m)
    assert it == st.iterations and (gp[:7] == rp).all() and (gc == rc).all()


def test_lm_pose_refiners_bit_exact_on_rough_starts_and_all_lm_options():
    """Large LM steps (rough starting poses, few correspondences, Marquardt damping, both lambda updates) make every rounding of
    the quaternion step count: this is the regime in which the device's sin / cos pair differed from the reference's sincos()
    (DESIGN §5, round 3).  Absolute and relative pose, the device arithmetic on the host against the oracle, bit for bit."""
    rs = np.random.RandomState(77)
    checked = 0
    for k in range(40):
        n = int(rs.choice([8, 15, 60, 250]))
        lu, dm = int(rs.randint(2)), int(rs.randint(2))
        loss = int(rs.choice([0, 2, 3]))
        d = synth.absolute_pose_scene(n, 0.0, 1200 + k)
        un = O.unproject(d["camera"], d["p2d"])
        q = d["q_gt"] + 0.15 * rs.randn(4)
        p0 = np.r_[q / np.linalg.norm(q), d["t_gt"] + 0.3 * rs.randn(3)]
        bo = dict(loss_type=loss, loss_scale=0.01, max_iterations=60, lambda_update=lu, damping=dm)
        ref, st = O.bundle_adjust(un, d["p3d"], {"model": "NULL", "params": []}, p0, bo)
        got, it, _ = HM.lm("abs", [un[:, 0], un[:, 1], d["p3d"][:, 0], d["p3d"][:, 1], d["p3d"][:, 2]], p0,
                           HM.lm_options(60, loss, 0.01, lambda_update=lu, damping=dm))
        assert it == st.iterations and np.array_equal(got[:7], ref, equal_nan=True), ("abs", k, n, lu, dm, loss)
        dr = synth.relative_pose_scene(n, 0.0, 1300 + k)
        a, b = O.unproject(dr["camera1"], dr["x1"]), O.unproject(dr["camera2"], dr["x2"])
        q = dr["q_gt"] + 0.1 * rs.randn(4)
        t = dr["t_gt"] / np.linalg.norm(dr["t_gt"]) + 0.2 * rs.randn(3)
        p0 = np.r_[q / np.linalg.norm(q), t]
        bo = dict(loss_type=loss, loss_scale=1e-3, max_iterations=60, lambda_update=lu, damping=dm)
        ref, st = O.refine("relpose", a, b, p0, bo)
        got, it, _ = HM.lm("rel", [a[:, 0], a[:, 1], b[:, 0], b[:, 1]], p0, HM.lm_options(60, loss, 1e-3, lambda_update=lu, damping=dm))
        assert it == st.iterations and np.array_equal(got[:7], ref, equal_nan=True), ("rel", k, n, lu, dm, loss)
        checked += 2
    assert checked == 80


def _centered(d):
    f, cx, cy = d["camera"]["params"]
    return np.asarray(d["p2d"]) - np.array([cx, cy])


def test_p35pf_device_header_bit_exact():
    """pl_solver_p35pf.h (the serial statement of the solver whose steps the kernels of focal.hip distribute over lanes) against
    the oracle's restatement of the reference's template solver: the same solutions in the same order, bit for bit, on exact and
    noisy minimal problems"""
    total = 0
    for seed in range(300):
        d = synth.absolute_pose_scene(4, 0.0, 9000 + seed, noise_px=0.0 if seed % 2 else 1.0)
        x = _centered(d)
        op, of = O.p35pf(x, d["p3d"])
        hp, hf = HM.p35pf(x, d["p3d"], stride=1 + seed % 3)
        assert len(hf) == len(of) and np.array_equal(hp, op) and np.array_equal(hf, of), seed
        total += len(of)
    assert total > 1000
    # degenerate input (two identical correspondences): whatever comes out is the same on both sides
    d = synth.absolute_pose_scene(4, 0.0, 1)
    x, X = _centered(d), np.array(d["p3d"])
    x[1], X[1] = x[0], X[0]
    op, of = O.p35pf(x, X)
    hp, hf = HM.p35pf(x, X)
    assert len(hf) == len(of) and np.array_equal(hp, op, equal_nan=True) and np.array_equal(hf, of, equal_nan=True)


def test_ransac_pnpf_loop_of_the_product_takes_the_oracles_decisions():
    """pl_focal.h focal_lo_ransac - the loop the device driver runs - over a serial evaluation of the device functions
    (generator, scorer, k_lm_cam's algorithm): every decision and the returned model equal the oracle's ransac_pnpf"""
    for seed in range(8):
        n = [800, 300, 2000, 60][seed % 4]
        d = synth.absolute_pose_scene(n, [0.3, 0.5, 0.6][seed % 3], 8200 + seed, noise_px=0.5)
        x = _centered(d)
        op, of, om, ost = O.ransac_pnpf(x, d["p3d"], {"max_error": 4.0, "ransac": {"seed": seed}})
        hp, hf, hm, hst = HM.ransac_pnpf(x, d["p3d"], max_error=4.0, seed=seed)
        for k in ("iterations", "refinements", "num_inliers", "hypotheses", "model_score"):
            assert hst[k] == ost[k], (seed, k, hst[k], ost[k])
        assert np.array_equal(hm, om) and np.array_equal(hp, op) and hf == of, seed
        assert hst["iterations_evaluated"] >= hst["iterations"]
    # early stop inside a batch, a small iteration budget and fewer points than a sample
    d = synth.absolute_pose_scene(500, 0.1, 8300, noise_px=0.3)
    x = _centered(d)
    for ro in ({"min_iterations": 10, "max_iterations": 5000, "seed": 5}, {"min_iterations": 0, "max_iterations": 37, "seed": 6},
               {"min_iterations": 300, "max_iterations": 300, "seed": 7}):
        op, of, om, ost = O.ransac_pnpf(x, d["p3d"], {"max_error": 3.0, "ransac": ro})
        hp, hf, hm, hst = HM.ransac_pnpf(x, d["p3d"], max_error=3.0, seed=ro["seed"], max_iterations=ro["max_iterations"],
                                         min_iterations=ro["min_iterations"])
        assert hst["iterations"] == ost["iterations"] and hst["refinements"] == ost["refinements"], (ro, hst, ost)
        assert np.array_equal(hm, om) and np.array_equal(hp, op) and hf == of, ro
    op, of, om, ost = O.ransac_pnpf(x[:3], d["p3d"][:3], {"max_error": 3.0})
    hp, hf, hm, hst = HM.ransac_pnpf(x[:3], d["p3d"][:3], max_error=3.0)
    assert hst["iterations"] == ost["iterations"] == 0 and np.array_equal(hp, op) and hf == of == 1.0 and np.array_equal(hm, om)


# ---- the shared-focal relative pose estimator: pl_solver_6ptf.h, pl_sfocal.h and pl_focal.h's loop template ----
def _six_bearings(rng):
    from scipy.spatial.transform import Rotation
    f = rng.uniform(300, 3000)
    X = rng.uniform(-1, 1, (6, 3)) * [2, 2, 1] + [0, 0, 5]
    R = Rotation.from_rotvec(rng.normal(size=3) * 0.2).as_matrix()
    t = rng.normal(size=3)
    X2 = X @ R.T + t / np.linalg.norm(t)
    s = rng.uniform(400, 2500)
    b1 = np.c_[f * X[:, :2] / X[:, 2:] / s, np.ones(6)]
    b2 = np.c_[f * X2[:, :2] / X2[:, 2:] / s, np.ones(6)]
    return b1 / np.linalg.norm(b1, axis=1)[:, None], b2 / np.linalg.norm(b2, axis=1)[:, None]


def test_six_point_shared_focal_device_header_bit_exact():
    """pl_solver_6ptf.h (the serial statement of the solver whose steps the kernels of sfocal.hip distribute over lanes) against the
    oracle's restatement of the reference's template solver (solvers_focal.cc; both with the correctly rounded cube): poses, focal
    lengths and their order, bit for bit"""
    rng = np.random.default_rng(3)
    total = 0
    for k in range(300):
        b1, b2 = _six_bearings(rng)
        po, fo = O.relpose_6pt_shared_focal(b1, b2)
        ph, fh = HM.relpose_6pt_shared_focal(b1, b2, stride=1 + k % 3)
        assert po.shape == ph.shape and np.array_equal(po, ph) and np.array_equal(fo, fh), k
        total += len(fo)
    assert total > 300
    # degenerate input: all six correspondences equal
    b = np.tile(np.array([[0.1, 0.2, 1.0]]) / np.linalg.norm([0.1, 0.2, 1.0]), (6, 1))
    po, fo = O.relpose_6pt_shared_focal(b, b)
    ph, fh = HM.relpose_6pt_shared_focal(b, b)
    assert po.shape == ph.shape and np.array_equal(po, ph, equal_nan=True)


@pytest.mark.parametrize("seed", range(3))
def test_shared_focal_refiner_of_the_product_bit_exact(seed):
    """pl_sfocal.h's residual / Jacobian row / step under pl_refine.h's LM control, summed in k_sfocal_lm's order, against the
    oracle's refine_shared_focal_relpose (itself bit-exact with the reference's sources) - every loss"""
    d = synth.relative_pose_scene(300, 0.2, 100 + seed, focal=900.0)
    f, cx, cy = d["camera1"]["params"]
    a, b = (d["x1"] - [cx, cy]) / 700.0, (d["x2"] - [cx, cy]) / 700.0
    q = np.r_[d["q_gt"], d["t_gt"]] + 0.01 * np.random.default_rng(seed).normal(size=7)
    q[:4] /= np.linalg.norm(q[:4])
    for loss in range(6):
        po, fo, so = O.refine_shared_focal_relpose(a, b, q, 1.1 * f / 700, {"loss_type": loss, "loss_scale": 0.003, "max_iterations": 30})
        ph, fh, its, costs, skipped = HM.sfocal_lm(a, b, q, 1.1 * f / 700, HM.lm_options(30, loss, 0.003))
        assert not skipped and np.array_equal(po, ph) and fo == fh and its == so.iterations
        assert costs[0] == so.initial_cost and costs[1] == so.cost


def test_ransac_shared_focal_loop_of_the_product_takes_the_oracles_decisions():
    """pl_focal.h's loop template with SharedFocalTraits over the device functions evaluated serially: every decision, the score
    and the returned model equal the oracle's ransac_shared_focal_relpose"""
    for seed, outl, n in [(0, 0.3, 1200), (1, 0.5, 1200), (2, 0.2, 400), (3, 0.4, 2500), (4, 0.3, 40)]:
        d = synth.relative_pose_scene(n, outl, 8400 + seed)
        f, cx, cy = d["camera1"]["params"]
        a, b = (d["x1"] - [cx, cy]) / 500.0, (d["x2"] - [cx, cy]) / 500.0
        ro = {"seed": seed, "max_iterations": 3000} if seed != 2 else {"seed": 2, "min_iterations": 20, "success_prob": 0.95}
        po, fo, mo, so = O.ransac_shared_focal_relpose(a, b, {"max_error": 2.0 / 500, "ransac": ro})
        ph, fh, mh, sh = HM.ransac_shared_focal(a, b, max_error=2.0 / 500, seed=seed, max_iterations=ro.get("max_iterations", 100000),
                                                min_iterations=ro.get("min_iterations", 1000), success_prob=ro.get("success_prob", 0.9999))
        assert np.array_equal(po, ph) and fo == fh and np.array_equal(mo, mh), seed
        for k in ("iterations", "refinements", "num_inliers", "model_score"):
            assert so[k] == sh[k], (seed, k)
    # an initial model (score_initial_model)
    d = synth.relative_pose_scene(500, 0.3, 8500)
    f, cx, cy = d["camera1"]["params"]
    a, b = (d["x1"] - [cx, cy]) / 500.0, (d["x2"] - [cx, cy]) / 500.0
    init = np.r_[d["q_gt"], d["t_gt"]]
    po, fo, mo, so = O.ransac_shared_focal_relpose(a, b, {"max_error": 2.0 / 500, "ransac": {"seed": 9, "max_iterations": 1500, "score_initial_model": True}},
                                                   init_pose=init, init_focal=1.1 * f / 500)
    ph, fh, mh, sh = HM.ransac_shared_focal(a, b, max_error=2.0 / 500, seed=9, max_iterations=1500, init_pose=init, init_focal=1.1 * f / 500)
    assert np.array_equal(po, ph) and fo == fh and np.array_equal(mo, mh)
    assert (so["iterations"], so["refinements"], so["model_score"]) == (sh["iterations"], sh["refinements"], sh["model_score"])


def test_six_point_shared_focal_degenerate_and_hostile_inputs_terminate_and_agree():
    """planar scene, pure rotation, identical views, repeated / collinear correspondences, zero / NaN / huge / tiny bearings and
    random garbage: the device header and the oracle return the same models (usually none) - and RETURN (the eigenvalue
    iteration gives up after 60 sweeps, the balancing is not entered with a non-finite matrix: a hang on the device is a lost GPU)"""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)

    def unit(b):
        return b / np.linalg.norm(b, axis=1)[:, None]

    R = Rotation.from_rotvec([0.1, -0.2, 0.05]).as_matrix()
    X = np.c_[rng.uniform(-1, 1, (6, 2)), np.full(6, 4.0)]
    X2 = X @ R.T + [0.3, 0.1, 0.05]
    planar = (unit(np.c_[X[:, :2] / X[:, 2:], np.ones(6)]), unit(np.c_[X2[:, :2] / X2[:, 2:], np.ones(6)]))
    Y = rng.uniform(-1, 1, (6, 3)) + [0, 0, 4]
    Y2 = Y @ R.T
    rep1, rep2 = planar[0].copy(), planar[1].copy()
    rep1[5], rep2[5] = rep1[0], rep2[0]
    line = np.c_[np.linspace(-0.5, 0.5, 6), np.zeros(6), np.ones(6)]
    h = unit(rng.normal(size=(6, 3)))
    cases = [planar, (unit(np.c_[Y[:, :2] / Y[:, 2:], np.ones(6)]), unit(np.c_[Y2[:, :2] / Y2[:, 2:], np.ones(6)])), (planar[0], planar[0]),
             (rep1, rep2), (unit(line), unit(line + [0.01, 0, 0])), (np.zeros((6, 3)), np.zeros((6, 3))),
             (np.full((6, 3), np.nan), np.full((6, 3), np.nan)), (h * 1e200, h * 1e200), (h * 1e-200, unit(rng.normal(size=(6, 3))) * 1e-200),
             (np.tile([[0, 0, 1.0]], (6, 1)), np.tile([[0, 0, 1.0]], (6, 1)))]
    for k in range(400):  # garbage of every magnitude, some entries infinite or NaN
        a = rng.normal(size=(6, 3)) * 10.0 ** rng.integers(-300, 300, size=(6, 3))
        b = rng.normal(size=(6, 3)) * 10.0 ** rng.integers(-30, 30, size=(6, 3))
        if k % 7 == 0:
            a[rng.integers(6), rng.integers(3)] = [np.inf, -np.inf, np.nan][k % 3]
        cases.append((a, b))
    with np.errstate(all="ignore"):
        for i, (b1, b2) in enumerate(cases):
            po, fo = O.relpose_6pt_shared_focal(b1, b2)
            ph, fh = HM.relpose_6pt_shared_focal(b1, b2, stride=1 + i % 2)
            assert po.shape == ph.shape and np.array_equal(po, ph, equal_nan=True) and np.array_equal(fo, fh, equal_nan=True), i


def test_p35pf_hostile_inputs_terminate_and_agree():
    """garbage of every magnitude, infinite / NaN coordinates, four equal 3-D points, image points at the origin: the device
    header and the oracle return the same (usually no) solutions - and return"""
    rng = np.random.default_rng(9)
    with np.errstate(all="ignore"):
        for k in range(600):
            x = rng.normal(size=(4, 2)) * 10.0 ** rng.integers(-200, 200, size=(4, 2))
            X = rng.normal(size=(4, 3)) * 10.0 ** rng.integers(-30, 30, size=(4, 3))
            if k % 5 == 0:
                X[rng.integers(4), rng.integers(3)] = [np.inf, -np.inf, np.nan][k % 3]
            if k % 9 == 0:
                X[:] = X[0]
            if k % 11 == 0:
                x[:] = 0
            po, fo = O.p35pf(x, X)
            ph, fh = HM.p35pf(x, X, stride=1 + k % 3)
            assert po.shape == ph.shape and np.array_equal(po, ph, equal_nan=True) and np.array_equal(fo, fh, equal_nan=True), k


def test_focal_loops_with_prosac_sampling_take_the_oracles_decisions():
    """progressive_sampling (sampling.cc:85-136) in the two focal-length estimators: the reference constructs their sampler from
    opt.ransac like every other estimator's (absolute_pose.h:80, relative_pose.h:155).  The product's loop template draws the PROSAC
    samples on the host (pl_sampler.h ProsacSampler) and hands them to the generators explicitly; decisions and models equal the
    oracle's, including the cross-over to uniform sampling after max_prosac_iterations."""
    HM.set_prosac(True, 100000)
    try:
        for seed in range(4):
            n = [600, 250, 1500, 50][seed]
            d = synth.absolute_pose_scene(n, [0.3, 0.5, 0.4, 0.2][seed], 8600 + seed, noise_px=0.5)
            x = _centered(d)
            ro = {"seed": seed, "progressive_sampling": True}
            op, of, om, ost = O.ransac_pnpf(x, d["p3d"], {"max_error": 4.0, "ransac": ro})
            hp, hf, hm, hst = HM.ransac_pnpf(x, d["p3d"], max_error=4.0, seed=seed)
            for k in ("iterations", "refinements", "num_inliers", "hypotheses", "model_score"):
                assert hst[k] == ost[k], (seed, k, hst[k], ost[k])
            assert np.array_equal(hm, om) and np.array_equal(hp, op) and hf == of, seed
        for seed, n in [(0, 800), (1, 300)]:
            d = synth.relative_pose_scene(n, 0.3, 8700 + seed)
            f, cx, cy = d["camera1"]["params"]
            a, b = (d["x1"] - [cx, cy]) / 500.0, (d["x2"] - [cx, cy]) / 500.0
            po, fo, mo, so = O.ransac_shared_focal_relpose(a, b, {"max_error": 2.0 / 500, "ransac": {"seed": seed, "max_iterations": 3000, "progressive_sampling": True}})
            ph, fh, mh, sh = HM.ransac_shared_focal(a, b, max_error=2.0 / 500, seed=seed, max_iterations=3000)
            assert np.array_equal(po, ph) and fo == fh and np.array_equal(mo, mh), seed
            for k in ("iterations", "refinements", "num_inliers", "model_score"):
                assert so[k] == sh[k], (seed, k)
        # the cross-over: uniform sampling after max_prosac_iterations
        HM.set_prosac(True, 400)
        d = synth.absolute_pose_scene(700, 0.5, 8650, noise_px=0.5)
        x = _centered(d)
        op, of, om, ost = O.ransac_pnpf(x, d["p3d"], {"max_error": 4.0, "ransac": {"seed": 3, "progressive_sampling": True, "max_prosac_iterations": 400}})
        hp, hf, hm, hst = HM.ransac_pnpf(x, d["p3d"], max_error=4.0, seed=3)
        assert hst["iterations"] == ost["iterations"] and hst["refinements"] == ost["refinements"] and hst["model_score"] == ost["model_score"]
        assert np.array_equal(hm, om) and np.array_equal(hp, op) and hf == of
    finally:
        HM.set_prosac(False)


def test_entry_svd_of_the_fundamental_refinement_bit_exact():
    """poselib_amd/csrc/pl_svd3.h (the product's host-side SVD in front of every fundamental-matrix refinement) against the
    oracle's svd3 = the routine the reference sources run through the Eigen stand-in: U, singular values and V bit for bit,
    on full-rank, rank-2 (the fundamental matrices of the path) and degenerate inputs; and the factors are an SVD."""
    import ctypes as C

    rs = np.random.RandomState(99)
    ol, hl = O.lib(), HM.lib()

    def run(lib, fn, A):
        U, s, V = np.zeros(9), np.zeros(3), np.zeros(9)
        getattr(lib, fn)(*(a.ctypes.data_as(C.c_void_p) for a in (A, U, s, V)))
        return U.reshape(3, 3), s, V.reshape(3, 3)

    cases = []
    for i in range(3000):
        A = rs.randn(3, 3) * 10.0 ** rs.uniform(-6, 6)
        if i % 3 == 1:  # rank 2 up to rounding, like every F of the path
            u, s, vt = np.linalg.svd(A)
            A = (u[:, :2] * s[:2]) @ vt[:2]
        elif i % 3 == 2:  # a normalised fundamental matrix [t]_x R
            t = rs.randn(3)
            q = rs.randn(4)
            A = np.cross(np.eye(3), t) @ synth.quat_to_rotmat(q / np.linalg.norm(q))
            A /= np.linalg.norm(A)
        cases.append(np.ascontiguousarray(A).ravel())
    cases += [np.zeros(9), np.eye(3).ravel(), np.diag([3.0, -2.0, 0.0]).ravel(), np.outer([1.0, 2, 3], [4.0, 5, 6]).ravel(),
              np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=np.float64).ravel(), np.full(9, 1e-310)]
    for A in cases:
        a, b = run(ol, "orc_svd3", A), run(hl, "hm_svd3", A)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        U, s, V = a
        scale = max(np.abs(A).max(), 1e-300)
        assert np.abs((U * s) @ V.T - A.reshape(3, 3)).max() <= 1e-13 * scale
        assert s[0] >= s[1] >= s[2] >= 0
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-13 and np.abs(V.T @ V - np.eye(3)).max() < 1e-13


def test_flat_root_isolation_finds_the_recursions_leaves_bit_for_bit():
    """sturm_isolate_flat (one Sturm evaluation per round, leaves ranked afterwards: what k_rel_roots runs since round 5)
    against the recursion-shaped loop (sturm.h:210-231 restated) and against the oracle: the same roots, bit for bit and
    in the same order - on the determinant polynomials of real 5-point samples, on polynomials with prescribed clusters
    (double roots, roots 1e-11 apart: the narrow-interval branch), huge and tiny leading coefficients, no real roots."""
    rs = np.random.RandomState(77)
    polys = []
    for _ in range(1500):  # products of real and complex-pair factors: 0 .. 10 real roots, clustered on purpose
        nr = 2 * rs.randint(0, 6)
        roots = list(rs.randn(nr) * 10.0 ** rs.uniform(-2, 2))
        if nr >= 2 and rs.rand() < 0.5:
            roots[1] = roots[0] + 10.0 ** rs.uniform(-13, -3)  # a close pair
        if nr >= 4 and rs.rand() < 0.3:
            roots[3] = roots[2]  # a double root
        p = np.poly1d([1.0])
        for r in roots:
            p *= np.poly1d([1.0, -r])
        for _ in range((10 - nr) // 2):
            a, b = rs.randn(2)
            p *= np.poly1d([1.0, -2 * a, a * a + b * b + 1e-3])
        c = p.coeffs[::-1] * 10.0 ** rs.uniform(-8, 8)  # ascending, any scale
        polys.append(np.ascontiguousarray(c, dtype=np.float64))
    for _ in range(500):
        polys.append(rs.randn(11) * 10.0 ** rs.uniform(-3, 3, 11))
    polys += [np.r_[np.zeros(10), 1.0], np.r_[1.0, np.zeros(9), 1.0], np.r_[-1.0, np.zeros(9), 1.0], np.r_[rs.randn(10), 0.0],
              np.r_[rs.randn(10), 1e-300], np.r_[rs.randn(10) * 1e200, 1.0]]
    differ = 0
    with_roots = 0
    for c in polys:
        a, b = HM.sturm10(c), HM.sturm10(c, flat=True)
        assert len(a) == len(b) and np.array_equal(a, b), (c, a, b)
        with_roots += len(a) > 0
        o = O.sturm_roots(c)
        differ += not (len(o) == len(a) and np.array_equal(np.asarray(o), a))
    assert with_roots > 1200
    assert differ == 0  # (the oracle has no slot limits: equal wherever neither list overflows - everywhere here)


# ---- the pieces of the two template solvers (round 6: pl_action_template.h, pl_sturm_n.h, pl_general_eigenvalues) on their own ----
def test_general_eigenvalues_of_the_device_header_against_lapack():
    """pl_general_eigenvalues<10> (Hessenberg + Francis QR: the routine behind the P3.5Pf action matrix, serial form; the wavefront
    form of pl_eigen_wave.h performs the same operations on every element) - the spectrum of random and of structured matrices to
    1e-9 of LAPACK's; the bit-level check is the solver test above (the oracle runs the same iteration)"""
    import ctypes as C

    rng = np.random.default_rng(21)
    mats = [rng.normal(size=(10, 10)) for _ in range(200)]
    mats += [np.diag(rng.normal(size=10)) + 1e-3 * rng.normal(size=(10, 10)) for _ in range(50)]
    comp = np.zeros((10, 10))
    comp[1:, :-1] = np.eye(9)
    comp[:, -1] = rng.normal(size=10)
    mats.append(comp)
    M = np.ascontiguousarray(np.stack(mats))
    wr, wi = np.zeros((len(mats), 10)), np.zeros((len(mats), 10))
    rc = HM.lib().hm_general_eigenvalues(C.c_int(10), M.ctypes.data_as(C.c_void_p), C.c_int(len(mats)), wr.ctypes.data_as(C.c_void_p),
                                         wi.ctypes.data_as(C.c_void_p))
    assert rc == 0
    for k, A in enumerate(mats):
        got = np.sort_complex(wr[k] + 1j * wi[k])
        want = np.sort_complex(np.linalg.eigvals(A))
        assert np.abs(got - want).max() < 1e-9 * max(1.0, np.abs(want).max()), k


def test_sturm_of_degree_15_and_danilevsky_of_the_device_headers():
    """sturm_n_roots<15> (pl_sturm_n.h) against the oracle's generic Sturm bisection bit for bit at the six-point solver's tolerance;
    danilevsky_charpoly<15> (pl_action_template.h) against numpy's characteristic polynomial"""
    import ctypes as C

    rng = np.random.default_rng(22)
    with_roots = 0
    for k in range(400):
        roots_true = rng.normal(size=15) * 10.0 ** rng.integers(-2, 2)
        c = np.poly(roots_true)[::-1].copy() if k % 2 else rng.normal(size=16)
        c = np.ascontiguousarray(c * rng.uniform(0.5, 2.0))
        out = np.zeros(16)
        n = HM.lib().hm_sturm15(c.ctypes.data_as(C.c_void_p), C.c_double(1e-12), out.ctypes.data_as(C.c_void_p))
        want = O.sturm_roots(c, tol=1e-12)
        assert n == len(want) and np.array_equal(out[:n], np.asarray(want)), (k, out[:n], want)
        with_roots += n > 0
    assert with_roots > 300
    for k in range(50):
        A = np.ascontiguousarray(rng.normal(size=(15, 15)))
        p = np.zeros(16)
        HM.lib().hm_charpoly15(A.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p))
        want = np.poly(A)[::-1]
        assert np.abs(p - want).max() < 1e-7 * np.abs(want).max(), k
