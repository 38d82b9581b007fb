"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Bit-exact for integer / index work (iterations, refinements, inlier counts, masks);
poses / matrices within 1e-6 (BASELINE.json: "pose within 1e-6 rotation / 1e-6 translation norm").
"""
import numpy as np
import pytest

import oracle_lib as O
from poselib_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-6


def bear(p):
    b = np.c_[p, np.ones(len(p))]
    return b / np.sqrt((b * b).sum(1))[:, None]


def rot(q):
    return synth.quat_to_rotmat(np.asarray(q))


def pose_close(got, ref7, tol=POSE_TOL):
    dr = np.linalg.norm(rot(got.q) - rot(ref7[:4]))
    dt = np.linalg.norm(got.t - ref7[4:])
    return dr < tol and dt < tol, (dr, dt)


def mat_close(A, B, tol=POSE_TOL):
    """sign-SENSITIVE: a drop-in returns the reference's F / H, not its negative"""
    A = A / np.linalg.norm(A)
    B = B / np.linalg.norm(B)
    d = np.linalg.norm(A - B)
    return d < tol, d


# ------------------------------------------------------------------------------------------ solvers
def test_p3p_batch_matches_oracle(gpu):
    d = synth.absolute_pose_scene(3000, 0.3, 11)
    un = O.unproject(d["camera"], d["p2d"])
    idx, _ = O.sampler_draw(5, 3000, 3, 4000)
    idx = idx.astype(np.int64)
    xb = np.stack([bear(un[s]) for s in idx])
    Xp = np.stack([d["p3d"][s] for s in idx])
    rec, cnt = gpu.solve_batch(gpu.KIND_ABS, xb, Xp)
    worst = 0.0
    for i in range(len(idx)):
        ref = O.p3p(xb[i], Xp[i])
        assert cnt[i] == len(ref), (i, cnt[i], len(ref))
        for m in range(len(ref)):
            worst = max(worst, np.abs(rec[i, m, :7] - ref[m]).max())
    print("p3p: max |pose diff| device vs oracle =", worst)
    # bit for bit: cbrt / acos / cos of the cubic are glibc's algorithms on the device (pl_libm.h), everything else is
    # IEEE add / mul / div / sqrt in the oracle's order
    assert worst == 0.0, worst


def test_single_solver_entry_points(gpu):
    d = synth.absolute_pose_scene(50, 0.0, 3, noise_px=0.0)
    un = O.unproject(d["camera"], d["p2d"])
    sols = gpu.p3p(bear(un[:3]), d["p3d"][:3])
    ref = O.p3p(bear(un[:3]), d["p3d"][:3])
    assert len(sols) == len(ref) and len(ref) >= 1
    for s, r in zip(sols, ref):
        ok, err = pose_close(s, r, 1e-9)
        assert ok, err
    assert any(pose_close(s, np.r_[d["q_gt"], d["t_gt"]], 1e-6)[0] for s in sols)

    dd = synth.relative_pose_scene(7, 0.0, 7, noise_px=0.0)
    a = O.unproject(dd["camera1"], dd["x1"])
    b = O.unproject(dd["camera2"], dd["x2"])
    sols = gpu.relpose_5pt(bear(a[:5]), bear(b[:5]))
    ref = O.relpose_5pt(bear(a[:5]), bear(b[:5]))
    assert len(sols) == len(ref) and len(ref) >= 1
    for s, r in zip(sols, ref):
        ok, err = pose_close(s, r, 1e-7)
        assert ok, err
    Es = gpu.essential_matrix_5pt(bear(a[:5]), bear(b[:5]))
    Er = O.essential_5pt(bear(a[:5]), bear(b[:5]))
    assert len(Es) == len(Er)
    for E, R in zip(Es, Er):
        assert np.abs(E - R).max() < 1e-7
    Fs = gpu.relpose_7pt(bear(a), bear(b))
    Fr = O.relpose_7pt(bear(a), bear(b))
    assert len(Fs) == len(Fr) and len(Fr) >= 1
    for F, R in zip(Fs, Fr):
        assert np.abs(F - R).max() < 1e-9

    dh = synth.homography_scene(4, 0.0, 5, noise_px=0.0)
    a, b = dh["x1"] / 1000.0, dh["x2"] / 1000.0
    Hs = gpu.homography_4pt(bear(a), bear(b))
    n, Hr = O.homography_4pt(bear(a), bear(b))
    assert len(Hs) == n
    if n:
        assert np.abs(Hs[0] - Hr).max() < 1e-12


@pytest.mark.parametrize("kind,name,K", [(1, "rel", 5), (2, "fund", 7), (3, "hom", 4)])
def test_two_view_solver_batches(gpu, kind, name, K):
    if name == "hom":
        d = synth.homography_scene(2000, 0.3, 21)
    else:
        d = synth.relative_pose_scene(2000, 0.3, 22)
    a, b = (d["x1"] - 500.0) / 1000.0, (d["x2"] - 500.0) / 1000.0
    idx, _ = O.sampler_draw(9, 2000, K, 1500)
    idx = idx.astype(np.int64)
    A = np.stack([bear(a[s]) for s in idx])
    B = np.stack([bear(b[s]) for s in idx])
    rec, cnt = gpu.solve_batch(kind, A, B)
    worst, mismatched, diffs = 0.0, 0, []
    for i in range(len(idx)):
        if name == "rel":
            ref = O.relpose_5pt(A[i], B[i])
            got = [rec[i, m, :7] for m in range(cnt[i])]
        elif name == "fund":
            ref = [F.reshape(9) for F in O.relpose_7pt(A[i], B[i])]
            got = [rec[i, m, 7:] for m in range(cnt[i])]
        else:
            n, H = O.homography_4pt(A[i], B[i])
            ref = [H.reshape(9)] if n else []
            got = [rec[i, m, 7:] for m in range(cnt[i])]
        if len(ref) != len(got):
            mismatched += 1
            continue
        for g, r in zip(got, ref):
            worst = max(worst, np.abs(g - r).max())
            diffs.append(np.abs(g - r).max())
    diffs = np.sort(np.array(diffs))
    p99 = diffs[int(0.99 * (len(diffs) - 1))] if len(diffs) else 0.0
    print(f"{name}: solution-count mismatches {mismatched}/{len(idx)}, max |diff| {worst}, median "
          f"{np.median(diffs) if len(diffs) else 0}, p99 {p99}")
    # The device solvers are the oracle's arithmetic operation for operation (tests/test_hostmath_vs_oracle.py checks the
    # same headers bit for bit on the host): -ffp-contract=off on both sides, IEEE division and square root.  The 5-point
    # and the homography solvers call nothing but sqrt from libm, so they must agree to the bit; the 7-point solver's
    # cubic goes through cbrt or acos / cos, which pl_libm.h restates from glibc (bit-identical to the host's libm).
    assert mismatched == 0
    assert worst == 0.0, worst

# ------------------------------------------------------------------------------------------ scoring / refinement
def test_score_and_refine_match_oracle(gpu):
    d = synth.absolute_pose_scene(5000, 0.7, 1001)
    un = O.unproject(d["camera"], d["p2d"])
    thr = 12.0 / 1000.0
    prob = gpu.Problem(gpu.KIND_ABS, un, d["p3d"])
    rs = np.random.RandomState(0)
    for k in range(6):
        q = d["q_gt"] + 0.02 * k * rs.randn(4)
        q /= np.linalg.norm(q)
        t = d["t_gt"] + 0.02 * k * rs.randn(3)
        sc, cnt = prob.score(gpu.CameraPose(q, t), thr)
        osc, ocnt = O.score("reproj", np.r_[q, t], un, d["p3d"], thr * thr)
        assert cnt == ocnt, (k, cnt, ocnt)
        assert abs(sc - osc) <= 1e-10 * abs(osc)
    q = d["q_gt"] + 0.01 * np.array([0.3, -0.2, 0.5, 0.1])
    q /= np.linalg.norm(q)
    t = d["t_gt"] + np.array([0.01, -0.02, 0.015])
    for loss, scale, mi in [("TRUNCATED", thr, 25), ("CAUCHY", 0.001, 100), ("HUBER", 0.002, 50)]:
        bo = {"loss_type": loss, "loss_scale": scale, "max_iterations": mi}
        ref, st = O.bundle_adjust(un, d["p3d"], {"model": "NULL", "params": []}, np.r_[q, t], bo)
        got, it = prob.refine(gpu.CameraPose(q, t), bo)
        ok, err = pose_close(got, ref, 1e-9)
        print("refine abs", loss, "iters", it, st.iterations, "err", err)
        assert ok, err
    prob.close()


def test_prefilter_never_drops_an_inlier(gpu):
    """The scoring kernel's conservative fp32 pre-filter may only skip points that are certainly outliers:
    inlier counts must equal the oracle's for good, bad and borderline models, also when the world frame is
    far from the origin (fp32 cancellation => the error bound must widen, not the result change)."""
    d = synth.absolute_pose_scene(5000, 0.7, 1001)
    un = O.unproject(d["camera"], d["p2d"])
    R = rot(d["q_gt"])
    rs = np.random.RandomState(5)
    idx, _ = O.sampler_draw(3, 5000, 3, 150)
    idx = idx.astype(np.int64)
    rec, cnt = gpu.solve_batch(gpu.KIND_ABS, np.stack([bear(un[s]) for s in idx]), np.stack([d["p3d"][s] for s in idx]))
    models = [rec[i, m, :7] for i in range(len(idx)) for m in range(cnt[i]) if np.isfinite(rec[i, m, :7]).all()]
    for k in range(60):  # from perfect to useless
        q = d["q_gt"] + 0.002 * k * rs.randn(4)
        q /= np.linalg.norm(q)
        models.append(np.r_[q, d["t_gt"] + 0.002 * k * rs.randn(3)])
    for shift, thr in [(0.0, 0.012), (0.0, 0.001), (0.0, 0.2), (1e5, 0.012), (1e7, 0.012)]:
        off = np.array([shift, -shift, 0.5 * shift])
        X = d["p3d"] + off  # same scene expressed in a shifted world frame
        prob = gpu.Problem(gpu.KIND_ABS, un, X)
        for mdl in models:
            t = mdl[4:] - rot(mdl[:4]) @ off
            sc, c = prob.score(gpu.CameraPose(mdl[:4], t), thr)
            osc, oc = O.score("reproj", np.r_[mdl[:4], t], un, X, thr * thr)
            assert c == oc, (shift, thr, c, oc)
            assert abs(sc - osc) <= 1e-9 * abs(osc)
        prob.close()


def test_streaming_filters_do_not_change_any_result(gpu):
    """The main-loop scorers (fp16 / MFMA filter for absolute pose, fp32 filters for the two-view scores) against the
    same runs with the filters switched off (POSELIB_AMD_NO_PREFILTER=1: every pair evaluated exactly) and with the
    fp32 filter instead of the MFMA one (POSELIB_AMD_NO_MFMA=1): iterations, refinements, inlier counts, masks AND
    the final MSAC score must be bit-identical - the filters only remove exact evaluations of proven non-inliers.
    Scenes include a world frame shifted by 1e5 (beyond fp16: the MFMA path must fall back to exact evaluation per
    point), worlds scaled by 1e-2 / 1e-4 (fp16 subnormal operands), tiny and huge thresholds.  The settings are read once per process, hence subprocesses."""
    import json
    import os
    import subprocess
    import sys

    code = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import poselib_amd as P
from poselib_amd import synth
out = []
def rec(info, model):
    out.append([info["iterations"], info["refinements"], info["num_inliers"], repr(info["model_score"]),
                int(np.packbits(np.array(info["inliers"], dtype=np.uint8)).sum()), [repr(float(v)) for v in np.ravel(model)]])
for seed, n, outl, err, shift, scale in ((1, 5000, 0.7, 12.0, 0.0, 1.0), (2, 3000, 0.4, 1.0, 0.0, 1.0), (3, 2000, 0.5, 200.0, 0.0, 1.0),
                                         (4, 4000, 0.6, 12.0, 1e5, 1.0), (5, 1500, 0.2, 0.05, 0.0, 1.0),
                                         (9, 3000, 0.5, 2.0, 0.0, 0.01), (10, 3000, 0.5, 12.0, 0.0, 1e-4)):
    d = synth.absolute_pose_scene(n, outl, 900 + seed)
    par = d["camera"]["params"]
    x = (d["p2d"] - par[-2:]) / par[0]
    X = d["p3d"] * scale + np.array([shift, -shift, 0.5 * shift])
    pose, info = P.ransac_pnp(x, X, {"max_error": err / par[0], "ransac": {"seed": seed, "max_iterations": 6000, "min_iterations": 6000}})
    rec(info, np.r_[pose.q, pose.t])
for seed, gen, fn in ((6, synth.relative_pose_scene, P.ransac_relpose), (7, synth.fundamental_scene, P.ransac_fundamental),
                      (8, synth.homography_scene, P.ransac_homography)):
    d = gen(4000, 0.5, 900 + seed)
    x1, x2 = (d["x1"] - 500.0) / 1000.0, (d["x2"] - 500.0) / 1000.0
    for err in (1e-3, 3e-2):
        m, info = fn(x1, x2, {"max_error": err, "ransac": {"seed": seed, "max_iterations": 3000, "min_iterations": 3000}})
        rec(info, np.r_[m.q, m.t] if hasattr(m, "q") else m)
print("RESULT " + json.dumps(out))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    results = {}
    for tag, extra in (("mfma", {}), ("fp32", {"POSELIB_AMD_NO_MFMA": "1"}), ("exact", {"POSELIB_AMD_NO_PREFILTER": "1"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code, root], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        results[tag] = json.loads(line[7:])
    assert results["mfma"] == results["exact"]
    assert results["fp32"] == results["exact"]
    assert all(row[2] > 100 for row in results["exact"][:3])  # the runs found their models


def test_host_fallbacks_of_the_bookkeeping_agree_with_the_device_path(gpu):
    """The sampler's orbit walk and the scan for improving hypotheses run on the device; both have host fall-backs
    (orbit window too small / too many redraws; record list overflow).  Forced through the diagnostic switches they
    must reproduce the device path on runs of several batches (draw positions of a later batch are relative to its
    start), and POSELIB_AMD_CHECK_POSITIONS=1 compares the two position tables batch by batch."""
    import json
    import os
    import subprocess
    import sys

    code = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import poselib_amd as P
from poselib_amd import synth
out = []
for gen, fn, n, seed in ((synth.fundamental_scene, P.estimate_fundamental, 2000, 3), (synth.homography_scene, P.estimate_homography, 40, 4),
                         (synth.fundamental_scene, P.estimate_fundamental, 30, 5)):
    d = gen(n, 0.4, 70 + seed)
    M, info = fn(d["x1"], d["x2"], {"ransac": {"seed": seed, "min_iterations": 2500}})
    out.append([info["iterations"], info["refinements"], info["num_inliers"], repr(info["model_score"]), [repr(float(v)) for v in np.ravel(M)]])
d = synth.absolute_pose_scene(25, 0.3, 99)
img, info = P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], {"ransac": {"seed": 6, "min_iterations": 5000}})
out.append([info["iterations"], info["refinements"], info["num_inliers"], repr(info["model_score"]), [repr(float(v)) for v in list(img.pose.q) + list(img.pose.t)]])
print("RESULT " + json.dumps(out))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    results = {}
    for tag, extra in (("device", {"POSELIB_AMD_CHECK_POSITIONS": "1"}), ("host_positions", {"POSELIB_AMD_HOST_POSITIONS": "1"}),
                       ("host_records", {"POSELIB_AMD_HOST_RECORDS": "1"}), ("host_both", {"POSELIB_AMD_HOST_BOOKKEEPING": "1"})):
        r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        results[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert all(row[0] > 2500 for row in results["device"])  # several batches each
    for tag in ("host_positions", "host_records", "host_both"):
        assert results[tag] == results["device"], tag


# ------------------------------------------------------------------------------------------ end to end
ABS_CASES = [(200, 0.5, 1000, 0), (200, 0.5, 1000, 7), (5000, 0.7, 1001, 0), (5000, 0.7, 1001, 3), (1500, 0.3, 77, 1)]


@pytest.mark.parametrize("n,outl,dseed,rseed", ABS_CASES)
def test_estimate_absolute_pose_parity(gpu, n, outl, dseed, rseed):
    d = synth.absolute_pose_scene(n, outl, dseed)
    opt = {"ransac": {"seed": rseed}}
    img, info = gpu.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
    pose, mask, st = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
    print("abs", n, outl, "iters", info["iterations"], st["iterations"], "refinements", info["refinements"],
          st["refinements"], "inliers", info["num_inliers"], st["num_inliers"], "evaluated",
          info["iterations_evaluated"], "hyp", info["hypotheses"])
    assert info["iterations"] == st["iterations"]
    assert info["refinements"] == st["refinements"]
    assert info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    assert abs(info["model_score"] - st["model_score"]) <= 1e-9 * abs(st["model_score"])
    ok, err = pose_close(img.pose, pose)
    assert ok, err
    ok, err = pose_close(img.pose, np.r_[d["q_gt"], d["t_gt"]], 0.05)
    assert ok, err


@pytest.mark.parametrize("n,outl,dseed,rseed", [(5000, 0.5, 1002, 0), (1000, 0.3, 55, 2)])
def test_estimate_relative_pose_parity(gpu, n, outl, dseed, rseed):
    d = synth.relative_pose_scene(n, outl, dseed)
    opt = {"ransac": {"seed": rseed}}
    pose_g, info = gpu.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
    pose, mask, st = O.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
    print("rel", n, "iters", info["iterations"], st["iterations"], "refinements", info["refinements"],
          st["refinements"], "inliers", info["num_inliers"], st["num_inliers"], "hyp", info["hypotheses"])
    assert info["iterations"] == st["iterations"]
    assert info["refinements"] == st["refinements"]
    assert info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    ok, err = pose_close(pose_g, pose)
    assert ok, err


@pytest.mark.parametrize("n,outl,dseed,rseed", [(10000, 0.5, 1003, 0), (2000, 0.3, 31, 4)])
def test_estimate_homography_parity(gpu, n, outl, dseed, rseed):
    d = synth.homography_scene(n, outl, dseed, noise_px=0.3)
    opt = {"ransac": {"seed": rseed}}
    H, info = gpu.estimate_homography(d["x1"], d["x2"], opt)
    Hr, mask, st = O.estimate_homography(d["x1"], d["x2"], opt)
    print("hom", n, "iters", info["iterations"], st["iterations"], "refinements", info["refinements"],
          st["refinements"], "inliers", info["num_inliers"], st["num_inliers"])
    assert info["iterations"] == st["iterations"]
    assert info["refinements"] == st["refinements"]
    assert info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    ok, err = mat_close(H, Hr)
    assert ok, err


def test_homography_decision_between_h_and_minus_h(gpu):
    """Round 6 (item 1186 of tests/parity_soak_estimate_batch.py 4000 11): two local optimisations from minimal models of opposite
    sign end in the same optimum, as H and -H, with scores that agree to the last bits; `score < best` then hangs on the bits of the
    refined models.  With the default tree-order sums the device returned -1 x the reference's matrix (everything else identical);
    the driver now repeats the two refinements in the reference's order when such a decision comes up (RansacRun::resolve_sign_tie).
    Single call and batch call, sign-sensitive."""
    n, outl = 1616, 0.12012363631497766
    d = synth.homography_scene(n, outl, 60000 + 1186)
    order = np.argsort(~d["inlier_gt"], kind="stable")
    x1, x2 = np.asarray(d["x1"])[order], np.asarray(d["x2"])[order]
    opt = {"ransac": {"seed": 579604402, "progressive_sampling": True}}
    Hr, mask, st = O.estimate_homography(x1, x2, opt)
    H, info = gpu.estimate_homography(x1, x2, opt)
    (res,) = gpu.estimate_batch([("hom", x1, x2, opt)], max_in_flight=1)
    for got, inf in ((H, info), res):
        assert inf["iterations"] == st["iterations"] and inf["refinements"] == st["refinements"]
        assert inf["num_inliers"] == st["num_inliers"] and (np.array(inf["inliers"]) == mask).all()
        ok, err = mat_close(got, Hr)
        assert ok, err


@pytest.mark.parametrize("n,outl,dseed,rseed", [(10000, 0.5, 1004, 0), (2000, 0.3, 32, 5)])
def test_estimate_fundamental_parity(gpu, n, outl, dseed, rseed):
    d = synth.fundamental_scene(n, outl, dseed)
    opt = {"ransac": {"seed": rseed}}
    F, info = gpu.estimate_fundamental(d["x1"], d["x2"], opt)
    Fr, mask, st = O.estimate_fundamental(d["x1"], d["x2"], opt)
    print("fund", n, "iters", info["iterations"], st["iterations"], "refinements", info["refinements"],
          st["refinements"], "inliers", info["num_inliers"], st["num_inliers"])
    assert info["iterations"] == st["iterations"]
    assert info["refinements"] == st["refinements"]
    assert info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    ok, err = mat_close(F, Fr)
    assert ok, err


@pytest.mark.parametrize("max_prosac", [100000, 300])
def test_prosac_sampling_parity(gpu, max_prosac):
    """progressive sampling (sampling.cc:85-136): correspondences sorted by quality, samples from a growing
    prefix; with max_prosac_iterations = 300 the run crosses over to uniform sampling half way"""
    d = synth.absolute_pose_scene(2000, 0.6, 91)
    order = np.argsort(~d["inlier_gt"], kind="stable")  # "best" matches first
    p2d, p3d = d["p2d"][order], d["p3d"][order]
    opt = {"ransac": {"seed": 4, "progressive_sampling": True, "max_prosac_iterations": max_prosac}}
    img, info = gpu.estimate_absolute_pose(p2d, p3d, d["camera"], opt)
    pose, mask, st = O.estimate_absolute_pose(p2d, p3d, d["camera"], opt)
    assert info["iterations"] == st["iterations"] and info["refinements"] == st["refinements"]
    assert info["num_inliers"] == st["num_inliers"] and (np.array(info["inliers"]) == mask).all()
    ok, err = pose_close(img.pose, pose)
    assert ok, err
    h = synth.homography_scene(3000, 0.5, 92)
    order = np.argsort(~h["inlier_gt"], kind="stable")
    x1, x2 = h["x1"][order], h["x2"][order]
    Hg, info = gpu.estimate_homography(x1, x2, opt)
    Ho, mask, st = O.estimate_homography(x1, x2, opt)
    assert info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    ok, err = mat_close(Hg, Ho)
    assert ok, err


def test_estimate_batch_matches_single_calls(gpu):
    """pl_estimate_batch (array of problem descriptors, BASELINE config 4) = the single-problem entry points"""
    probs, singles = [], []
    for i in range(12):
        n = 300 + 137 * i
        opt = {"ransac": {"seed": i}}
        if i % 3 == 0:
            d = synth.absolute_pose_scene(n, 0.4, 600 + i)
            probs.append(("abs", d["p2d"], d["p3d"], d["camera"], opt))
            img, info = gpu.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
            singles.append((np.r_[img.pose.q, img.pose.t], info))
        elif i % 3 == 1:
            d = synth.relative_pose_scene(n, 0.4, 600 + i)
            probs.append(("rel", d["x1"], d["x2"], d["camera1"], d["camera2"], opt))
            pose, info = gpu.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
            singles.append((np.r_[pose.q, pose.t], info))
        else:
            d = synth.homography_scene(n, 0.4, 600 + i)
            probs.append(("hom", d["x1"], d["x2"], opt))
            H, info = gpu.estimate_homography(d["x1"], d["x2"], opt)
            singles.append((H.reshape(-1), info))
    for in_flight in (1, 5):
        res = gpu.estimate_batch(probs, max_in_flight=in_flight)
        assert len(res) == len(probs)
        for (model, info), (ref_model, ref_info), pr in zip(res, singles, probs):
            flat = (np.r_[model.pose.q, model.pose.t] if pr[0] == "abs" else
                    np.r_[model.q, model.t] if pr[0] == "rel" else model.reshape(-1))
            assert np.array_equal(flat, ref_model)  # same kernels, same order of operations: bit-identical
            for k in ("iterations", "refinements", "num_inliers", "model_score"):
                assert info[k] == ref_info[k]
            assert info["inliers"] == ref_info["inliers"]


@pytest.mark.parametrize("mode", ["1", "0"])
def test_multi_workgroup_and_single_workgroup_lm(gpu, mode):
    """POSELIB_AMD_LATENCY_MODE=1 sends the LO of large homography / fundamental problems through k_lm2 (one task
    spread over several workgroups, one launch per LM iteration); the default keeps every task on one workgroup (k_lm).
    Both have to reproduce the oracle; run in subprocesses because the setting is read once per process."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import poselib_amd as P, oracle_lib as O
from poselib_amd import synth
for gen, fn, ofn, seed in ((synth.homography_scene, P.estimate_homography, O.estimate_homography, 1003),
                           (synth.fundamental_scene, P.estimate_fundamental, O.estimate_fundamental, 1004)):
    d = gen(10000, 0.5, seed)
    opt = {"ransac": {"seed": 0}}
    M, info = fn(d["x1"], d["x2"], opt)
    Mo, mask, st = ofn(d["x1"], d["x2"], opt)
    assert info["iterations"] == st["iterations"] and info["refinements"] == st["refinements"], (info, st)
    assert info["num_inliers"] == st["num_inliers"] and (np.array(info["inliers"]) == mask).all()
    A, B = M / np.linalg.norm(M), Mo / np.linalg.norm(Mo)
    assert np.linalg.norm(A - B) < 1e-6  # sign included
print("latency mode ok")
"""
    env = dict(os.environ, POSELIB_AMD_LATENCY_MODE=mode)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, root], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "latency mode ok" in out.stdout, out.stdout + out.stderr


def test_edge_cases(gpu):
    d = synth.absolute_pose_scene(200, 0.5, 1000)
    # fewer points than the sample size: default stats, identity pose, mask computed for it (ransac_impl.h:161-163)
    img, info = gpu.estimate_absolute_pose(d["p2d"][:2], d["p3d"][:2], d["camera"], {})
    pose, mask, st = O.estimate_absolute_pose(d["p2d"][:2], d["p3d"][:2], d["camera"], {})
    assert info["iterations"] == 0 and st["iterations"] == 0
    assert (np.array(info["inliers"]) == mask).all()
    # warm start (score_initial_model): supplied pose is scored and refined before the loop
    init = gpu.CameraPose(d["q_gt"], d["t_gt"])
    img, info = gpu.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], {}, initial_pose=init)
    o = O.robust_opt({"ransac": {"score_initial_model": True}}, 12.0)
    pose, mask, st = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], {"ransac": {"score_initial_model": True}},
                                              init_pose=np.r_[d["q_gt"], d["t_gt"]])
    assert info["iterations"] == st["iterations"] and info["refinements"] == st["refinements"]
    assert (np.array(info["inliers"]) == mask).all()
    ok, err = pose_close(img.pose, pose)
    assert ok, err
    # max_iterations smaller than min_iterations; tiny budgets
    for mx, mn in [(50, 1000), (1, 0), (300, 100)]:
        opt = {"ransac": {"max_iterations": mx, "min_iterations": mn, "seed": 9}}
        img, info = gpu.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
        pose, mask, st = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
        assert info["iterations"] == st["iterations"], (mx, mn, info["iterations"], st["iterations"])
        assert (np.array(info["inliers"]) == mask).all()


@pytest.mark.parametrize("n", [3, 4, 5, 7, 63, 64, 65, 319, 320, 321, 383, 384, 385, 640, 1281])
def test_point_counts_around_the_chunk_boundaries(gpu, n):
    """ragged inputs: the streaming scorer holds 64 * P correspondences per wavefront (P = 1..6, chosen from N), the
    small scorers 256 * P - every size class, exact multiples and off-by-ones, for all four estimators"""
    opt = {"ransac": {"seed": n, "max_iterations": 400, "min_iterations": 100}}
    d = synth.absolute_pose_scene(max(n, 3), 0.3, 40 + n)
    img, info = gpu.estimate_absolute_pose(d["p2d"][:n], d["p3d"][:n], d["camera"], opt)
    pose, mask, st = O.estimate_absolute_pose(d["p2d"][:n], d["p3d"][:n], d["camera"], opt)
    assert info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    h = synth.homography_scene(max(n, 4), 0.3, 41 + n)
    H, info = gpu.estimate_homography(h["x1"][:n], h["x2"][:n], opt)
    Ho, mask, st = O.estimate_homography(h["x1"][:n], h["x2"][:n], opt)
    assert info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    r = synth.relative_pose_scene(max(n, 7), 0.3, 42 + n)
    pg, info = gpu.estimate_relative_pose(r["x1"][:n], r["x2"][:n], r["camera1"], r["camera2"], opt)
    po, mask, st = O.estimate_relative_pose(r["x1"][:n], r["x2"][:n], r["camera1"], r["camera2"], opt)
    assert info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    F, info = gpu.estimate_fundamental(r["x1"][:n], r["x2"][:n], opt)
    Fo, mask, st = O.estimate_fundamental(r["x1"][:n], r["x2"][:n], opt)
    assert info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()


@pytest.mark.parametrize("n", [8, 11, 24, 60, 200])
def test_sampler_with_frequent_redraws(gpu, n):
    """Few correspondences: most iterations redraw a duplicate index (sampling.cc:46-61), so the device sampler's
    orbit has a flag at almost every position (bitmap -> successor links -> segments; beyond 12288 flags or 4096
    segments per batch it must fall back to the host walk) - long fixed-length runs must still follow the
    reference's draw stream exactly.  With so few correspondences the same sample recurs in permuted order and its
    models tie to the last bits of the MSAC score: `refinements` only agrees because every score
    a decision is taken on is summed in the reference's order (k_score_seq) and because the tied models themselves are
    bit-identical to the oracle's - the device's cbrt / acos / cos are glibc's algorithms (pl_libm.h)."""
    refs = lambda a, b: a == b  # noqa: E731
    refs_p3p = refs
    opt = {"ransac": {"seed": 5 + n, "max_iterations": 30000, "min_iterations": 30000}}
    r = synth.relative_pose_scene(max(n, 8), 0.25, 70 + n)
    F, info = gpu.estimate_fundamental(r["x1"][:n], r["x2"][:n], opt)
    Fo, mask, st = O.estimate_fundamental(r["x1"][:n], r["x2"][:n], opt)
    assert info["iterations"] == st["iterations"] == 30000 and refs(info["refinements"], st["refinements"])
    assert abs(info["model_score"] - st["model_score"]) <= 1e-9 * st["model_score"]
    assert info["num_inliers"] == st["num_inliers"] and (np.array(info["inliers"]) == mask).all()
    d = synth.absolute_pose_scene(max(n, 8), 0.25, 71 + n)
    img, info = gpu.estimate_absolute_pose(d["p2d"][:n], d["p3d"][:n], d["camera"], opt)
    pose, mask, st = O.estimate_absolute_pose(d["p2d"][:n], d["p3d"][:n], d["camera"], opt)
    assert info["iterations"] == st["iterations"] == 30000 and refs_p3p(info["refinements"], st["refinements"])
    assert info["num_inliers"] == st["num_inliers"] and (np.array(info["inliers"]) == mask).all()
    h = synth.homography_scene(max(n, 8), 0.25, 72 + n)
    H, info = gpu.estimate_homography(h["x1"][:n], h["x2"][:n], opt)
    Ho, mask, st = O.estimate_homography(h["x1"][:n], h["x2"][:n], opt)
    assert info["iterations"] == st["iterations"] == 30000 and refs(info["refinements"], st["refinements"])
    assert info["num_inliers"] == st["num_inliers"] and (np.array(info["inliers"]) == mask).all()


def _run_sharded_threads(kind, a, b, opt, world):
    """`world` host threads, each with its own Problem (own HIP stream / scratch) on the one GPU of the box, exchange
    through sharding.thread_allgather: the within-problem sharding of pl_ransac_run_sharded, rank by rank."""
    import threading

    import poselib_amd as P
    from poselib_amd import sharding

    ag_for = sharding.thread_allgather(world)
    out = [None] * world
    err = []

    def run(rank):
        try:
            prob = P.Problem(kind, a, b)
            out[rank] = prob.run_sharded(opt, rank, world, ag_for(rank))
        except Exception as e:  # noqa: BLE001
            err.append(e)
            raise

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not err, err
    return out


SHARD_FIELDS = ("iterations", "refinements", "num_inliers", "model_score", "hypotheses", "inlier_ratio")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_one_problem_sharded_over_ranks_equals_the_single_device_run(gpu, world):
    """iteration ranges per rank + one all-gather of improving hypotheses per batch.  Every rank must
    return exactly the single-device result: model bits, mask, iterations (early stop in the middle of a batch
    included), refinements, hypothesis count."""
    import poselib_amd as P

    cases = []
    d = synth.absolute_pose_scene(3000, 0.6, 811)
    par = d["camera"]["params"]
    cases.append((0, (d["p2d"] - par[-2:]) / par[0], d["p3d"], 12.0 / par[0]))
    r = synth.relative_pose_scene(2000, 0.4, 812)
    cases.append((1, (r["x1"] - 500.0) / 1000.0, (r["x2"] - 500.0) / 1000.0, 1e-3))
    f = synth.fundamental_scene(2500, 0.4, 813)
    cases.append((2, (f["x1"] - 500.0) / 1000.0, (f["x2"] - 500.0) / 1000.0, 1e-3))
    h = synth.homography_scene(2500, 0.4, 814)
    cases.append((3, (h["x1"] - 500.0) / 1000.0, (h["x2"] - 500.0) / 1000.0, 1e-3))
    for kind, a, b, err in cases:
        for ransac in ({"seed": 3}, {"seed": 4, "max_iterations": 20000, "min_iterations": 20000},
                       {"seed": 5, "progressive_sampling": True, "max_prosac_iterations": 500}):
            opt = {"max_error": err, "ransac": ransac}
            m0, i0 = P.Problem(kind, a, b).run(opt)
            flat0 = np.r_[m0.q, m0.t] if hasattr(m0, "q") else np.ravel(m0)
            for m, info in _run_sharded_threads(kind, a, b, opt, world):
                flat = np.r_[m.q, m.t] if hasattr(m, "q") else np.ravel(m)
                assert (flat == flat0).all(), (kind, ransac)
                assert all(info[k] == i0[k] for k in SHARD_FIELDS), (kind, ransac, {k: (info[k], i0[k]) for k in SHARD_FIELDS})
                assert (np.array(info["inliers"]) == np.array(i0["inliers"])).all()


def test_sharded_run_over_torch_distributed_gloo(gpu):
    """the same through torch.distributed (two processes sharing the box's GPU, gloo all-gather on CPU tensors - on a
    node with one GPU per rank the callback is sharding.dist_allgather(device=...) over RCCL)"""
    import json
    import os
    import subprocess
    import sys

    code = r"""
import os, sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import poselib_amd as P
from poselib_amd import synth, sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
d = synth.absolute_pose_scene(2000, 0.5, 815)
par = d["camera"]["params"]
x, X = (d["p2d"] - par[-2:]) / par[0], d["p3d"]
opt = {"max_error": 12.0 / par[0], "ransac": {"seed": 9, "max_iterations": 5000, "min_iterations": 5000}}
prob = P.Problem(0, x, X)
pose, info = prob.run_sharded(opt, rank, world, sharding.dist_allgather())
single_pose, single = prob.run(opt)
ok = (list(pose.q) + list(pose.t) == list(single_pose.q) + list(single_pose.t)) and all(
    info[k] == single[k] for k in ("iterations", "refinements", "num_inliers", "model_score", "hypotheses")) and info["inliers"] == single["inliers"]
print("RESULT " + json.dumps([rank, bool(ok), info["iterations"], info["num_inliers"]]))
dist.destroy_process_group()
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(32500 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, "-c", code, root], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        out, errtxt = p.communicate(timeout=600)
        assert p.returncode == 0, out + errtxt
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1]
        rank, ok, its, inl = json.loads(line[7:])
        assert ok and its == 5000 and inl > 500, (rank, ok, its, inl)


@pytest.mark.parametrize("mode,count", [("", 40), ("SOAK_FUZZ", 20), ("SOAK_FUZZ2", 12), ("SOAK_FUZZ3", 20), ("SOAK_FUZZ4", 20)])
def test_random_problem_soak(gpu, monkeypatch, mode, count):
    """tests/parity_soak.py in small: 4 x `count` random problems against the oracle - iterations, inlier count, mask,
    model.  Modes: plain (12..3000 correspondences, 10..70 % outliers, default / fixed-length / PROSAC option sets), odd
    option values (SOAK_FUZZ: success_prob = 1, max < min iterations, ...), final-refinement options and camera
    models (SOAK_FUZZ2), degraded data (SOAK_FUZZ3), warm starts and real_focal_check (SOAK_FUZZ4)."""
    import importlib.util
    import os

    if mode:
        monkeypatch.setenv(mode, "1")
        monkeypatch.setenv("SOAK_NMAX", "800")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parity_soak.py")
    spec = importlib.util.spec_from_file_location("parity_soak", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(count, 2024) == 0


def test_non_finite_inputs_do_not_hang_or_crash(gpu):
    """NaN / inf correspondences: the reference happily computes with them (comparisons fail, the point is an
    outlier); the device path must do the same - same counts and masks as the oracle, no hang"""
    d = synth.absolute_pose_scene(600, 0.3, 77)
    p2d, p3d = d["p2d"].copy(), d["p3d"].copy()
    p2d[5] = [np.nan, 10.0]
    p3d[9] = [np.inf, 0.0, 1.0]
    p2d[17] = [1e300, -1e300]
    opt = {"ransac": {"seed": 2, "max_iterations": 500, "min_iterations": 100}}
    img, info = gpu.estimate_absolute_pose(p2d, p3d, d["camera"], opt)
    pose, mask, st = O.estimate_absolute_pose(p2d, p3d, d["camera"], opt)
    assert info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()
    h = synth.homography_scene(600, 0.3, 78)
    x1, x2 = h["x1"].copy(), h["x2"].copy()
    x1[3] = [np.nan, np.nan]
    H, info = gpu.estimate_homography(x1, x2, opt)
    Ho, mask, st = O.estimate_homography(x1, x2, opt)
    assert info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
    assert (np.array(info["inliers"]) == mask).all()


def test_throughput_mode_full_size_properties(gpu):
    """BASELINE config 1 at full size (100k iterations): size-independent properties — the run is
    deterministic, the reported best score is reproduced by re-scoring the returned model, and the
    mask agrees with the oracle's mask for that model."""
    d = synth.absolute_pose_scene(5000, 0.7, 1001)
    un = O.unproject(d["camera"], d["p2d"])
    thr = 12.0 / 1000.0
    prob = gpu.Problem(gpu.KIND_ABS, un, d["p3d"])
    opt = {"max_error": thr, "ransac": {"max_iterations": 100000, "min_iterations": 100000, "seed": 1}}
    pose1, info1 = prob.run(opt)
    pose2, info2 = prob.run(opt)
    assert info1["iterations"] == 100000 and info1["hypotheses"] == info2["hypotheses"]
    assert (pose1.q == pose2.q).all() and (pose1.t == pose2.t).all()
    assert info1["inliers"] == info2["inliers"]
    m = O.inliers("reproj", np.r_[pose1.q, pose1.t], un, d["p3d"], thr * thr)
    assert (np.array(info1["inliers"]) == m).all()
    sc, cnt = prob.score(pose1, thr)
    assert cnt == info1["num_inliers"]
    # the first 3000 iterations of the same seed reproduce the oracle's loop exactly
    opt_small = {"max_error": thr, "ransac": {"max_iterations": 3000, "min_iterations": 3000, "seed": 1}}
    pose3, info3 = prob.run(opt_small)
    rp, rm, rst = O.ransac_pnp(un, d["p3d"], opt_small)
    assert info3["hypotheses"] == rst["hypotheses"], (info3["hypotheses"], rst["hypotheses"])
    assert info3["refinements"] == rst["refinements"] and info3["num_inliers"] == rst["num_inliers"]
    assert (np.array(info3["inliers"]) == rm).all()
    print("throughput: hyp", info1["hypotheses"], "seconds", info1["seconds"], "score kernel ms",
          info1["score_kernel_ms"], "=> hyp/s", info1["hypotheses"] / info1["seconds"])
    prob.close()


def test_decision_scores_and_small_problem_refinements_equal_the_oracle_bit_for_bit(gpu):
    """Every score a RANSAC decision is taken on comes from k_score_seq, which adds the terms in the reference's order:
    the inliers' r^2 and one product for the outliers (absolute pose, utils.cc:57-63), r^2 OR thr^2 correspondence by
    correspondence (two-view scores, utils.cc:188-198, 230-235, 320-325).  pl_score_model must therefore EQUAL the
    oracle's score, not approximate it - with a handful of correspondences whole runs hinge on exact ties.  Likewise the
    LM of problems up to 256 correspondences sums its normal equations correspondence by correspondence: the refined
    models equal the oracle's to the bit (the one libm call of the LM loop, pow(., 3), is a correctly rounded cube on the
    device: equal to glibc's for 99.9 % of the arguments, hence the few exceptions allowed below)."""
    rs = np.random.RandomState(17)
    checked = 0
    for trial in range(24):
        n = [8, 40, 200, 256, 257, 3000][trial % 6]
        outl = [0.3, 0.6][trial % 2]
        # ---- absolute pose
        d = synth.absolute_pose_scene(n, outl, 7000 + trial)
        par = d["camera"]["params"]
        x = (np.asarray(d["p2d"]) - np.array(par[-2:])) / par[0]
        X = np.asarray(d["p3d"])
        prob = gpu.Problem(gpu.KIND_ABS, x, X)
        for k in range(3):
            q = np.asarray(d["q_gt"]) + 0.01 * k * rs.randn(4)
            q /= np.linalg.norm(q)
            t = np.asarray(d["t_gt"]) + 0.01 * k * rs.randn(3)
            sc, cnt = prob.score(gpu.CameraPose(q, t), 0.012)
            osc, ocnt = O.score("reproj", np.r_[q, t], x, X, 0.012 ** 2)
            assert (sc, cnt) == (osc, ocnt), ("abs", n, k, sc, osc)
            checked += 1
        prob.close()
        # ---- two-view
        r = synth.relative_pose_scene(n, outl, 7100 + trial)
        x1, x2 = (np.asarray(r["x1"]) - 500.0) / 1000.0, (np.asarray(r["x2"]) - 500.0) / 1000.0
        h = synth.homography_scene(n, outl, 7200 + trial)
        y1, y2 = (np.asarray(h["x1"]) - 500.0) / 1000.0, (np.asarray(h["x2"]) - 500.0) / 1000.0
        pr, pf, ph = gpu.Problem(gpu.KIND_REL, x1, x2), gpu.Problem(gpu.KIND_FUND, x1, x2), gpu.Problem(gpu.KIND_HOM, y1, y2)
        Hm, _ = gpu.ransac_homography(y1, y2, {"max_error": 1e-3, "ransac": {"seed": trial, "max_iterations": 300}})
        for k in range(3):
            q = np.asarray(r["q_gt"]) + 0.003 * k * rs.randn(4)
            q /= np.linalg.norm(q)
            t = np.asarray(r["t_gt"]) + 0.003 * k * rs.randn(3)
            for thr in (1e-3, 5e-2):
                sc, cnt = pr.score(gpu.CameraPose(q, t), thr)
                osc, ocnt = O.score("sampson_pose", np.r_[q, t], x1, x2, thr * thr)
                assert (sc, cnt) == (osc, ocnt), ("rel", n, k, thr, sc, osc)
                tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
                F = tx @ rot(q) * 10.0 ** rs.randint(-3, 4)
                sc, cnt = pf.score(F, thr)
                osc, ocnt = O.score("sampson_F", F, x1, x2, thr * thr)
                assert (sc, cnt) == (osc, ocnt), ("fund", n, k, thr, sc, osc)
                Hk = Hm + 1e-4 * k * np.abs(Hm).max() * rs.randn(3, 3)
                sc, cnt = ph.score(Hk, thr)
                osc, ocnt = O.score("homography", Hk, y1, y2, thr * thr)
                assert (sc, cnt) == (osc, ocnt), ("hom", n, k, thr, sc, osc)
                checked += 3
        # ---- LM on small problems: bit-identical refined models
        if n <= 256:
            exact = total = 0
            for loss in ("TRUNCATED", "CAUCHY", "TRIVIAL"):
                bo = {"loss_type": loss, "loss_scale": 1e-3, "max_iterations": 25}
                q = np.asarray(r["q_gt"]) + 1e-3 * rs.randn(4)
                q /= np.linalg.norm(q)
                t = np.asarray(r["t_gt"]) + 1e-3 * rs.randn(3)
                got, it = pr.refine(gpu.CameraPose(q, t), bo)
                want, st = O.refine("relpose", x1, x2, np.r_[q, t], bo)
                exact += int((np.r_[got.q, got.t] == want).all() and it == st.iterations)
                M = Hm + 1e-4 * np.abs(Hm).max() * rs.randn(3, 3)
                got, it = ph.refine(M, bo)
                want, st = O.refine("homography", y1, y2, M, bo)
                exact += int((np.ravel(got) == np.ravel(want)).all() and it == st.iterations)
                total += 2
            assert exact >= total - 1, (n, exact, total)
        pr.close(), pf.close(), ph.close()
    assert checked > 500


def test_ordered_lm_mode_refines_bit_for_bit_at_every_size(gpu):
    """pl_set_lm_mode(1) routes every refinement through k_lm_ordered: the normal equations and the robust cost are added
    correspondence after correspondence like optim/jacobian_accumulator.h:82-97 - for EVERY n, not only up to 256 (the producers
    hand the entry terms of 64 correspondences at a time through an LDS ring to one wavefront that adds them in order).  Refined
    models, iteration counts and - through ransac_* - complete runs must EQUAL the oracle's (the exception budget is the cube
    of the Nielsen update, 0.08 % of the arguments, as in the 256-correspondence test)."""
    rs = np.random.RandomState(23)
    prev = gpu.set_lm_mode(True)
    try:
        exact = total = 0
        for trial, n in enumerate([257, 300, 1000, 2750, 5000, 10000, 12000]):
            outl = [0.3, 0.6][trial % 2]
            d = synth.absolute_pose_scene(n, outl, 7400 + trial)
            par = d["camera"]["params"]
            x = (np.asarray(d["p2d"]) - np.array(par[-2:])) / par[0]
            X = np.asarray(d["p3d"])
            r = synth.relative_pose_scene(n, outl, 7500 + trial)
            x1, x2 = (np.asarray(r["x1"]) - 500.0) / 1000.0, (np.asarray(r["x2"]) - 500.0) / 1000.0
            h = synth.homography_scene(n, outl, 7600 + trial)
            y1, y2 = (np.asarray(h["x1"]) - 500.0) / 1000.0, (np.asarray(h["x2"]) - 500.0) / 1000.0
            pa, pr, pf, ph = (gpu.Problem(gpu.KIND_ABS, x, X), gpu.Problem(gpu.KIND_REL, x1, x2), gpu.Problem(gpu.KIND_FUND, x1, x2),
                              gpu.Problem(gpu.KIND_HOM, y1, y2))
            Hm, _ = gpu.ransac_homography(y1, y2, {"max_error": 1e-3, "ransac": {"seed": trial, "max_iterations": 300}})
            Fm, _ = gpu.ransac_fundamental(x1, x2, {"max_error": 1e-3, "ransac": {"seed": trial, "max_iterations": 300}})
            for loss in ("TRUNCATED", "CAUCHY", "HUBER"):
                bo = {"loss_type": loss, "loss_scale": 1e-3, "max_iterations": 25}
                q = np.asarray(d["q_gt"]) + 1e-3 * rs.randn(4)
                q /= np.linalg.norm(q)
                t = np.asarray(d["t_gt"]) + 1e-3 * rs.randn(3)
                got, it = pa.refine(gpu.CameraPose(q, t), dict(bo, loss_scale=0.012))
                want, st = O.bundle_adjust(x, X, {"model": "NULL", "width": 0, "height": 0, "params": []}, np.r_[q, t], dict(bo, loss_scale=0.012))
                exact += int((np.r_[got.q, got.t] == want).all() and it == st.iterations)
                q = np.asarray(r["q_gt"]) + 1e-3 * rs.randn(4)
                q /= np.linalg.norm(q)
                t = np.asarray(r["t_gt"]) + 1e-3 * rs.randn(3)
                got, it = pr.refine(gpu.CameraPose(q, t), bo)
                want, st = O.refine("relpose", x1, x2, np.r_[q, t], bo)
                exact += int((np.r_[got.q, got.t] == want).all() and it == st.iterations)
                M = Hm + 1e-4 * np.abs(Hm).max() * rs.randn(3, 3)
                got, it = ph.refine(M, bo)
                want, st = O.refine("homography", y1, y2, M, bo)
                exact += int((np.ravel(got) == np.ravel(want)).all() and it == st.iterations)
                got, it = pf.refine(Fm, bo)
                want, st = O.refine("fundamental", x1, x2, Fm, bo)
                exact += int((np.ravel(got) == np.ravel(want)).all() and it == st.iterations)
                total += 4
            # a complete run: every LO and the final refinement in the reference's order
            opt = {"max_error": 0.012, "ransac": {"seed": 5 + trial, "max_iterations": 2000, "min_iterations": 200}}
            pose, info = gpu.ransac_pnp(x, X, opt)
            ref, mask, st = O.ransac_pnp(x, X, opt)
            assert (info["iterations"], info["refinements"], info["num_inliers"]) == (st["iterations"], st["refinements"], st["num_inliers"])
            assert (np.array(info["inliers"], dtype=bool) == mask).all()
            exact += int((np.r_[pose.q, pose.t] == np.asarray(ref)).all())
            total += 1
            for p in (pa, pr, pf, ph):
                p.close()
        assert exact >= total - 2, (exact, total)
    finally:
        gpu.set_lm_mode(prev)
