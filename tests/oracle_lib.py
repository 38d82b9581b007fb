"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference hot path).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg only.  The product package (poselib_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "liboracle.so")

u64, i32, f64 = C.c_uint64, C.c_int32, C.c_double
PD = C.POINTER(C.c_double)


class RansacOpt(C.Structure):
    _fields_ = [("max_iterations", u64), ("min_iterations", u64), ("dyn_num_trials_mult", f64), ("success_prob", f64),
                ("seed", u64), ("progressive_sampling", i32), ("score_initial_model", i32),
                ("max_prosac_iterations", u64)]


class BundleOpt(C.Structure):
    _fields_ = [("max_iterations", u64), ("loss_type", i32), ("lambda_update", i32), ("damping", i32),
                ("refine_flags", i32), ("loss_scale", f64), ("gradient_tol", f64), ("step_tol", f64),
                ("relative_cost_tol", f64), ("initial_lambda", f64), ("min_lambda", f64), ("max_lambda", f64),
                ("lambda_factor", f64)]


class RobustOpt(C.Structure):
    _fields_ = [("ransac", RansacOpt), ("bundle", BundleOpt), ("max_error", f64), ("real_focal_check", i32),
                ("estimate_focal_length", i32), ("min_fov", f64)]


class Stats(C.Structure):
    _fields_ = [("refinements", u64), ("iterations", u64), ("num_inliers", u64), ("inlier_ratio", f64),
                ("model_score", f64), ("hypotheses", u64), ("seconds", f64)]


class Camera(C.Structure):
    _fields_ = [("model_id", i32), ("width", i32), ("height", i32), ("num_params", i32), ("params", f64 * 12)]


class BundleStats(C.Structure):
    _fields_ = [("iterations", u64), ("initial_cost", f64), ("cost", f64), ("lambda_", f64), ("nu", f64),
                ("invalid_steps", u64), ("step_norm", f64), ("grad_norm", f64)]


LOSS = {"TRIVIAL": 0, "TRUNCATED": 1, "HUBER": 2, "CAUCHY": 3, "TRUNCATED_CAUCHY": 4, "TRUNCATED_LE_ZACH": 5}
MODEL_IDS = {"NULL": -1, "SIMPLE_PINHOLE": 0, "PINHOLE": 1, "OPENCV": 4}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_ORACLE_DIR, "src", f) for f in os.listdir(os.path.join(_ORACLE_DIR, "src"))]
    srcs.append(os.path.join(_ORACLE_DIR, "oracle.h"))
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_all_inlier_probability.restype = f64
        _lib.orc_all_inlier_probability.argtypes = [u64, u64, u64]
        _lib.orc_dynamic_max_iter.restype = u64
        _lib.orc_dynamic_max_iter.argtypes = [u64, u64, u64, f64, f64, u64, u64]
        _lib.orc_random_int.restype = i32
        for name in ("orc_score_reproj", "orc_score_sampson_pose", "orc_score_sampson_F", "orc_score_homography"):
            getattr(_lib, name).restype = f64
        _lib.orc_normalize_points.restype = f64
        _lib.orc_solve_cubic_single_real.argtypes = [f64, f64, f64, C.c_void_p]
        _lib.orc_solve_cubic_real.argtypes = [f64, f64, f64, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ransac_opt(d=None) -> RansacOpt:
    d = d or {}
    return RansacOpt(d.get("max_iterations", 100000), d.get("min_iterations", 1000), d.get("dyn_num_trials_mult", 3.0),
                     d.get("success_prob", 0.9999), d.get("seed", 0), int(d.get("progressive_sampling", False)),
                     int(d.get("score_initial_model", False)), d.get("max_prosac_iterations", 100000))


def bundle_opt(d=None) -> BundleOpt:
    d = d or {}
    lt = d.get("loss_type", "CAUCHY")
    lt = LOSS[lt] if isinstance(lt, str) else int(lt)
    flags = (1 if d.get("refine_focal_length") else 0) | (2 if d.get("refine_principal_point") else 0) | (4 if d.get("refine_extra_params") else 0)
    return BundleOpt(d.get("max_iterations", 100), lt, int(d.get("lambda_update", 0)), int(d.get("damping", 0)), flags,
                     d.get("loss_scale", 1.0), d.get("gradient_tol", 1e-12), d.get("step_tol", 1e-8),
                     d.get("relative_cost_tol", 1e-10), d.get("initial_lambda", 1e-3), d.get("min_lambda", 1e-10),
                     d.get("max_lambda", 1e10), d.get("lambda_factor", 10.0))


def robust_opt(d=None, default_max_error=1.0) -> RobustOpt:
    d = d or {}
    return RobustOpt(ransac_opt(d.get("ransac")), bundle_opt(d.get("bundle")), d.get("max_error", default_max_error),
                     int(d.get("real_focal_check", False)), int(d.get("estimate_focal_length", False)), float(d.get("min_fov", 5.0)))


def camera(d) -> Camera:
    c = Camera()
    m = d["model"]
    c.model_id = MODEL_IDS[m] if isinstance(m, str) else int(m)
    c.width = int(d.get("width", 0))
    c.height = int(d.get("height", 0))
    params = list(d.get("params", []))
    c.num_params = len(params)
    for i, v in enumerate(params):
        c.params[i] = v
    return c


def stats_dict(s: Stats):
    return {k: getattr(s, k) for k, _ in Stats._fields_}


# ------------------------------------------------------------------ sampler / control
def sampler_draw(seed, N, K, n_samples, prosac=False, max_prosac=100000):
    out = np.zeros((n_samples, K), dtype=np.uint64)
    st = u64(0)
    lib().orc_sampler_draw(u64(seed), u64(N), u64(K), u64(n_samples), i32(int(prosac)), u64(max_prosac), _p(out),
                           C.byref(st))
    return out, st.value


def mock_ransac(num_data, sample_sz, inlier_count, ropt):
    st = Stats()
    o = ransac_opt(ropt)
    lib().orc_mock_ransac(u64(num_data), u64(sample_sz), u64(inlier_count), C.byref(o), C.byref(st))
    return stats_dict(st)


# ------------------------------------------------------------------ solvers
def p3p(x, X):
    x, X = _f(x), _f(X)
    out = np.zeros((4, 7))
    n = lib().orc_p3p(_p(x), _p(X), _p(out))
    return out[:n]


def essential_5pt(x1, x2):
    x1, x2 = _f(x1), _f(x2)
    out = np.zeros((10, 9))
    n = lib().orc_essential_5pt(_p(x1), _p(x2), _p(out))
    return [out[i].reshape(3, 3).T.copy() for i in range(n)]


def relpose_5pt(x1, x2):
    x1, x2 = _f(x1), _f(x2)
    out = np.zeros((40, 7))
    n = lib().orc_relpose_5pt(_p(x1), _p(x2), _p(out))
    return out[:n]


def relpose_7pt(x1, x2):
    x1, x2 = _f(x1), _f(x2)
    out = np.zeros((3, 9))
    n = lib().orc_relpose_7pt(_p(x1), _p(x2), _p(out))
    return [out[i].reshape(3, 3).T.copy() for i in range(n)]


def homography_4pt(x1, x2, check_cheirality=True):
    x1, x2 = _f(x1), _f(x2)
    out = np.zeros(9)
    n = lib().orc_homography_4pt(_p(x1), _p(x2), _p(out), int(check_cheirality))
    return n, out.reshape(3, 3).T.copy()


def sturm_roots(coeffs, tol=None):
    c = _f(coeffs)
    out = np.zeros(len(c))
    if tol is None:
        n = lib().orc_sturm_roots(_p(c), len(c) - 1, _p(out))
    else:
        n = lib().orc_sturm_roots_tol(_p(c), len(c) - 1, C.c_double(tol), _p(out))
    return out[:n]


def nullspace(A):
    """A: rows x cols (rows >= cols) -> rows x (rows-cols) orthonormal complement basis."""
    A = np.asarray(A, dtype=np.float64)
    rows, cols = A.shape
    Af = np.asfortranarray(A)
    out = np.zeros((rows - cols, rows))
    lib().orc_nullspace(Af.ctypes.data_as(C.c_void_p), rows, cols, _p(out))
    return out.T.copy()


# ------------------------------------------------------------------ scoring
def _mat(M):
    return np.ascontiguousarray(np.asarray(M, dtype=np.float64).T.reshape(9))  # column-major


def score(kind, model, a, b, sq_thr):
    a, b = _f(a), _f(b)
    cnt = u64(0)
    fn = {"reproj": "orc_score_reproj", "sampson_pose": "orc_score_sampson_pose", "sampson_F": "orc_score_sampson_F",
          "homography": "orc_score_homography"}[kind]
    m = _f(model) if kind in ("reproj", "sampson_pose") else _mat(model)
    sc = getattr(lib(), fn)(_p(m), _p(a), _p(b), C.c_size_t(a.shape[0]), f64(sq_thr), C.byref(cnt))
    return sc, cnt.value


def inliers(kind, model, a, b, sq_thr):
    a, b = _f(a), _f(b)
    mask = np.zeros(a.shape[0], dtype=np.uint8)
    fn = {"reproj": "orc_inliers_reproj", "sampson_pose": "orc_inliers_sampson_pose",
          "sampson_F": "orc_inliers_sampson_F", "homography": "orc_inliers_homography"}[kind]
    m = _f(model) if kind in ("reproj", "sampson_pose") else _mat(model)
    getattr(lib(), fn)(_p(m), _p(a), _p(b), C.c_size_t(a.shape[0]), f64(sq_thr), _p(mask))
    return mask.astype(bool)


def unproject(cam_dict, xp):
    xp = _f(xp)
    out = np.zeros_like(xp)
    c = camera(cam_dict)
    lib().orc_unproject(C.byref(c), _p(xp), C.c_size_t(xp.shape[0]), _p(out))
    return out


def normalize_points(x1, x2, scale=True, centroid=True, shared=True):
    a, b = _f(x1).copy(), _f(x2).copy()
    T1, T2 = np.zeros(9), np.zeros(9)
    s = lib().orc_normalize_points(_p(a), _p(b), C.c_size_t(a.shape[0]), _p(T1), _p(T2), int(scale), int(centroid),
                                   int(shared))
    return s, a, b, T1.reshape(3, 3).T.copy(), T2.reshape(3, 3).T.copy()


# ------------------------------------------------------------------ refinement
def bundle_adjust(x, X, cam_dict, pose7, bopt=None):
    x, X = _f(x), _f(X)
    p = _f(pose7).copy()
    c = camera(cam_dict)
    o = bundle_opt(bopt)
    st = BundleStats()
    lib().orc_bundle_adjust(_p(x), _p(X), C.c_size_t(x.shape[0]), C.byref(c), _p(p), C.byref(o), C.byref(st))
    return p, st


def bundle_adjust_camera(x, X, cam_dict, pose7, bopt=None):
    """bundle_adjust with the camera in / out: bopt["refine_focal_length" / "refine_principal_point" / "refine_extra_params"]
    name the parameters refined along with the pose.  Returns (pose, camera parameters, stats)."""
    x, X = _f(x), _f(X)
    p = _f(pose7).copy()
    c = camera(cam_dict)
    o = bundle_opt(bopt)
    st = BundleStats()
    lib().orc_bundle_adjust_camera(_p(x), _p(X), C.c_size_t(x.shape[0]), C.byref(c), _p(p), C.byref(o), C.byref(st))
    return p, np.array(c.params[: c.num_params]), st


def p35pf(x, X):
    """solvers/p35pf.h: x 4 x 2 image points (principal point at the origin), X 4 x 3.
    Returns (poses n x 7, focals n)."""
    x, X = _f(x), _f(X)
    poses = np.zeros((10, 7))
    focals = np.zeros(10)
    n = lib().orc_p35pf(_p(x), _p(X), _p(poses), _p(focals))
    return poses[:n].copy(), focals[:n].copy()


def relpose_6pt_shared_focal(b1, b2):
    """solvers/relpose_6pt_focal.h: six pairs of unit bearings (principal point at the origin, unit focal length).
    Returns (poses n x 7, focals n) in the solver's order."""
    b1, b2 = _f(b1), _f(b2)
    poses = np.zeros((60, 7))
    focals = np.zeros(60)
    n = lib().orc_relpose_6pt_shared_focal(_p(b1), _p(b2), _p(poses), _p(focals))
    return poses[:n].copy(), focals[:n].copy()


class libm_cubes:
    """Context: the six-point solver's cubes through std::pow(d, 3) like the reference's formulas (glibc: the correctly rounded cube
    except for ~8 of 10^4 arguments, one unit in the last place) instead of the oracle's default, the correctly rounded cube the
    device computes - oracle/src/solvers_focal.cc.  For comparisons with oracle/_ref (tests/ref_lib.py), which always runs the
    reference's own call."""

    def __enter__(self):
        lib().orc_set_exact_cubes(0)
        return self

    def __exit__(self, *exc):
        lib().orc_set_exact_cubes(1)
        return False


def ransac_pnpf(x, X, opt=None):
    """robust/ransac.h ransac_pnpf: pose + focal length of a SIMPLE_PINHOLE camera with the
    principal point at the origin.  Returns (pose7, focal, mask, stats)."""
    x, X = _f(x), _f(X)
    n = x.shape[0]
    o = robust_opt(opt, 12.0)
    pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
    focal = C.c_double(0.0)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = Stats()
    lib().orc_ransac_pnpf(_p(x), _p(X), C.c_size_t(n), C.byref(o), _p(pose), C.byref(focal), _p(mask), C.byref(st))
    return pose, focal.value, mask[:n].astype(bool), stats_dict(st)


def estimate_shared_focal_relative_pose(x1, x2, pp, opt=None, init_pose=None, init_focal=1.0):
    """robust.h:84-87: relative pose of two views with ONE unknown focal length; pixel points, principal point pp.
    Returns (pose7, focal, mask, stats)."""
    x1, x2 = _f(x1), _f(x2)
    n = x1.shape[0]
    o = robust_opt(opt, 1.0)
    pose = np.array([1.0, 0, 0, 0, 0, 0, 0]) if init_pose is None else _f(init_pose).copy()
    focal = C.c_double(init_focal)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = Stats()
    ppa = _f(pp)
    lib().orc_estimate_shared_focal_relative_pose(_p(x1), _p(x2), C.c_size_t(n), _p(ppa), C.byref(o), _p(pose), C.byref(focal),
                                                  _p(mask), C.byref(st))
    return pose, focal.value, mask[:n].astype(bool), stats_dict(st)


def ransac_shared_focal_relpose(x1, x2, opt=None, init_pose=None, init_focal=1.0):
    """robust/ransac.h:71-73 ransac_shared_focal_relpose: points relative to the principal point.
    Returns (pose7, focal, mask, stats)."""
    x1, x2 = _f(x1), _f(x2)
    n = x1.shape[0]
    o = robust_opt(opt, 1.0)
    pose = np.array([1.0, 0, 0, 0, 0, 0, 0]) if init_pose is None else _f(init_pose).copy()
    focal = C.c_double(init_focal)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = Stats()
    lib().orc_ransac_shared_focal_relpose(_p(x1), _p(x2), C.c_size_t(n), C.byref(o), _p(pose), C.byref(focal), _p(mask),
                                          C.byref(st))
    return pose, focal.value, mask[:n].astype(bool), stats_dict(st)


def refine_shared_focal_relpose(x1, x2, pose, focal, bopt=None):
    """robust/bundle.h:108-111.  Returns (pose7, focal, BundleStats)."""
    x1, x2 = _f(x1), _f(x2)
    o = bundle_opt(bopt)
    st = BundleStats()
    p = _f(pose).copy()
    f = C.c_double(focal)
    lib().orc_refine_shared_focal_relpose(_p(x1), _p(x2), C.c_size_t(x1.shape[0]), _p(p), C.byref(f), C.byref(o), C.byref(st))
    return p, f.value, st


def refine(kind, x1, x2, model, bopt=None):
    x1, x2 = _f(x1), _f(x2)
    o = bundle_opt(bopt)
    st = BundleStats()
    if kind == "relpose":
        m = _f(model).copy()
        lib().orc_refine_relpose(_p(x1), _p(x2), C.c_size_t(x1.shape[0]), _p(m), C.byref(o), C.byref(st))
        return m, st
    m = _mat(model).copy()
    fn = {"homography": "orc_refine_homography", "fundamental": "orc_refine_fundamental"}[kind]
    getattr(lib(), fn)(_p(x1), _p(x2), C.c_size_t(x1.shape[0]), _p(m), C.byref(o), C.byref(st))
    return m.reshape(3, 3).T.copy(), st


# ------------------------------------------------------------------ RANSAC entry points / front-ends
def _run_pose(fn, a, b, opt, default_err, init_pose=None, cams=None):
    a, b = _f(a), _f(b)
    n = a.shape[0]
    pose = np.array([1.0, 0, 0, 0, 0, 0, 0]) if init_pose is None else _f(init_pose).copy()
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = Stats()
    o = robust_opt(opt, default_err)
    args = [_p(a), _p(b), C.c_size_t(n)]
    if cams is not None:
        args += [C.byref(c) for c in cams]
    args += [C.byref(o), _p(pose), _p(mask), C.byref(st)]
    getattr(lib(), fn)(*args)
    return pose, mask[:n].astype(bool), stats_dict(st)


def _run_mat(fn, a, b, opt, init=None):
    a, b = _f(a), _f(b)
    n = a.shape[0]
    M = _mat(np.eye(3) if init is None else init).copy()
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = Stats()
    o = robust_opt(opt, 1.0)
    getattr(lib(), fn)(_p(a), _p(b), C.c_size_t(n), C.byref(o), _p(M), _p(mask), C.byref(st))
    return M.reshape(3, 3).T.copy(), mask[:n].astype(bool), stats_dict(st)


def ransac_pnp(x, X, opt=None, init_pose=None):
    return _run_pose("orc_ransac_pnp", x, X, opt, 12.0, init_pose)


def ransac_relpose(x1, x2, opt=None, init_pose=None):
    return _run_pose("orc_ransac_relpose", x1, x2, opt, 1.0, init_pose)


def ransac_fundamental(x1, x2, opt=None, init=None):
    return _run_mat("orc_ransac_fundamental", x1, x2, opt, init)


def ransac_homography(x1, x2, opt=None, init=None):
    return _run_mat("orc_ransac_homography", x1, x2, opt, init)


def estimate_absolute_pose(p2d, p3d, cam_dict, opt=None, init_pose=None, return_camera=False):
    a, b = _f(p2d), _f(p3d)
    n = a.shape[0]
    pose = np.array([1.0, 0, 0, 0, 0, 0, 0]) if init_pose is None else _f(init_pose).copy()
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = Stats()
    o = robust_opt(opt, 12.0)
    c = camera(cam_dict)
    lib().orc_estimate_absolute_pose(_p(a), _p(b), C.c_size_t(n), C.byref(o), C.byref(c), _p(pose), _p(mask),
                                     C.byref(st))
    if return_camera:
        return pose, mask[:n].astype(bool), stats_dict(st), np.array(c.params[: c.num_params])
    return pose, mask[:n].astype(bool), stats_dict(st)


def estimate_relative_pose(x1, x2, cam1, cam2, opt=None, init_pose=None):
    return _run_pose("orc_estimate_relative_pose", x1, x2, opt, 1.0, init_pose, cams=[camera(cam1), camera(cam2)])


def estimate_fundamental(x1, x2, opt=None, init=None):
    return _run_mat("orc_estimate_fundamental", x1, x2, opt, init)


def estimate_homography(x1, x2, opt=None, init=None):
    return _run_mat("orc_estimate_homography", x1, x2, opt, init)
