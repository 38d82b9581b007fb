"""ctypes binding of tests/hostmath/libhostmath.so — a TEST-ONLY host compilation of the product's
device math headers (poselib_amd/csrc/pl_*.h).  Lets the CPU test-suite compare the exact source
that hipcc compiles into the kernels against the oracle.  Never used by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostmath")
_LIB = os.path.join(_DIR, "libhostmath.so")
EST = {"abs": 0, "rel": 1, "fund": 2, "hom": 3}
STRIDE = 24
MAT = 7


class LMOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_uint32), ("loss_type", C.c_int32), ("lambda_update", C.c_int32),
                ("damping", C.c_int32), ("loss_scale", C.c_double), ("gradient_tol", C.c_double),
                ("step_tol", C.c_double), ("relative_cost_tol", C.c_double), ("initial_lambda", C.c_double),
                ("min_lambda", C.c_double), ("max_lambda", C.c_double), ("lambda_factor", C.c_double)]


class CameraParams(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("num_params", C.c_int32), ("p", C.c_double * 12)]


def lm_options(max_iterations=100, loss_type=3, loss_scale=1.0, **kw):
    return LMOptions(max_iterations, loss_type, kw.get("lambda_update", 0), kw.get("damping", 0), loss_scale,
                     kw.get("gradient_tol", 1e-12), kw.get("step_tol", 1e-8), kw.get("relative_cost_tol", 1e-10),
                     kw.get("initial_lambda", 1e-3), kw.get("min_lambda", 1e-10), kw.get("max_lambda", 1e10),
                     kw.get("lambda_factor", 10.0))


def camera_params(model_id=-1, params=()):
    c = CameraParams()
    c.model_id = model_id
    c.num_params = len(params)
    for i, v in enumerate(params):
        c.p[i] = v
    return c


_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_DIR, "hostmath.cc")]
        csrc = os.path.join(os.path.dirname(_DIR), "..", "poselib_amd", "csrc")
        srcs += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.startswith("pl_") and f.endswith(".h")]
        if not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
            subprocess.check_call(["make", "-C", _DIR, "-s", "libhostmath.so"])
        _lib = C.CDLL(_LIB)
        _lib.hm_score.restype = C.c_double
        _lib.hm_draw_samples.restype = C.c_uint32
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _soa(arrs):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    return arrs, ptrs


def draw_samples(seed, N, K, n_iters):
    idx = np.zeros((n_iters, K), dtype=np.uint32)
    pos = np.zeros(n_iters, dtype=np.uint32)
    total = lib().hm_draw_samples(C.c_uint64(seed), C.c_uint64(N), K, n_iters, _p(idx), _p(pos))
    return idx, pos, total


def solve(est, first, second):
    inp = np.ascontiguousarray(np.concatenate([np.asarray(first, float).ravel(), np.asarray(second, float).ravel()]))
    maxm = {0: 4, 1: 40, 2: 3, 3: 1}[EST[est]]
    rec = np.zeros((maxm, STRIDE))
    n = lib().hm_solve(EST[est], _p(inp), _p(rec))
    return rec[:n]


def essential_5pt(x1, x2):
    inp = np.ascontiguousarray(np.concatenate([np.asarray(x1, float).ravel(), np.asarray(x2, float).ravel()]))
    out = np.zeros((10, 9))
    n = lib().hm_essential_5pt(_p(inp), _p(out))
    return [out[i].reshape(3, 3).copy() for i in range(n)]


def sturm10(coef, flat=False):
    """real roots of a degree-10 polynomial; flat=True: through the batched generator's isolation (sturm_isolate_flat)"""
    c = np.ascontiguousarray(coef, dtype=np.float64)
    out = np.zeros(10)
    n = (lib().hm_sturm10_flat if flat else lib().hm_sturm10)(_p(c), _p(out))
    return out[:n]


def pose_record(q, t, essential=False):
    rec = np.zeros(STRIDE)
    q = np.ascontiguousarray(q, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    lib().hm_pose_record(_p(q), _p(t), int(essential), _p(rec))
    return rec


def matrix_record(M):
    rec = np.zeros(STRIDE)
    M = np.ascontiguousarray(np.asarray(M, float).reshape(9))
    lib().hm_matrix_record(_p(M), _p(rec))
    return rec


def prefilter(est, rec, cols, thr2, xy_absmax=0.0):
    """the scoring kernels' fp32 pre-filter: (enabled, mask of correspondences PROVEN not to be inliers)"""
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    out = np.zeros(n, dtype=np.uint8)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    en = lib().hm_prefilter(EST[est], _p(rec), ptrs, C.c_uint32(n), C.c_double(thr2), C.c_float(xy_absmax), _p(out))
    return bool(en), out.astype(bool)


def prefilter16(est, rec, cols, thr2, uv_absmax, random_order=0):
    """the Sampson pre-filter in its fp16 / matrix-core form: (available, mask of PROVEN non-inliers)"""
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    out = np.zeros(n, dtype=np.uint8)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    en = lib().hm_prefilter16(EST[est], _p(rec), ptrs, C.c_uint32(n), C.c_double(thr2), C.c_float(uv_absmax),
                              C.c_uint32(random_order), _p(out))
    return bool(en), out.astype(bool)


def prefilter16_abs(rec, cols, thr2, xy_absmax, random_order=0):
    """the reprojection pre-filter in its fp16 / matrix-core form: (available, mask of PROVEN non-inliers)"""
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    out = np.zeros(n, dtype=np.uint8)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    en = lib().hm_prefilter16_abs(_p(rec), ptrs, C.c_uint32(n), C.c_double(thr2), C.c_float(xy_absmax),
                                  C.c_uint32(random_order), _p(out))
    return bool(en), out.astype(bool)


def prefilter16_hom(rec, cols, thr2, uv_absmax, random_order=0):
    """the homography pre-filter in its fp16 / matrix-core form: (available, mask of PROVEN non-inliers)"""
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    out = np.zeros(n, dtype=np.uint8)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    en = lib().hm_prefilter16_hom(_p(rec), ptrs, C.c_uint32(n), C.c_double(thr2), C.c_float(uv_absmax),
                                  C.c_uint32(random_order), _p(out))
    return bool(en), out.astype(bool)


def half_rn(v):
    """pl_prefilter.h's float -> fp16 conversion: (bits, value of those bits as float32)"""
    v = np.ascontiguousarray(v, dtype=np.float32)
    bits = np.zeros(v.size, dtype=np.uint16)
    back = np.zeros(v.size, dtype=np.float32)
    lib().hm_half_rn(_p(v), C.c_uint64(v.size), _p(bits), _p(back))
    return bits, back


def score(est, rec, cols, thr2):
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    cnt = C.c_uint32(0)
    flags = np.zeros(n, dtype=np.uint8)
    r2 = np.zeros(n)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    sc = lib().hm_score(EST[est], _p(rec), ptrs, C.c_uint32(n), C.c_double(thr2), C.byref(cnt), _p(flags), _p(r2))
    return sc, cnt.value, flags.astype(bool), r2


def mask_abs(rec, cols, thr2):
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    m = np.zeros(n, dtype=np.uint8)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    lib().hm_mask_abs(_p(rec), ptrs, C.c_uint32(n), C.c_double(thr2), _p(m))
    return m.astype(bool)


def unproject(cam: CameraParams, xp):
    xp = np.ascontiguousarray(xp, dtype=np.float64)
    out = np.zeros_like(xp)
    lib().hm_unproject(C.byref(cam), _p(xp), C.c_uint32(xp.shape[0]), _p(out))
    return out


def lm(est, cols, params, opt: LMOptions, cam: CameraParams = None, point_scale=1.0, prefilter_thr2=0.0, mask=None):
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    p = np.zeros(16)
    p[: len(params)] = params
    cam = cam or camera_params()
    it = C.c_uint32(0)
    sk = C.c_uint32(0)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().hm_lm(EST[est], ptrs, C.c_uint32(n), _p(p), C.byref(opt), C.byref(cam), C.c_double(point_scale),
                C.c_double(prefilter_thr2), None if m is None else _p(m), C.byref(it), C.byref(sk))
    return p, it.value, bool(sk.value)


def lm_cam(cols, params, opt: LMOptions, cam: CameraParams, cam_flags, point_scale=1.0, mask=None):
    """k_lm_cam's algorithm serially: (pose parameters, camera parameters, iterations, (initial cost, cost))"""
    arrs, ptrs = _soa(cols)
    n = arrs[0].shape[0]
    p = np.zeros(16)
    p[: len(params)] = params
    c = CameraParams.from_buffer_copy(cam)
    it = C.c_uint32(0)
    costs = np.zeros(2)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().hm_lm_cam(ptrs, C.c_uint32(n), _p(p), C.byref(opt), C.byref(c), C.c_int(cam_flags), C.c_double(point_scale),
                    None if m is None else _p(m), C.byref(it), _p(costs))
    return p, np.array(c.p[: c.num_params]), it.value, costs


def factorized_F(params):
    p = np.zeros(16)
    p[: len(params)] = params
    F = np.zeros(9)
    lib().hm_factorized_F(_p(p), _p(F))
    return F.reshape(3, 3)


def p35pf(x, X, stride=1):
    """pl_solver_p35pf.h on the host; the elimination matrix at `stride` doubles between elements, as on the device"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    poses = np.zeros((10, 7))
    focals = np.zeros(10)
    n = lib().hm_p35pf(_p(x), _p(X), C.c_uint32(stride), _p(poses), _p(focals))
    return poses[:n].copy(), focals[:n].copy()


def relpose_6pt_shared_focal(b1, b2, stride=1):
    """pl_solver_6ptf.h on the host; the workspace at `stride` doubles between elements, as on the device"""
    b1 = np.ascontiguousarray(b1, dtype=np.float64)
    b2 = np.ascontiguousarray(b2, dtype=np.float64)
    poses = np.zeros((60, 7))
    focals = np.zeros(60)
    n = lib().hm_relpose_6pt_shared_focal(_p(b1), _p(b2), C.c_uint32(stride), _p(poses), _p(focals))
    return poses[:n].copy(), focals[:n].copy()


def sfocal_lm(x1, x2, pose, focal, opt, prefilter_thr2=0.0, mask=None):
    """the shared-focal refiner as k_sfocal_lm runs it.  Returns (pose7, focal, iterations, (initial cost, cost), skipped)"""
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    x2 = np.ascontiguousarray(x2, dtype=np.float64)
    n = x1.shape[0]
    cols, ptrs = _soa([x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1]])
    p = np.ascontiguousarray(pose, dtype=np.float64).copy()
    f = C.c_double(focal)
    its = C.c_uint32(0)
    costs = np.zeros(2)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().hm_sfocal_lm.restype = C.c_int
    skipped = lib().hm_sfocal_lm(ptrs, C.c_uint32(n), _p(p), C.byref(f), C.byref(opt), C.c_double(prefilter_thr2),
                                 None if m is None else _p(m), C.byref(its), _p(costs))
    return p, f.value, its.value, costs, bool(skipped)


def ransac_shared_focal(x1, x2, max_error=1.0, seed=0, max_iterations=100000, min_iterations=1000, dyn_mult=3.0,
                        success_prob=0.9999, init_pose=None, init_focal=1.0):
    """the product's ransac_shared_focal_relpose loop (pl_focal.h + pl_sfocal.h) over a serial evaluation of the device
    functions.  Returns (pose7, focal, mask, stats dict)."""
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    x2 = np.ascontiguousarray(x2, dtype=np.float64)
    n = x1.shape[0]
    cols, ptrs = _soa([x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1]])
    pose = np.array([1.0, 0, 0, 0, 0, 0, 0]) if init_pose is None else np.ascontiguousarray(init_pose, dtype=np.float64).copy()
    focal = C.c_double(init_focal)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = np.zeros(5, dtype=np.uint64)
    score = C.c_double(0.0)
    lib().hm_ransac_shared_focal(ptrs, C.c_uint32(n), C.c_uint64(max_iterations), C.c_uint64(min_iterations), C.c_uint64(seed),
                                 C.c_double(dyn_mult), C.c_double(success_prob), C.c_int(int(init_pose is not None)),
                                 C.c_double(max_error), _p(pose), C.byref(focal), _p(mask), _p(st), C.byref(score))
    return pose, focal.value, mask[:n].astype(bool), {"refinements": int(st[0]), "iterations": int(st[1]), "num_inliers": int(st[2]),
                                                     "hypotheses": int(st[3]), "iterations_evaluated": int(st[4]),
                                                     "model_score": score.value}


def set_prosac(on: bool, max_prosac_iterations: int = 100000):
    """PROSAC sampling (sampling.cc:85-136) for ransac_pnpf / ransac_shared_focal below"""
    lib().hm_set_prosac(C.c_int(int(bool(on))), C.c_uint64(max_prosac_iterations))


def ransac_pnpf(x, X, max_error=12.0, seed=0, max_iterations=100000, min_iterations=1000, dyn_mult=3.0, success_prob=0.9999,
                score_initial=False, min_fov=5.0):
    """the product's ransac_pnpf loop (pl_focal.h) over a serial evaluation of the device functions.
    Returns (pose7, focal, mask, stats dict)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = x.shape[0]
    cols, ptrs = _soa([x[:, 0], x[:, 1], X[:, 0], X[:, 1], X[:, 2]])
    pose = np.zeros(7)
    focal = C.c_double(0.0)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    st = np.zeros(5, dtype=np.uint64)
    score = C.c_double(0.0)
    lib().hm_ransac_pnpf(ptrs, C.c_uint32(n), C.c_uint64(max_iterations), C.c_uint64(min_iterations), C.c_uint64(seed),
                         C.c_double(dyn_mult), C.c_double(success_prob), C.c_int(int(score_initial)), C.c_double(max_error),
                         C.c_double(min_fov), _p(pose), C.byref(focal), _p(mask), _p(st), C.byref(score))
    return pose, focal.value, mask[:n].astype(bool), {"refinements": int(st[0]), "iterations": int(st[1]), "num_inliers": int(st[2]),
                                                     "hypotheses": int(st[3]), "iterations_evaluated": int(st[4]),
                                                     "model_score": score.value}
