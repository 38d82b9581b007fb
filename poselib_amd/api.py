"""Python surface mirroring the reference's `poselib` module for the accelerated path.

Signatures, option-dict keys and return shapes follow the pybind11 module of PoseLib 3.0.0:
    estimate_absolute_pose(points2D, points3D, camera, opt={}, initial_pose=None) -> (Image, info)
        pybind/bindings/estimators/absolute_pose.cc:15-47, 317-330
    estimate_relative_pose(points2D_1, points2D_2, camera1, camera2, opt={}, initial_pose=None) -> (CameraPose, info)
        pybind/bindings/estimators/relative_pose.cc:15-52, 405-420
    estimate_fundamental(points2D_1, points2D_2, opt={}, initial_F=None) -> (3x3 ndarray, info)   (same file :221-243)
    estimate_homography(points2D_1, points2D_2, opt={}, initial_H=None) -> (3x3 ndarray, info)
        pybind/bindings/estimators/homography.cc:15-38, 75-77
    p3p / relpose_5pt / essential_matrix_5pt / relpose_7pt / homography_4pt   pybind/bindings/solvers.cc:305-365
Option dicts: nested 'ransac' / 'bundle' plus 'max_error', 'real_focal_check' (pybind/helpers.h:31-173);
`info` holds the RansacStats fields and 'inliers' as list[bool] (helpers.h:247-253, 266-272).
Passing an initial model sets ransac.score_initial_model (absolute_pose.cc(pybind):24-27).
All numerical work happens in the HIP library; this file only marshals numpy arrays through the C-ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

LAMBDA_UPDATES = {"NIELSEN": 0, "FIXED_FACTOR": 1}
DAMPINGS = {"LEVENBERG": 0, "MARQUARDT": 1}
LOSS_TYPES = {"TRIVIAL": 0, "TRUNCATED": 1, "HUBER": 2, "CAUCHY": 3, "TRUNCATED_CAUCHY": 4, "TRUNCATED_LE_ZACH": 5}
CAMERA_MODEL_IDS = {"NULL": -1, "SIMPLE_PINHOLE": 0, "PINHOLE": 1, "OPENCV": 4}
_CAMERA_NAMES = {v: k for k, v in CAMERA_MODEL_IDS.items()}
KIND_ABS, KIND_REL, KIND_FUND, KIND_HOM = 0, 1, 2, 3
KIND_SHARED_FOCAL = 4  # pl_batch_item only: estimate_shared_focal_relative_pose


# ------------------------------------------------------------------------------------------ types
class CameraPose:
    """poselib.CameraPose: q (w,x,y,z), t  (pybind/bindings/types.cc:35-45)."""

    def __init__(self, q=None, t=None):
        self.q = np.array([1.0, 0.0, 0.0, 0.0]) if q is None else np.asarray(q, dtype=np.float64).copy()
        self.t = np.zeros(3) if t is None else np.asarray(t, dtype=np.float64).copy()

    @property
    def R(self):
        w, x, y, z = self.q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    @property
    def Rt(self):
        return np.concatenate([self.R, self.t[:, None]], axis=1)

    def center(self):
        return -self.R.T @ self.t

    def __repr__(self):
        return f"CameraPose(q={self.q.tolist()}, t={self.t.tolist()})"


class Camera:
    """poselib.Camera subset (pybind/bindings/types.cc:68-113)."""

    def __init__(self, model="SIMPLE_PINHOLE", params=(), width=0, height=0):
        if isinstance(model, dict):
            d = model
            model, params = d["model"], d["params"]
            width, height = d.get("width", 0), d.get("height", 0)
        self.model_id = CAMERA_MODEL_IDS[model] if isinstance(model, str) else int(model)
        self.params = [float(p) for p in params]
        self.width, self.height = int(width), int(height)

    def model_name(self):
        return _CAMERA_NAMES[self.model_id]

    def focal(self):
        if not self.params:
            return 1.0
        if self.model_id == 0:
            return self.params[0]
        if self.model_id in (1, 4):
            return 0.5 * self.params[0] + 0.5 * self.params[1]
        return 1.0

    def todict(self):
        return {"model": self.model_name(), "width": self.width, "height": self.height, "params": list(self.params)}

    def _c(self) -> L.Camera:
        c = L.Camera()
        c.model_id, c.width, c.height, c.num_params = self.model_id, self.width, self.height, len(self.params)
        for i, v in enumerate(self.params):
            c.params[i] = v
        return c


class Image:
    """poselib.Image {camera, pose} (pybind/bindings/types.cc:117-119)."""

    def __init__(self, pose=None, camera=None):
        self.pose = pose or CameraPose()
        self.camera = camera or Camera()


def RansacOptions():
    return {"max_iterations": 100000, "min_iterations": 1000, "dyn_num_trials_mult": 3.0, "success_prob": 0.9999,
            "seed": 0, "progressive_sampling": False, "max_prosac_iterations": 100000}


def BundleOptions():
    return {"max_iterations": 100, "loss_type": "CAUCHY", "loss_scale": 1.0, "gradient_tol": 1e-12, "step_tol": 1e-8,
            "relative_cost_tol": 1e-10, "initial_lambda": 1e-3, "min_lambda": 1e-10, "max_lambda": 1e10,
            "lambda_factor": 10.0, "lambda_update": "NIELSEN", "damping": "LEVENBERG", "verbose": False}


# ------------------------------------------------------------------------------------------ marshalling
_RANSAC_KEYS = {"max_iterations", "min_iterations", "dyn_num_trials_mult", "success_prob", "seed",
                "progressive_sampling", "max_prosac_iterations", "score_initial_model"}
_BUNDLE_KEYS = {"max_iterations", "loss_type", "loss_scale", "gradient_tol", "step_tol", "relative_cost_tol",
                "initial_lambda", "min_lambda", "max_lambda", "lambda_factor", "lambda_update", "damping", "verbose",
                "refine_focal_length", "refine_extra_params", "refine_principal_point"}
_TOP_KEYS = {"ransac", "bundle", "max_error", "real_focal_check", "tangent_sampson", "estimate_focal_length",
             "estimate_extra_params", "min_fov"}


def _warn_unknown(d, known, where):
    # the reference's pybind ignores unknown keys silently (helpers.h:21-29); say so instead
    extra = sorted(set(d) - known)
    if extra:
        import warnings

        warnings.warn(f"poselib_amd: unknown {where} option(s) {extra} ignored", stacklevel=4)


def _robust_options(opt, kind: int, score_initial: bool) -> L.RobustOptions:
    o = L.RobustOptions()
    L.lib().pl_default_robust_options(C.byref(o), kind)
    opt = opt or {}
    _warn_unknown(opt, _TOP_KEYS, "top-level")
    _warn_unknown(opt.get("ransac", {}), _RANSAC_KEYS, "'ransac'")
    _warn_unknown(opt.get("bundle", {}), _BUNDLE_KEYS, "'bundle'")
    r = opt.get("ransac", {})
    for k in ("max_iterations", "min_iterations", "seed", "max_prosac_iterations"):
        if k in r:
            setattr(o.ransac, k, int(r[k]))
    for k in ("dyn_num_trials_mult", "success_prob"):
        if k in r:
            setattr(o.ransac, k, float(r[k]))
    if "progressive_sampling" in r:
        o.ransac.progressive_sampling = int(bool(r["progressive_sampling"]))
    o.ransac.score_initial_model = int(score_initial)
    b = opt.get("bundle", {})
    if "max_iterations" in b:
        o.bundle.max_iterations = int(b["max_iterations"])
    for k in ("loss_scale", "gradient_tol", "step_tol", "relative_cost_tol", "initial_lambda", "min_lambda",
              "max_lambda", "lambda_factor"):
        if k in b:
            setattr(o.bundle, k, float(b[k]))
    # enumerations: case-insensitive names like the reference's pybind (helpers.h:54-92), or the integer value
    for key, table in (("loss_type", LOSS_TYPES), ("lambda_update", LAMBDA_UPDATES), ("damping", DAMPINGS)):
        if key in b:
            v = b[key]
            setattr(o.bundle, key, table[v.upper()] if isinstance(v, str) else int(v))
    for k in ("refine_focal_length", "refine_extra_params", "refine_principal_point"):
        if k in b:
            setattr(o.bundle, k, int(bool(b[k])))
    if "max_error" in opt:
        o.max_error = float(opt["max_error"])
    for k in ("real_focal_check", "tangent_sampson", "estimate_focal_length", "estimate_extra_params"):
        if k in opt:
            setattr(o, k, int(bool(opt[k])))
    if "min_fov" in opt:  # types.h:126; read by the focal-length estimator only (absolute_pose.h:78)
        o.min_fov = float(opt["min_fov"])
    return o


def _pts(a, dim):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != dim:
        raise ValueError(f"expected an (N, {dim}) array")
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _info(st: L.RansacStats, inliers: np.ndarray):
    return {"refinements": st.refinements, "iterations": st.iterations, "num_inliers": st.num_inliers,
            "inlier_ratio": st.inlier_ratio, "model_score": st.model_score, "inliers": inliers.astype(bool).tolist(),
            # extras (not in the reference): metric numerator and device timing
            "hypotheses": st.hypotheses, "iterations_evaluated": st.iterations_evaluated, "seconds": st.seconds,
            "score_kernel_ms": st.score_kernel_ms, "score_kernel_launches": st.score_kernel_launches,
            "nan_hypotheses": st.nan_hypotheses}


def _cpose(p: CameraPose) -> L.CameraPose:
    c = L.CameraPose()
    for i in range(4):
        c.q[i] = p.q[i]
    for i in range(3):
        c.t[i] = p.t[i]
    return c


def _pypose(c: L.CameraPose) -> CameraPose:
    return CameraPose(np.array(c.q[:]), np.array(c.t[:]))


def _as_camera(cam) -> Camera:
    return cam if isinstance(cam, Camera) else Camera(cam)


# ------------------------------------------------------------------------------------------ estimators
def estimate_absolute_pose(points2D, points3D, camera, opt=None, initial_pose=None):
    p2, p3 = _pts(points2D, 2), _pts(points3D, 3)
    cam = _as_camera(camera)
    o = _robust_options(opt, KIND_ABS, initial_pose is not None)
    c = cam._c()
    pose = _cpose(initial_pose if initial_pose is not None else CameraPose())
    n = p2.shape[0]
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.RansacStats()
    L.check(L.lib().pl_estimate_absolute_pose(_ptr(p2), _ptr(p3), C.c_size_t(n), C.byref(o), C.byref(c), C.byref(pose),
                                              _ptr(inl), C.byref(st)))
    out_cam = Camera(cam.model_id, list(c.params[: c.num_params]), cam.width, cam.height)
    return Image(_pypose(pose), out_cam), _info(st, inl[:n])


def estimate_relative_pose(points2D_1, points2D_2, camera1, camera2, opt=None, initial_pose=None):
    a, b = _pts(points2D_1, 2), _pts(points2D_2, 2)
    c1, c2 = _as_camera(camera1)._c(), _as_camera(camera2)._c()
    o = _robust_options(opt, KIND_REL, initial_pose is not None)
    pose = _cpose(initial_pose if initial_pose is not None else CameraPose())
    n = a.shape[0]
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.RansacStats()
    L.check(L.lib().pl_estimate_relative_pose(_ptr(a), _ptr(b), C.c_size_t(n), C.byref(c1), C.byref(c2), C.byref(o),
                                              C.byref(pose), _ptr(inl), C.byref(st)))
    return _pypose(pose), _info(st, inl[:n])


def _estimate_matrix(fn, kind, points2D_1, points2D_2, opt, initial):
    a, b = _pts(points2D_1, 2), _pts(points2D_2, 2)
    o = _robust_options(opt, kind, initial is not None)
    M = np.ascontiguousarray((np.eye(3) if initial is None else np.asarray(initial, dtype=np.float64)).T.reshape(9))
    n = a.shape[0]
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.RansacStats()
    L.check(getattr(L.lib(), fn)(_ptr(a), _ptr(b), C.c_size_t(n), C.byref(o), _ptr(M), _ptr(inl), C.byref(st)))
    return M.reshape(3, 3).T.copy(), _info(st, inl[:n])


class Batch:
    """An array of pl_batch_item descriptors (include/poselib_amd.h) marshalled once: `run()` is the C-ABI call
    pl_estimate_batch and nothing else (what bench_batch.py times), `results()` turns the outputs into the objects the
    single-problem functions return."""

    def __init__(self, problems):
        kinds = {"abs": KIND_ABS, "rel": KIND_REL, "fund": KIND_FUND, "hom": KIND_HOM, "shared_focal": KIND_SHARED_FOCAL}
        self.items = (L.BatchItem * len(problems))()
        self.keep = []
        for it, pr in zip(self.items, problems):
            kind = kinds[pr[0]]
            it.kind = kind
            # a warm start: the options dict may carry "initial_model" (a CameraPose for "abs" / "rel", a 3 x 3 matrix for "fund" /
            # "hom") - the initial_pose / initial_F / initial_H argument of the single-problem functions; it sets score_initial_model
            opt = dict(pr[-1] or {})
            init = opt.pop("initial_model", None)
            pr = tuple(pr[:-1]) + (opt,)
            if kind == KIND_ABS:
                _, a, b, cam, opt = pr
                a, b = _pts(a, 2), _pts(b, 3)
                cam = _as_camera(cam)
                c1, c2 = cam._c(), None
                model = _cpose(init if init is not None else CameraPose())
            elif kind == KIND_REL:
                _, a, b, cam1, cam2, opt = pr
                a, b = _pts(a, 2), _pts(b, 2)
                cam = None
                c1, c2 = _as_camera(cam1)._c(), _as_camera(cam2)._c()
                model = _cpose(init if init is not None else CameraPose())
            elif kind == KIND_SHARED_FOCAL:
                _, a, b, pp, opt = pr
                a, b = _pts(a, 2), _pts(b, 2)
                cam = Camera("SIMPLE_PINHOLE", [1.0, float(pp[0]), float(pp[1])])
                c1, c2 = cam._c(), None
                model = _cpose(CameraPose())
            else:
                _, a, b, opt = pr
                a, b = _pts(a, 2), _pts(b, 2)
                cam, c1, c2 = None, None, None
                model = np.ascontiguousarray((np.eye(3) if init is None else np.asarray(init, dtype=np.float64)).T.reshape(9))
            o = _robust_options(opt, KIND_REL if kind == KIND_SHARED_FOCAL else kind, init is not None and kind != KIND_SHARED_FOCAL)
            n = a.shape[0]
            inl = np.zeros(max(n, 1), dtype=np.uint8)
            st = L.RansacStats()
            it.a, it.b, it.n = _ptr(a), _ptr(b), n
            it.opt = C.pointer(o)
            it.camera1 = C.pointer(c1) if c1 is not None else None
            it.camera2 = C.pointer(c2) if c2 is not None else None
            it.model = C.cast(C.pointer(model), C.c_void_p) if kind in (KIND_ABS, KIND_REL, KIND_SHARED_FOCAL) else _ptr(model)
            it.inliers = _ptr(inl)
            it.stats = C.pointer(st)
            self.keep.append((kind, a, b, o, cam, c1, c2, model, inl, st, n))

    def run(self, max_in_flight=8, devices=None):
        """devices: None = the calling thread's device (pl_estimate_batch); a list of device indices, or "all", = the items
        round-robined over those devices from this one process (pl_estimate_batch_devices)"""
        if devices is None:
            L.check(L.lib().pl_estimate_batch(self.items, C.c_size_t(len(self.keep)), C.c_int(int(max_in_flight))))
            return
        lst = [] if isinstance(devices, str) else [int(d) for d in devices]
        arr = (C.c_int * max(len(lst), 1))(*lst)
        L.check(L.lib().pl_estimate_batch_devices(self.items, C.c_size_t(len(self.keep)), arr if lst else None, C.c_int(len(lst)),
                                                  C.c_int(int(max_in_flight))))

    def stats(self):
        """(iterations, num_inliers, hypotheses) arrays without building the per-problem Python objects"""
        it = np.array([k[9].iterations for k in self.keep], dtype=np.int64)
        ni = np.array([k[9].num_inliers for k in self.keep], dtype=np.int64)
        hy = np.array([k[9].hypotheses for k in self.keep], dtype=np.int64)
        return it, ni, hy

    def results(self):
        out = []
        for kind, a, b, o, cam, c1, c2, model, inl, st, n in self.keep:
            info = _info(st, inl[:n])
            if kind == KIND_ABS:
                out_cam = Camera(cam.model_id, list(c1.params[: c1.num_params]), cam.width, cam.height)
                out.append((Image(_pypose(model), out_cam), info))
            elif kind == KIND_REL:
                out.append((_pypose(model), info))
            elif kind == KIND_SHARED_FOCAL:
                out.append((_shared_focal_pair(model, c1.params[0], (c1.params[1], c1.params[2])), info))
            else:
                out.append((model.reshape(3, 3).T.copy(), info))
        return out


def last_batch_report():
    """What this thread's last batch call did with its items (include/poselib_amd.h pl_batch_report): items in lock-step groups run
    at the advertised rate, `solo` and `fallback` items one at a time."""
    r = L.BatchReport()
    L.lib().pl_last_batch_report(C.byref(r))
    return {k: int(getattr(r, k)) for k in ("items", "grouped", "focal_grouped", "solo", "fallback")}


def estimate_batch(problems, max_in_flight=8, devices=None):
    """Many independent problems in one call (include/poselib_amd.h pl_estimate_batch; BASELINE config 4).

    problems: list of tuples, the arguments of the single-problem functions prefixed by the kind:
        ("abs", points2D, points3D, camera, opt)            -> (Image, info)
        ("rel", points2D_1, points2D_2, camera1, camera2, opt) -> (CameraPose, info)
        ("fund", points2D_1, points2D_2, opt) / ("hom", points2D_1, points2D_2, opt) -> (3x3 ndarray, info)
        ("shared_focal", points2D_1, points2D_2, pp, opt)   -> (ImagePair, info)
    Returns the list of results in the same order.  Problems of the same kind advance in groups through ONE launch
    sequence (the problem index is a grid dimension of every kernel); `max_in_flight` host threads inside the library
    work on groups concurrently, each with its own HIP stream.  devices: a list of device indices (or "all") spreads the problems
    over several GPUs from this one process (problem i on devices[i mod len]); None: the calling thread's device."""
    b = Batch(problems)
    b.run(max_in_flight, devices)
    return b.results()


class RansacBatch:
    """pl_ransac_item descriptors for device-resident problems (`Problem`), marshalled once: `run()` is the C-ABI call
    pl_ransac_batch and nothing else (what bench.py times), `results()` returns what `Problem.run` returns per item."""

    def __init__(self, problems, opts):
        assert len(problems) == len(opts)
        self.items = (L.RansacItem * len(problems))()
        self.keep = []
        for it, pr, opt in zip(self.items, problems, opts):
            o = _robust_options(opt, pr.kind, False)
            inl = np.zeros(max(pr.n, 1), dtype=np.uint8)
            st = L.RansacStats()
            if pr.kind in (KIND_ABS, KIND_REL):
                model = _cpose(CameraPose())
                it.model = C.cast(C.pointer(model), C.c_void_p)
            else:
                model = np.ascontiguousarray(np.eye(3).reshape(9))
                it.model = _ptr(model)
            it.problem = pr._h
            it.opt = C.pointer(o)
            it.inliers = _ptr(inl)
            it.stats = C.pointer(st)
            self.keep.append((pr, o, model, inl, st))

    def run(self, max_in_flight=4, group_size=16):
        L.check(L.lib().pl_ransac_batch(self.items, C.c_size_t(len(self.items)), int(max_in_flight), int(group_size)))

    def stats(self):
        return [k[4] for k in self.keep]

    def results(self, first=None):
        """what `Problem.run` returns, per item (`first`: only the first so many items)"""
        out = []
        for pr, o, model, inl, st in (self.keep if first is None else self.keep[:first]):
            if pr.kind in (KIND_ABS, KIND_REL):
                out.append((_pypose(model), _info(st, inl[: pr.n])))
            else:
                out.append((model.reshape(3, 3).T.copy(), _info(st, inl[: pr.n])))
        return out


def device_math(fn: int, x):
    """Diagnostic (pl_debug_device_math): the device kernels' scalar math on an array - 0: cube of the LM's Nielsen update,
    1: sqrt, 2: reciprocal, 3: cbrt, 4: cos, 5: sin, 6: acos."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    L.check(L.lib().pl_debug_device_math(int(fn), _ptr(x), C.c_size_t(x.size), _ptr(out)))
    return out


def ransac_batch(problems, opts, max_in_flight=4, group_size=16):
    """Many device-resident problems in one call (pl_ransac_batch): results of `problems[i].run(opts[i])`, bit for bit.
    Problems of the same kind advance in lock-step groups of `group_size` through one launch sequence."""
    b = RansacBatch(problems, opts)
    b.run(max_in_flight, group_size)
    return b.results()


def estimate_fundamental(points2D_1, points2D_2, opt=None, initial_F=None):
    return _estimate_matrix("pl_estimate_fundamental", KIND_FUND, points2D_1, points2D_2, opt, initial_F)


def estimate_homography(points2D_1, points2D_2, opt=None, initial_H=None):
    return _estimate_matrix("pl_estimate_homography", KIND_HOM, points2D_1, points2D_2, opt, initial_H)


# ------------------------------------------------------------------------------------------ ransac_* on normalised points
def _ransac(fn, kind, a, b, dim_b, opt, initial):
    a, b = _pts(a, 2), _pts(b, dim_b)
    o = _robust_options(opt, kind, initial is not None)
    n = a.shape[0]
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.RansacStats()
    if kind in (KIND_ABS, KIND_REL):
        pose = _cpose(initial if initial is not None else CameraPose())
        L.check(getattr(L.lib(), fn)(_ptr(a), _ptr(b), C.c_size_t(n), C.byref(o), C.byref(pose), _ptr(inl), C.byref(st)))
        return _pypose(pose), _info(st, inl[:n])
    M = np.ascontiguousarray((np.eye(3) if initial is None else np.asarray(initial, dtype=np.float64)).T.reshape(9))
    L.check(getattr(L.lib(), fn)(_ptr(a), _ptr(b), C.c_size_t(n), C.byref(o), _ptr(M), _ptr(inl), C.byref(st)))
    return M.reshape(3, 3).T.copy(), _info(st, inl[:n])


def ransac_pnp(x, X, opt=None, initial_pose=None):
    return _ransac("pl_ransac_pnp", KIND_ABS, x, X, 3, opt, initial_pose)


def ransac_pnpf(x, X, opt=None):
    """robust/ransac.h ransac_pnpf: pose and focal length of a SIMPLE_PINHOLE camera whose principal point is the origin of `x`.
    Returns (Image(pose, Camera SIMPLE_PINHOLE [f, 0, 0]), info)."""
    x, X = _pts(x, 2), _pts(X, 3)
    o = _robust_options(opt, KIND_ABS, False)
    n = x.shape[0]
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.RansacStats()
    pose = _cpose(CameraPose())
    focal = C.c_double(1.0)
    L.check(L.lib().pl_ransac_pnpf(_ptr(x), _ptr(X), C.c_size_t(n), C.byref(o), C.byref(pose), C.byref(focal), _ptr(inl), C.byref(st)))
    return Image(_pypose(pose), Camera("SIMPLE_PINHOLE", [focal.value, 0.0, 0.0])), _info(st, inl[:n])


class ImagePair:
    """poselib.ImagePair {pose, camera1, camera2} (types.h / pybind types.cc): here always two copies of one SIMPLE_PINHOLE camera"""

    def __init__(self, pose=None, camera1=None, camera2=None):
        self.pose = pose or CameraPose()
        self.camera1 = camera1 or Camera("SIMPLE_PINHOLE", [1.0, 0.0, 0.0])
        self.camera2 = camera2 or Camera("SIMPLE_PINHOLE", list(self.camera1.params))


def _shared_focal_pair(pose, focal, pp):
    return ImagePair(_pypose(pose), Camera("SIMPLE_PINHOLE", [focal, pp[0], pp[1]]), Camera("SIMPLE_PINHOLE", [focal, pp[0], pp[1]]))


def ransac_shared_focal_relpose(x1, x2, opt=None, initial_pair=None):
    """robust/ransac.h:71-73 ransac_shared_focal_relpose: relative pose and the focal length both views share; x1, x2 relative
    to the principal point.  Returns (ImagePair, info)."""
    x1, x2 = _pts(x1, 2), _pts(x2, 2)
    o = _robust_options(opt, KIND_REL, initial_pair is not None)
    n = x1.shape[0]
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.RansacStats()
    pose = _cpose(initial_pair.pose if initial_pair is not None else CameraPose())
    focal = C.c_double(initial_pair.camera1.focal() if initial_pair is not None else 1.0)
    L.check(L.lib().pl_ransac_shared_focal_relpose(_ptr(x1), _ptr(x2), C.c_size_t(n), C.byref(o), C.byref(pose), C.byref(focal),
                                                   _ptr(inl), C.byref(st)))
    return _shared_focal_pair(pose, focal.value, (0.0, 0.0)), _info(st, inl[:n])


def estimate_shared_focal_relative_pose(points2D_1, points2D_2, pp, opt=None, initial_pair=None):
    """robust.h:84-90 estimate_shared_focal_relative_pose: pixel coordinates, principal point pp.  Returns (ImagePair, info)."""
    x1, x2 = _pts(points2D_1, 2), _pts(points2D_2, 2)
    o = _robust_options(opt, KIND_REL, initial_pair is not None)
    n = x1.shape[0]
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.RansacStats()
    pose = _cpose(initial_pair.pose if initial_pair is not None else CameraPose())
    focal = C.c_double(initial_pair.camera1.focal() if initial_pair is not None else 1.0)
    ppa = np.ascontiguousarray(pp, dtype=np.float64)
    L.check(L.lib().pl_estimate_shared_focal_relative_pose(_ptr(x1), _ptr(x2), C.c_size_t(n), _ptr(ppa), C.byref(o), C.byref(pose),
                                                           C.byref(focal), _ptr(inl), C.byref(st)))
    return _shared_focal_pair(pose, focal.value, ppa), _info(st, inl[:n])


def refine_shared_focal_relpose(x1, x2, pair, bundle=None):
    """robust/bundle.h:108-111 refine_shared_focal_relpose on all correspondences (relative to the principal point).
    Returns (ImagePair, LM iterations)."""
    x1, x2 = _pts(x1, 2), _pts(x2, 2)
    o = _robust_options({"bundle": bundle or {}}, KIND_REL, False)
    pose = _cpose(pair.pose)
    focal = C.c_double(pair.camera1.focal())
    its = C.c_uint32(0)
    L.check(L.lib().pl_refine_shared_focal_relpose(_ptr(x1), _ptr(x2), C.c_size_t(x1.shape[0]), C.byref(o.bundle), C.byref(pose),
                                                   C.byref(focal), C.byref(its)))
    return _shared_focal_pair(pose, focal.value, (0.0, 0.0)), its.value


def solve_focal_batch(kind, inputs):
    """pl_solve_focal_batch: kind "p35pf" (inputs count x 20: x 4 x 2 | X 4 x 3) or "relpose_6pt_shared_focal" (count x 36: unit
    bearings x1 6 x 3 | x2 6 x 3).  Returns (models count x slots x 8 [q t focal], counts)."""
    k = {"p35pf": 0, "relpose_6pt_shared_focal": 1}[kind]
    a = np.ascontiguousarray(inputs, dtype=np.float64)
    per, slots = (20, 10) if k == 0 else (36, 60)
    if a.ndim != 2 or a.shape[1] != per:
        raise ValueError(f"expected a (count, {per}) array")
    models = np.zeros((a.shape[0], slots, 8))
    counts = np.zeros(max(a.shape[0], 1), dtype=np.uint32)
    L.check(L.lib().pl_solve_focal_batch(C.c_int(k), _ptr(a), C.c_size_t(a.shape[0]), _ptr(models), _ptr(counts)))
    return models, counts[: a.shape[0]]


def p35pf(x, X):
    """solvers/p35pf.h:39-54: four image points relative to the principal point and their 3-D points -> [(CameraPose, focal)]"""
    models, counts = solve_focal_batch("p35pf", np.r_[_pts(x, 2).reshape(-1), _pts(X, 3).reshape(-1)][None, :])
    return [(CameraPose(m[:4], m[4:7]), float(m[7])) for m in models[0, : counts[0]]]


def relpose_6pt_shared_focal(x1, x2):
    """solvers/relpose_6pt_focal.h:12-13: six pairs of unit bearings -> [(CameraPose, focal)] in the reference's order"""
    models, counts = solve_focal_batch("relpose_6pt_shared_focal", np.r_[_pts(x1, 3).reshape(-1), _pts(x2, 3).reshape(-1)][None, :])
    return [(CameraPose(m[:4], m[4:7]), float(m[7])) for m in models[0, : counts[0]]]


def ransac_relpose(x1, x2, opt=None, initial_pose=None):
    return _ransac("pl_ransac_relpose", KIND_REL, x1, x2, 2, opt, initial_pose)


def ransac_fundamental(x1, x2, opt=None, initial_F=None):
    return _ransac("pl_ransac_fundamental", KIND_FUND, x1, x2, 2, opt, initial_F)


def ransac_homography(x1, x2, opt=None, initial_H=None):
    return _ransac("pl_ransac_homography", KIND_HOM, x1, x2, 2, opt, initial_H)


# ------------------------------------------------------------------------------------------ device-resident problems
class Problem:
    """Correspondences uploaded once (SoA in HBM); `run` executes the LO-RANSAC loop on them.
    This is the entry bench.py times: inputs are resident before the timed region starts."""

    def __init__(self, kind: int, a, b):
        self.kind = kind
        a = _pts(a, 2)
        b = _pts(b, 3 if kind == KIND_ABS else 2)
        self.n = a.shape[0]
        self._h = C.c_void_p()
        L.check(L.lib().pl_problem_create(kind, _ptr(a), _ptr(b), C.c_size_t(self.n), C.byref(self._h)))

    def run(self, opt=None, initial=None):
        o = _robust_options(opt, self.kind, initial is not None)
        inl = np.zeros(max(self.n, 1), dtype=np.uint8)
        st = L.RansacStats()
        if self.kind in (KIND_ABS, KIND_REL):
            model = _cpose(initial if initial is not None else CameraPose())
            L.check(L.lib().pl_ransac_run(self._h, C.byref(o), C.byref(model), _ptr(inl), C.byref(st)))
            return _pypose(model), _info(st, inl[: self.n])
        M = np.ascontiguousarray((np.eye(3) if initial is None else np.asarray(initial, dtype=np.float64)).T.reshape(9))
        L.check(L.lib().pl_ransac_run(self._h, C.byref(o), _ptr(M), _ptr(inl), C.byref(st)))
        return M.reshape(3, 3).T.copy(), _info(st, inl[: self.n])

    def run_sharded(self, opt, rank: int, world: int, allgather, initial=None):
        """ONE problem across `world` GPUs (pl_ransac_run_sharded): every rank holds the same correspondences in its
        own Problem on its own device and calls this with the same options; each evaluates a contiguous share of every
        batch of iterations, and `allgather(send: bytes-like numpy uint8 array, recv: numpy uint8 array of world *
        len(send))` is the collective of the two exchange steps per batch (see poselib_amd.sharding.dist_allgather for the
        torch.distributed form).  All ranks return the same result - the single-device one."""
        o = _robust_options(opt, self.kind, initial is not None)
        inl = np.zeros(max(self.n, 1), dtype=np.uint8)
        st = L.RansacStats()
        errors = []

        def _cb(_user, send, recv, nbytes):
            try:
                s_arr = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                r_arr = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                allgather(s_arr, r_arr)
                return 0
            except Exception as e:  # surfaces as PL_ERR_COMM
                errors.append(e)
                return 1

        cb = L.ALLGATHER_FN(_cb)
        shard = L.Shard(rank, world, cb, None)
        if self.kind in (KIND_ABS, KIND_REL):
            model = _cpose(initial if initial is not None else CameraPose())
            rc = L.lib().pl_ransac_run_sharded(self._h, C.byref(o), C.byref(shard), C.byref(model), _ptr(inl), C.byref(st))
            if rc and errors:
                raise errors[0]
            L.check(rc)
            return _pypose(model), _info(st, inl[: self.n])
        M = np.ascontiguousarray((np.eye(3) if initial is None else np.asarray(initial, dtype=np.float64)).T.reshape(9))
        rc = L.lib().pl_ransac_run_sharded(self._h, C.byref(o), C.byref(shard), _ptr(M), _ptr(inl), C.byref(st))
        if rc and errors:
            raise errors[0]
        L.check(rc)
        return M.reshape(3, 3).T.copy(), _info(st, inl[: self.n])

    def score(self, model, max_error):
        """MSAC score + inlier count of one model (the estimators' score_model())."""
        cnt = C.c_uint64(0)
        sc = C.c_double(0.0)
        if self.kind in (KIND_ABS, KIND_REL):
            m = _cpose(model)
            L.check(L.lib().pl_score_model(self._h, C.byref(m), C.c_double(max_error), C.byref(cnt), C.byref(sc)))
        else:
            M = np.ascontiguousarray(np.asarray(model, dtype=np.float64).T.reshape(9))
            L.check(L.lib().pl_score_model(self._h, _ptr(M), C.c_double(max_error), C.byref(cnt), C.byref(sc)))
        return sc.value, cnt.value

    def score_stream(self, models, max_error):
        """Diagnostic (pl_debug_score_stream): a list of models through the streaming scorer of the main loop, i.e.
        through the conservative pre-filter in front of the exact evaluation.  models: (n, 7) array of q, t for pose
        problems, (n, 3, 3) matrices otherwise.  Returns (counts, scores, path) with path 2 = matrix-core filter,
        1 = fp32 filter, 0 = no filter."""
        m = np.ascontiguousarray(models, dtype=np.float64)
        n = m.shape[0]
        if self.kind in (KIND_ABS, KIND_REL):
            assert m.shape == (n, 7)
            flat = np.ascontiguousarray(m.reshape(-1))  # pl_camera_pose = 7 packed doubles
        else:
            assert m.shape == (n, 3, 3)
            flat = np.ascontiguousarray(np.transpose(m, (0, 2, 1)).reshape(-1))  # column-major
        cnt = np.zeros(max(n, 1), dtype=np.uint32)
        sc = np.zeros(max(n, 1), dtype=np.float64)
        path = C.c_int32(0)
        L.check(L.lib().pl_debug_score_stream(self._h, _ptr(flat), C.c_size_t(n), C.c_double(max_error), _ptr(cnt),
                                              _ptr(sc), C.byref(path)))
        return cnt[:n], sc[:n], path.value

    def refine(self, model, bundle_opt=None, camera=None, mask=None):
        """bundle_adjust / refine_relpose / refine_fundamental / refine_homography on the resident points."""
        o = _robust_options({"bundle": bundle_opt or {}}, self.kind, False).bundle
        cam = None if camera is None else _as_camera(camera)._c()
        it = C.c_uint32(0)
        m8 = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        args_tail = (C.byref(cam) if cam is not None else None, None if m8 is None else _ptr(m8))
        if self.kind in (KIND_ABS, KIND_REL):
            m = _cpose(model)
            L.check(L.lib().pl_refine_model(self._h, C.byref(o), *args_tail, C.byref(m), C.byref(it)))
            return _pypose(m), it.value
        M = np.ascontiguousarray(np.asarray(model, dtype=np.float64).T.reshape(9))
        L.check(L.lib().pl_refine_model(self._h, C.byref(o), *args_tail, _ptr(M), C.byref(it)))
        return M.reshape(3, 3).T.copy(), it.value

    def bundle_adjust(self, pose, camera, bundle_opt=None, mask=None):
        """poselib.bundle_adjust(points2D, points3D, camera, pose, opt) on the resident points (2-D points in pixels):
        refines the pose and - with refine_focal_length / refine_principal_point / refine_extra_params - the camera's
        intrinsics (robust/bundle.cc:93-118).  Returns (pose, camera, LM iterations)."""
        assert self.kind == KIND_ABS
        o = _robust_options({"bundle": bundle_opt or {}}, self.kind, False).bundle
        cam = _as_camera(camera)
        c = cam._c()
        it = C.c_uint32(0)
        m8 = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        m = _cpose(pose)
        L.check(L.lib().pl_bundle_adjust_camera(self._h, C.byref(o), C.byref(c), None if m8 is None else _ptr(m8), C.byref(m),
                                                C.byref(it)))
        return _pypose(m), Camera(cam.model_id, list(c.params[: c.num_params]), cam.width, cam.height), it.value

    def close(self):
        if self._h:
            L.lib().pl_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------ minimal solvers
def _bearings(x, k):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.shape != (k, 3):
        raise ValueError(f"expected {k} bearing vectors of dimension 3")
    return x


def p3p(x, X):
    x, X = _bearings(x, 3), _bearings(X, 3)
    out = (L.CameraPose * 4)()
    n = L.check(L.lib().pl_p3p(_ptr(x), _ptr(X), out))
    return [_pypose(out[i]) for i in range(n)]


def relpose_5pt(x1, x2):
    x1, x2 = _bearings(x1, 5), _bearings(x2, 5)
    out = (L.CameraPose * 40)()
    n = L.check(L.lib().pl_relpose_5pt(_ptr(x1), _ptr(x2), out))
    return [_pypose(out[i]) for i in range(n)]


p5p = relpose_5pt  # alias named by the task statement; the reference exposes relpose_5pt / essential_matrix_5pt


def essential_matrix_5pt(x1, x2):
    x1, x2 = _bearings(x1, 5), _bearings(x2, 5)
    out = np.zeros((10, 9))
    n = L.check(L.lib().pl_essential_matrix_5pt(_ptr(x1), _ptr(x2), _ptr(out)))
    return [out[i].reshape(3, 3).T.copy() for i in range(n)]


def relpose_7pt(x1, x2):
    x1, x2 = _bearings(x1, 7), _bearings(x2, 7)
    out = np.zeros((3, 9))
    n = L.check(L.lib().pl_relpose_7pt(_ptr(x1), _ptr(x2), _ptr(out)))
    return [out[i].reshape(3, 3).T.copy() for i in range(n)]


def homography_4pt(x1, x2):
    x1, x2 = _bearings(x1, 4), _bearings(x2, 4)
    out = np.zeros(9)
    n = L.check(L.lib().pl_homography_4pt(_ptr(x1), _ptr(x2), _ptr(out)))
    return [out.reshape(3, 3).T.copy()] if n else []


def solve_batch(kind: int, first, second):
    """Many minimal problems at once, one GPU lane each.  first/second: (B, K, 3).  Returns
    (records (B, max_models, 16), counts (B,)) — record layout: q[4] t[3] M[9 row-major]."""
    first = np.ascontiguousarray(first, dtype=np.float64)
    second = np.ascontiguousarray(second, dtype=np.float64)
    B, K = first.shape[0], first.shape[1]
    maxm = {0: 4, 1: 40, 2: 3, 3: 1}[kind]
    inp = np.ascontiguousarray(np.concatenate([first.reshape(B, -1), second.reshape(B, -1)], axis=1))
    out = np.zeros((B, maxm, 24))  # record pitch: 16 fp64 fields + fp32 shadow used by the scoring pre-filter
    cnt = np.zeros(B, dtype=np.uint32)
    L.check(L.lib().pl_solve_batch(kind, _ptr(inp), C.c_size_t(B), _ptr(out), _ptr(cnt)))
    return out[:, :, :16].copy(), cnt


def undistort_points(camera, points2D):
    """Pixels of a distorting camera -> pixels of the distortion-free camera with the same focal lengths and principal
    point (pl_undistort_points: Camera::unproject per point on the device, then fx u + cx, fy v + cy).  The stage in
    front of estimate_homography / estimate_fundamental, which take no camera."""
    a = _pts(points2D, 2)
    out = np.zeros_like(a)
    c = _as_camera(camera)._c()
    L.check(L.lib().pl_undistort_points(C.byref(c), _ptr(a), C.c_size_t(a.shape[0]), _ptr(out)))
    return out


def device_count() -> int:
    return L.lib().pl_device_count()


def set_device(i: int):
    L.check(L.lib().pl_set_device(int(i)))


def set_lm_mode(mode) -> int:
    """Summation order of the non-linear refinements.  0 / False (default): the reference's order up to 256 correspondences and
    tree order beyond for poses and homographies, the reference's order at every size for fundamental matrices (the sign of a
    refined F hangs on the bits of the refinement's input); 1 / True: the reference's order at every size for every estimator
    (bit-identical refined models, slower); 2: tree order beyond 256 correspondences for every estimator.  Returns the previous mode."""
    return int(L.lib().pl_set_lm_mode(int(mode)))
