"""ctypes binding of poselib_amd/lib/libposelib_amd.so (the C-ABI in include/poselib_amd.h).

The shared library is the product: hand-written HIP kernels for gfx950 plus the host driver.
There is no Python/CPU implementation behind this module — if the library is missing or no HIP
device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "lib", "libposelib_amd.so")
if os.environ.get("POSELIB_AMD_LIB"):  # development: another build of the same C-ABI (A/B timing of kernel variants)
    LIB_PATH = os.environ["POSELIB_AMD_LIB"]

u64, i32, u32, f64 = C.c_uint64, C.c_int32, C.c_uint32, C.c_double

PL_OK, PL_ERR_NO_DEVICE, PL_ERR_HIP, PL_ERR_INVALID, PL_ERR_UNSUPPORTED, PL_ERR_COMM = 0, -1, -2, -3, -4, -5


class RansacOptions(C.Structure):
    _fields_ = [("max_iterations", u64), ("min_iterations", u64), ("dyn_num_trials_mult", f64), ("success_prob", f64),
                ("seed", u64), ("progressive_sampling", i32), ("score_initial_model", i32),
                ("max_prosac_iterations", u64)]


class BundleOptions(C.Structure):
    _fields_ = [("max_iterations", u64), ("loss_type", i32), ("lambda_update", i32), ("damping", i32), ("verbose", i32),
                ("loss_scale", f64), ("gradient_tol", f64), ("step_tol", f64), ("relative_cost_tol", f64),
                ("initial_lambda", f64), ("min_lambda", f64), ("max_lambda", f64), ("lambda_factor", f64),
                ("refine_focal_length", i32), ("refine_extra_params", i32), ("refine_principal_point", i32),
                ("reserved", i32)]


class RobustOptions(C.Structure):
    _fields_ = [("ransac", RansacOptions), ("bundle", BundleOptions), ("max_error", f64), ("real_focal_check", i32),
                ("tangent_sampson", i32), ("estimate_focal_length", i32), ("estimate_extra_params", i32), ("min_fov", f64)]


class RansacStats(C.Structure):
    _fields_ = [("refinements", u64), ("iterations", u64), ("num_inliers", u64), ("inlier_ratio", f64),
                ("model_score", f64), ("hypotheses", u64), ("iterations_evaluated", u64), ("seconds", f64),
                ("score_kernel_ms", f64), ("score_kernel_launches", u32), ("nan_hypotheses", u32)]


class BatchItem(C.Structure):
    pass  # fields set below (needs Camera / RobustOptions)


class Camera(C.Structure):
    _fields_ = [("model_id", i32), ("width", i32), ("height", i32), ("num_params", i32), ("params", f64 * 12)]


class CameraPose(C.Structure):
    _fields_ = [("q", f64 * 4), ("t", f64 * 3)]


BatchItem._fields_ = [("kind", i32), ("status", i32), ("a", C.c_void_p), ("b", C.c_void_p), ("n", C.c_size_t),
                      ("opt", C.POINTER(RobustOptions)), ("camera1", C.POINTER(Camera)), ("camera2", C.POINTER(Camera)),
                      ("model", C.c_void_p), ("inliers", C.c_void_p), ("stats", C.POINTER(RansacStats))]


class BatchReport(C.Structure):  # pl_batch_report
    _fields_ = [("items", C.c_uint64), ("grouped", C.c_uint64), ("focal_grouped", C.c_uint64), ("solo", C.c_uint64),
                ("fallback", C.c_uint64)]


class RansacItem(C.Structure):  # pl_ransac_item
    _fields_ = [("problem", C.c_void_p), ("opt", C.POINTER(RobustOptions)), ("model", C.c_void_p), ("inliers", C.c_void_p),
                ("stats", C.POINTER(RansacStats)), ("status", i32), ("reserved", i32)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class Shard(C.Structure):  # pl_shard
    _fields_ = [("rank", i32), ("world", i32), ("allgather", ALLGATHER_FN), ("user", C.c_void_p)]


class PoseLibAmdError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".hip", ".cc", ".inc", "Makefile"))]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "poselib_amd.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        cmd = ["make", "-C", _CSRC]
        if not verbose:
            cmd.append("-s")
        subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """Load the C-ABI library.  Raises if it has not been built (use poselib_amd.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PoseLibAmdError(
                f"{LIB_PATH} is missing: build it with `python -c 'import poselib_amd; poselib_amd.build()'` "
                "(there is no CPU fallback)")
        # The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default; streams that share
        # a queue run one after the other) and reads the variable at its first call.  The batch entry points keep 8 - 16
        # streams in flight, so this binding - the application layer of a Python process - asks for 16 unless the user
        # chose a value or opted out (POSELIB_AMD_KEEP_ENV=1).  The C library itself never touches the environment.
        if not os.environ.get("POSELIB_AMD_KEEP_ENV"):
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        L = C.CDLL(LIB_PATH)
        _declare(L)
        _lib = L
    return _lib


def _declare(L):
    """argtypes / restype of every C-ABI function (include/poselib_amd.h), so that a bare Python int or None is
    converted to the declared width instead of ctypes' default c_int."""
    vp, sz, dbl, cint = C.c_void_p, C.c_size_t, C.c_double, C.c_int
    P = C.POINTER
    opt, stats, cam, pose = P(RobustOptions), P(RansacStats), P(Camera), P(CameraPose)
    sig = {
        "pl_default_ransac_options": (None, [P(RansacOptions)]),
        "pl_default_bundle_options": (None, [P(BundleOptions)]),
        "pl_default_robust_options": (None, [opt, cint]),
        "pl_device_count": (cint, []),
        "pl_set_lm_mode": (cint, [cint]),
        "pl_set_device": (cint, [cint]),
        "pl_last_error": (C.c_char_p, []),
        "pl_version": (C.c_char_p, []),
        "pl_abi_version": (cint, []),
        "pl_estimate_absolute_pose": (cint, [vp, vp, sz, opt, cam, pose, vp, stats]),
        "pl_estimate_relative_pose": (cint, [vp, vp, sz, cam, cam, opt, pose, vp, stats]),
        "pl_estimate_fundamental": (cint, [vp, vp, sz, opt, vp, vp, stats]),
        "pl_estimate_homography": (cint, [vp, vp, sz, opt, vp, vp, stats]),
        "pl_estimate_batch": (cint, [P(BatchItem), sz, cint]),
        "pl_estimate_batch_devices": (cint, [P(BatchItem), sz, P(C.c_int), cint, cint]),
        "pl_last_batch_report": (None, [P(BatchReport)]),
        "pl_ransac_batch": (cint, [P(RansacItem), sz, cint, cint]),
        "pl_undistort_points": (cint, [cam, vp, sz, vp]),
        "pl_ransac_pnp": (cint, [vp, vp, sz, opt, pose, vp, stats]),
        "pl_ransac_pnpf": (cint, [vp, vp, sz, opt, pose, P(dbl), vp, stats]),
        "pl_ransac_relpose": (cint, [vp, vp, sz, opt, pose, vp, stats]),
        "pl_ransac_shared_focal_relpose": (cint, [vp, vp, sz, opt, pose, P(dbl), vp, stats]),
        "pl_solve_focal_batch": (cint, [cint, vp, sz, vp, vp]),
        "pl_p35pf": (cint, [vp, vp, vp, vp]),
        "pl_relpose_6pt_shared_focal": (cint, [vp, vp, vp, vp]),
        "pl_refine_shared_focal_relpose": (cint, [vp, vp, sz, P(BundleOptions), pose, P(dbl), P(C.c_uint32)]),
        "pl_estimate_shared_focal_relative_pose": (cint, [vp, vp, sz, vp, opt, pose, P(dbl), vp, stats]),
        "pl_ransac_fundamental": (cint, [vp, vp, sz, opt, vp, vp, stats]),
        "pl_ransac_homography": (cint, [vp, vp, sz, opt, vp, vp, stats]),
        "pl_problem_create": (cint, [cint, vp, vp, sz, P(vp)]),
        "pl_problem_destroy": (None, [vp]),
        "pl_ransac_run": (cint, [vp, opt, vp, vp, stats]),
        "pl_ransac_run_sharded": (cint, [vp, opt, P(Shard), vp, vp, stats]),
        "pl_score_model": (cint, [vp, vp, dbl, P(C.c_uint64), P(dbl)]),
        "pl_debug_score_stream": (cint, [vp, vp, sz, dbl, vp, vp, P(C.c_int32)]),
        "pl_debug_device_math": (cint, [cint, vp, sz, vp]),
        "pl_refine_model": (cint, [vp, P(BundleOptions), cam, vp, vp, P(C.c_uint32)]),
        "pl_bundle_adjust_camera": (cint, [vp, P(BundleOptions), cam, vp, P(CameraPose), P(C.c_uint32)]),
        "pl_p3p": (cint, [vp, vp, P(CameraPose)]),
        "pl_relpose_5pt": (cint, [vp, vp, P(CameraPose)]),
        "pl_essential_matrix_5pt": (cint, [vp, vp, vp]),
        "pl_relpose_7pt": (cint, [vp, vp, vp]),
        "pl_homography_4pt": (cint, [vp, vp, vp]),
        "pl_solve_batch": (cint, [cint, vp, sz, vp, vp]),
    }
    missing = [name for name in EXPORTED_SYMBOLS if name not in sig]
    assert not missing, f"no signature declared for {missing}"
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    # the ctypes mirrors of this module restate the records of include/poselib_amd.h at PL_ABI_VERSION 5
    if L.pl_abi_version() != ABI_VERSION:
        raise PoseLibAmdError(f"{LIB_PATH} has record layout version {L.pl_abi_version()}, this binding was written for {ABI_VERSION}: rebuild")


def check(rc: int):
    if rc < 0:
        msg = lib().pl_last_error().decode()
        raise PoseLibAmdError(f"poselib_amd error {rc}: {msg}")
    return rc


ABI_VERSION = 5  # PL_ABI_VERSION of include/poselib_amd.h
EXPORTED_SYMBOLS = [
    "pl_default_ransac_options", "pl_default_bundle_options", "pl_default_robust_options", "pl_device_count",
    "pl_set_device", "pl_last_error", "pl_version", "pl_estimate_absolute_pose", "pl_estimate_relative_pose",
    "pl_estimate_fundamental", "pl_estimate_homography", "pl_ransac_pnp", "pl_ransac_relpose", "pl_ransac_fundamental",
    "pl_ransac_homography", "pl_problem_create", "pl_problem_destroy", "pl_ransac_run", "pl_ransac_run_sharded", "pl_score_model", "pl_debug_score_stream", "pl_refine_model", "pl_bundle_adjust_camera", "pl_p3p", "pl_relpose_5pt",
    "pl_essential_matrix_5pt", "pl_relpose_7pt", "pl_homography_4pt", "pl_solve_batch", "pl_estimate_batch", "pl_estimate_batch_devices", "pl_last_batch_report", "pl_undistort_points",
    "pl_ransac_batch", "pl_debug_device_math", "pl_ransac_pnpf", "pl_ransac_shared_focal_relpose", "pl_refine_shared_focal_relpose",
    "pl_estimate_shared_focal_relative_pose", "pl_solve_focal_batch", "pl_p35pf", "pl_relpose_6pt_shared_focal", "pl_set_lm_mode",
    "pl_abi_version",
]
