"""ctypes access to oracle/_ref/libposelib_ref.so — the REFERENCE'S OWN sources (front-ends, ransac_*, estimators,
loop, sampler, solvers, scoring, refiners + LM, camera models) compiled in place against oracle/eigen_shim
(recipe: oracle/Makefile.ref, C wrappers: oracle/ref_shim/ref_api.cc).  Test infrastructure only.

The library exports the oracle's C interface under the prefix ``ref_`` instead of ``orc_``, so `reference()` simply
swaps the handle behind tests/oracle_lib.py: inside the context every oracle_lib wrapper (sampler_draw, p3p, score,
ransac_pnp, ...) marshals exactly as for the oracle but runs the reference's code.
"""
import contextlib
import ctypes as C
import os
import subprocess

import oracle_lib as O

_ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_REF_LIB = os.path.join(_ORACLE_DIR, "_ref", "libposelib_ref.so")
_REF_LIB_FMA = os.path.join(_ORACLE_DIR, "_ref", "fma", "libposelib_ref.so")
REFERENCE_ROOT = "/root/reference"


def build(variant: str = "") -> str:
    """(Re)build where the reference sources exist; elsewhere (the GPU box) use the prebuilt file if it travelled.
    variant "": the reference's Release flags (-O3, SSE2, no contraction); "fma": its MARCH_NATIVE option restated
    portably (-O3 -march=x86-64-v3) - the second build of the reference-against-itself measurements."""
    O.build()
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "PoseLib")):
        cmd = ["make", "-C", _ORACLE_DIR, "-s", "-j8", "-f", "Makefile.ref", f"REF={REFERENCE_ROOT}"]
        subprocess.check_call(cmd + ([f"VARIANT={variant}"] if variant else []))
    return _REF_LIB_FMA if variant == "fma" else _REF_LIB


def available(variant: str = "") -> bool:
    try:
        return os.path.exists(build(variant))
    except (subprocess.CalledProcessError, OSError):
        return False


class _Proxy:
    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        assert name.startswith("orc_"), name
        return getattr(self._cdll, "ref_" + name[4:])


_proxies = {}


def _load(variant: str = ""):
    _proxy = _proxies.get(variant)
    if _proxy is None:
        cdll = C.CDLL(build(variant))
        cdll.ref_all_inlier_probability.restype = C.c_double
        cdll.ref_all_inlier_probability.argtypes = [C.c_uint64] * 3
        cdll.ref_dynamic_max_iter.restype = C.c_uint64
        cdll.ref_dynamic_max_iter.argtypes = [C.c_uint64] * 3 + [C.c_double] * 2 + [C.c_uint64] * 2
        for name in ("ref_score_reproj", "ref_score_sampson_pose", "ref_score_sampson_F", "ref_score_homography",
                     "ref_normalize_points"):
            getattr(cdll, name).restype = C.c_double
        cdll.ref_solve_cubic_single_real.argtypes = [C.c_double] * 3 + [C.c_void_p]
        cdll.ref_solve_cubic_real.argtypes = [C.c_double] * 3 + [C.c_void_p]
        _proxy = _proxies[variant] = _Proxy(cdll)
    return _proxy


@contextlib.contextmanager
def reference(variant: str = ""):
    O.lib()
    saved, O._lib = O._lib, _load(variant)
    try:
        yield O
    finally:
        O._lib = saved
