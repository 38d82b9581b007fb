"""integration/robust_amd.cc — the reference-side binding shown in INTEGRATION.md — must compile against the
reference's own headers (robust.h, ransac.h, types.h, camera_pose.h, solver headers) and link with the C-ABI library.
Real Eigen is not in this image, so the reference headers see oracle/eigen_shim; skipped where /root/reference is
absent (the GPU box)."""
import os
import subprocess

import pytest

import poselib_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "PoseLib")), reason="reference headers not available")
def test_reference_side_binding_compiles_and_links(tmp_path):
    poselib_amd.build()
    obj = tmp_path / "robust_amd.o"
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-Wall", "-Werror=return-type", "-c",
                           os.path.join(ROOT, "integration", "robust_amd.cc"), "-o", str(obj),
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle", "eigen_shim"),
                           "-I", REF])
    so = tmp_path / "librobust_amd.so"
    subprocess.check_call(["g++", "-shared", "-o", str(so), str(obj), "-L", os.path.dirname(poselib_amd.LIB_PATH),
                           "-lposelib_amd", "-Wl,--no-undefined", "-Wl,-rpath," + os.path.dirname(poselib_amd.LIB_PATH)])
    syms = subprocess.check_output(["nm", "-DC", "--defined-only", str(so)], text=True)
    for name in ("poselib::estimate_absolute_pose", "poselib::estimate_relative_pose", "poselib::estimate_fundamental",
                 "poselib::estimate_homography", "poselib::ransac_pnp", "poselib::ransac_relpose",
                 "poselib::ransac_fundamental", "poselib::ransac_homography", "poselib::p3p", "poselib::relpose_5pt"):
        assert name + "(" in syms, name
