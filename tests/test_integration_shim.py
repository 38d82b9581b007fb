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


CHECK_BIN = os.path.join(ROOT, "integration", "_build", "robust_amd_check")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "PoseLib")), reason="reference headers not available")
def test_check_program_builds():
    """the executed form of the binding (integration/check_main.cc + robust_amd.cc + the reference's Camera class)"""
    poselib_amd.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "integration"), "-s"])
    assert os.path.exists(CHECK_BIN)
    undefined = subprocess.check_output(["nm", "-DC", "--undefined-only", CHECK_BIN], text=True)
    assert "pl_estimate_absolute_pose" in undefined and "pl_estimate_homography" in undefined  # bound to the C-ABI
    defined = subprocess.check_output(["nm", "-C", "--defined-only", CHECK_BIN], text=True)
    assert "poselib::estimate_absolute_pose(" in defined  # the reference's own symbol, defined by robust_amd.cc
    assert "poselib::AbsolutePoseEstimator" not in defined  # none of the reference's CPU estimators inside


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4, 5, 6, 7])
def test_reference_side_binding_executes_and_equals_the_ctypes_path(gpu, tmp_path, kind):
    """A program written against the REFERENCE'S C++ API (std::vector<Eigen::Vector2d>, Image, Camera, CameraPose,
    *Options; robust.h:45-46, 68-70, 112-113, 133-134) and linked with integration/robust_amd.cc runs on the GPU and
    returns, bit for bit, what the ctypes binding returns for the same inputs."""
    import numpy as np

    from poselib_amd import synth

    assert os.path.exists(CHECK_BIN), "integration/_build/robust_amd_check was not built (python __graft_entry__.py)"
    seed = 3 + kind
    min_fov = 0.0
    if kind in (0, 5, 6, 7):  # (7: the binding's multi-device batch call over the device list {0, 0}; its first problem is compared)
        d = synth.absolute_pose_scene(1500, 0.4, 4100)
        a, b, cam = d["p2d"], d["p3d"], d["camera"]
        if kind == 6:  # ransac_pnpf: points relative to the principal point; a field-of-view bound ABOVE the camera's 53 degrees
            a = a - np.asarray(cam["params"][1:3])  # (the binding dropped opt.min_fov - the bound must be seen to act)
            min_fov = 90.0
        if kind == 5:  # estimate_focal_length: the camera's focal length is 20 % off, the estimator must not care
            cam = dict(cam, params=[1.2 * cam["params"][0]] + list(cam["params"][1:]))
    else:
        gen = {1: synth.relative_pose_scene, 2: synth.fundamental_scene, 3: synth.homography_scene, 4: synth.relative_pose_scene}[kind]
        d = gen(1500, 0.4, 4100 + kind)
        a, b = d["x1"], d["x2"]
        cam = d.get("camera1", {"model": "SIMPLE_PINHOLE", "params": [1000.0, 500.0, 500.0]})
    max_error = 12.0 if kind in (0, 5, 6, 7) else 1.0
    opt = {"max_error": max_error, "ransac": {"seed": seed}}
    if kind == 6:
        opt["min_fov"] = min_fov
        img, info = gpu.ransac_pnpf(a, b, opt)
        model = np.r_[img.pose.q, img.pose.t]
        cam_out = list(img.camera.params)
        _, info_default = gpu.ransac_pnpf(a, b, {"max_error": max_error, "ransac": {"seed": seed}})
        assert info_default["num_inliers"] > 500  # the default bound (5 degrees) lets the true focal length through ...
        assert info["num_inliers"] < info_default["num_inliers"]  # ... 90 degrees cuts it off: the option reaches the estimator
    elif kind in (0, 5, 7):
        if kind == 5:
            opt["estimate_focal_length"] = True
        img, info = gpu.estimate_absolute_pose(a, b, cam, opt)
        model = np.r_[img.pose.q, img.pose.t]
        cam_out = list(img.camera.params)
    elif kind == 1:
        pose, info = gpu.estimate_relative_pose(a, b, cam, cam, opt)
        model, cam_out = np.r_[pose.q, pose.t], []
    elif kind == 4:
        pair, info = gpu.estimate_shared_focal_relative_pose(a, b, cam["params"][1:3], opt)
        model, cam_out = np.r_[pair.pose.q, pair.pose.t], list(pair.camera1.params)
    elif kind == 2:
        F, info = gpu.estimate_fundamental(a, b, opt)
        model, cam_out = F.T.reshape(-1), []  # column-major like Eigen::Matrix3d::data()
    else:
        H, info = gpu.estimate_homography(a, b, opt)
        model, cam_out = H.T.reshape(-1), []
    n = a.shape[0]
    model_id = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1, "OPENCV": 4}[cam["model"]]
    params = list(cam["params"]) + [0.0] * (12 - len(cam["params"]))
    params[11] = min_fov
    blob = np.r_[float(kind), float(n), float(seed), max_error, float(model_id), float(len(cam["params"])), params,
                 np.ascontiguousarray(a, dtype=np.float64).reshape(-1), np.ascontiguousarray(b, dtype=np.float64).reshape(-1)]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    blob.astype(np.float64).tofile(fin)
    r = subprocess.run([CHECK_BIN, str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype=np.float64)
    nm = len(model)
    assert out[0] == info["iterations"] and out[1] == info["refinements"] and out[2] == info["num_inliers"]
    assert out[3] == info["model_score"]
    assert np.array_equal(out[4:4 + nm], model)  # bit for bit
    if kind in (0, 4, 5, 6):
        assert np.array_equal(out[4 + nm:4 + nm + len(cam_out)], np.array(cam_out))
    if kind == 4:
        assert abs(cam_out[0] - cam["params"][0]) < 1e-2 * cam["params"][0]
    assert np.array_equal(out[4 + nm + 12:].astype(bool), np.array(info["inliers"]))
    assert info["num_inliers"] > 500 or kind == 6
