"""Bit-reproducible synthetic correspondence sets for the parity tests and bench.py.

The recipe follows the reference's benchmark/problem_generator.cc:338-405 (absolute pose),
:537-639 (relative pose) and :641-743 (homography): random pose (quaternion = normalised
4 x N(0,1), t ~ U[-1,1]^3, unit |t| for two-view), image points U[-s,s]^2 with
s = tan(fov/2), depths U[0.1,10].  The reference draws from std::default_random_engine /
Eigen::setRandom (implementation defined); here every number comes from a counter-based
splitmix64 stream so that the CPU oracle and the GPU see identical buffers on any host.
"""
from __future__ import annotations

import math

import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


class Stream:
    """Counter-based splitmix64: value j is mix(seed + (j+1)*G)."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        self.pos = 0

    def _raw(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            j = np.arange(self.pos + 1, self.pos + n + 1, dtype=np.uint64)
            z = self.seed + j * _G
            z = (z ^ (z >> np.uint64(30))) * _C1
            z = (z ^ (z >> np.uint64(27))) * _C2
            z = z ^ (z >> np.uint64(31))
        self.pos += n
        return z

    def uniform(self, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
        u = (self._raw(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def normal(self, n: int) -> np.ndarray:
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)  # (0,1]
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        out = np.concatenate([r * np.cos(2 * math.pi * u2), r * np.sin(2 * math.pi * u2)])
        return out[:n]


def quat_to_rotmat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def random_pose(rs: Stream, unit_translation: bool = False, max_rotation: float = None):
    q = rs.normal(4)
    if max_rotation is not None:
        # two-view scenes: keep the relative rotation moderate so that both cameras see the scene
        q = np.array([1.0, max_rotation * q[1], max_rotation * q[2], max_rotation * q[3]])
    q = q / np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    t = rs.uniform(3, -1.0, 1.0)
    if unit_translation:
        t = t / np.linalg.norm(t)
    return q, t


def _fov_scale(fov_deg: float) -> float:
    return math.tan(fov_deg / 2.0 * math.pi / 180.0)


def absolute_pose_scene(n: int, outlier_ratio: float, seed: int, noise_px: float = 0.5, focal: float = 1000.0,
                        pp=(500.0, 500.0), fov_deg: float = 70.0):
    """2D-3D correspondences seen by a SIMPLE_PINHOLE camera (config 0 / 1 of BASELINE.json)."""
    rs = Stream(seed)
    q, t = random_pose(rs)
    R = quat_to_rotmat(q)
    s = _fov_scale(fov_deg)
    xy = rs.uniform(2 * n, -s, s).reshape(n, 2)
    depth = rs.uniform(n, 0.1, 10.0)
    bearing = np.concatenate([xy, np.ones((n, 1))], axis=1)
    bearing /= np.linalg.norm(bearing, axis=1, keepdims=True)
    Xc = bearing * depth[:, None]
    X = (Xc - t) @ R  # R^T (Xc - t)
    pix = xy * focal + np.asarray(pp)
    pix = pix + noise_px * rs.normal(2 * n).reshape(n, 2)
    n_out = int(round(outlier_ratio * n))
    is_out = np.zeros(n, dtype=bool)
    if n_out:
        order = np.argsort(rs.uniform(n), kind="stable")
        is_out[order[:n_out]] = True
        rnd = rs.uniform(2 * n, -s, s).reshape(n, 2) * focal + np.asarray(pp)
        pix[is_out] = rnd[is_out]
    camera = {"model": "SIMPLE_PINHOLE", "width": int(2 * pp[0]), "height": int(2 * pp[1]),
              "params": [focal, pp[0], pp[1]]}
    return {"p2d": np.ascontiguousarray(pix), "p3d": np.ascontiguousarray(X), "camera": camera, "q_gt": q, "t_gt": t,
            "inlier_gt": ~is_out}


def _two_view_points(rs: Stream, n: int, R, t, s: float, planar: bool):
    """Normalised image points in both views of n scene points in front of both cameras."""
    x1 = np.zeros((0, 2))
    x2 = np.zeros((0, 2))
    if planar:
        nrm = rs.normal(3)
        nrm = nrm / np.linalg.norm(nrm)
        if nrm[2] > 0:
            nrm = -nrm  # plane faces camera 1
        if abs(nrm[2]) < 0.5:
            nrm = np.array([0.2, -0.1, -1.0]) / np.linalg.norm([0.2, -0.1, -1.0])
        dist = float(rs.uniform(1, 2.0, 6.0)[0])
    rounds = 0
    while x1.shape[0] < n:
        rounds += 1
        if rounds > 1000:
            raise RuntimeError("synthetic two-view scene: cameras share no visible points")
        m = 2 * (n - x1.shape[0]) + 16
        xy = rs.uniform(2 * m, -s, s).reshape(m, 2)
        b = np.concatenate([xy, np.ones((m, 1))], axis=1)
        if planar:
            depth = -dist / (b @ nrm)  # n.X + dist = 0
            X1 = b * depth[:, None]
            ok = depth > 0.05
        else:
            bn = b / np.linalg.norm(b, axis=1, keepdims=True)
            depth = rs.uniform(m, 0.1, 10.0)
            X1 = bn * depth[:, None]
            ok = np.ones(m, dtype=bool)
        X2 = X1 @ R.T + t
        ok &= X2[:, 2] > 0.05
        p2 = X2[:, :2] / X2[:, 2:3]
        ok &= (np.abs(p2) < 2.0 * s).all(axis=1)
        x1 = np.concatenate([x1, xy[ok]])
        x2 = np.concatenate([x2, p2[ok]])
    return x1[:n], x2[:n]


def _corrupt(rs: Stream, x2n: np.ndarray, outlier_ratio: float, s: float):
    n = x2n.shape[0]
    n_out = int(round(outlier_ratio * n))
    is_out = np.zeros(n, dtype=bool)
    order = np.argsort(rs.uniform(n), kind="stable")
    rnd = rs.uniform(2 * n, -s, s).reshape(n, 2)
    if n_out:
        is_out[order[:n_out]] = True
        x2n = x2n.copy()
        x2n[is_out] = rnd[is_out]
    return x2n, is_out


def relative_pose_scene(n: int, outlier_ratio: float, seed: int, noise_px: float = 0.5, focal: float = 1000.0,
                        pp=(500.0, 500.0), fov_deg: float = 70.0):
    """2D-2D correspondences between two SIMPLE_PINHOLE cameras (config 2)."""
    rs = Stream(seed)
    q, t = random_pose(rs, unit_translation=True, max_rotation=0.2)
    R = quat_to_rotmat(q)
    s = _fov_scale(fov_deg)
    x1n, x2n = _two_view_points(rs, n, R, t, s, planar=False)
    x2n, is_out = _corrupt(rs, x2n, outlier_ratio, s)
    noise = noise_px * rs.normal(4 * n).reshape(n, 4)
    p1 = x1n * focal + np.asarray(pp) + noise[:, :2]
    p2 = x2n * focal + np.asarray(pp) + noise[:, 2:]
    cam = {"model": "SIMPLE_PINHOLE", "width": int(2 * pp[0]), "height": int(2 * pp[1]), "params": [focal, pp[0], pp[1]]}
    return {"x1": np.ascontiguousarray(p1), "x2": np.ascontiguousarray(p2), "camera1": cam, "camera2": dict(cam),
            "q_gt": q, "t_gt": t, "inlier_gt": ~is_out}


def homography_scene(n: int, outlier_ratio: float, seed: int, noise_px: float = 0.5, focal: float = 1000.0,
                     pp=(500.0, 500.0), fov_deg: float = 70.0):
    """Pixel correspondences of a planar scene (config 3, homography part)."""
    rs = Stream(seed)
    q, t = random_pose(rs, unit_translation=True, max_rotation=0.15)
    t = 0.5 * t
    R = quat_to_rotmat(q)
    s = _fov_scale(fov_deg)
    x1n, x2n = _two_view_points(rs, n, R, t, s, planar=True)
    x2n, is_out = _corrupt(rs, x2n, outlier_ratio, s)
    noise = noise_px * rs.normal(4 * n).reshape(n, 4)
    p1 = x1n * focal + np.asarray(pp) + noise[:, :2]
    p2 = x2n * focal + np.asarray(pp) + noise[:, 2:]
    return {"x1": np.ascontiguousarray(p1), "x2": np.ascontiguousarray(p2), "q_gt": q, "t_gt": t, "inlier_gt": ~is_out}


def fundamental_scene(n: int, outlier_ratio: float, seed: int, noise_px: float = 0.5, focal: float = 1000.0,
                      pp=(500.0, 500.0), fov_deg: float = 70.0):
    """Pixel correspondences of a general scene (config 3, 7-point part)."""
    d = relative_pose_scene(n, outlier_ratio, seed, noise_px, focal, pp, fov_deg)
    return {"x1": d["x1"], "x2": d["x2"], "q_gt": d["q_gt"], "t_gt": d["t_gt"], "inlier_gt": d["inlier_gt"]}


def opencv_distort_pixels(pix: np.ndarray, params) -> np.ndarray:
    """Push pinhole pixels through an OPENCV camera (fx, fy, cx, cy, k1, k2, p1, p2) with the same
    fx, fy, cx, cy — used for the "OPENCV camera model" variant of config 3."""
    fx, fy, cx, cy, k1, k2, p1, p2 = params
    u = (pix[:, 0] - cx) / fx
    v = (pix[:, 1] - cy) / fy
    r2 = u * u + v * v
    alpha = 1.0 + k1 * r2 + k2 * r2 * r2
    du = alpha * u + 2.0 * p1 * u * v + p2 * (r2 + 2.0 * u * u)
    dv = alpha * v + 2.0 * p2 * u * v + p1 * (r2 + 2.0 * v * v)
    return np.stack([fx * du + cx, fy * dv + cy], axis=1)
