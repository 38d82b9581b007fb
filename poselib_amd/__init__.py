"""poselib_amd — MI355X-native LO-RANSAC pose estimation, drop-in for the robust-estimation hot path of
PoseLib (estimate_absolute_pose / estimate_relative_pose / estimate_fundamental / estimate_homography,
ransac_*, and the p3p / relpose_5pt / relpose_7pt / homography_4pt minimal solvers).

All numerical work runs in hand-written HIP kernels (poselib_amd/csrc) behind the C-ABI declared in
include/poselib_amd.h; this package only marshals numpy arrays.  No CPU fallback.
"""
from ._lib import LIB_PATH, PoseLibAmdError, build  # noqa: F401
from .api import (  # noqa: F401
    KIND_ABS, KIND_FUND, KIND_HOM, KIND_REL, Batch, RansacBatch, ransac_batch, device_math, BundleOptions, Camera, CameraPose, Image, Problem, RansacOptions,
    device_count, essential_matrix_5pt, estimate_absolute_pose, estimate_batch, last_batch_report, estimate_fundamental, estimate_homography,
    estimate_relative_pose, homography_4pt, p3p, p5p, ransac_fundamental, ransac_homography, ransac_pnp, ransac_pnpf,
    ransac_relpose, ransac_shared_focal_relpose, estimate_shared_focal_relative_pose, refine_shared_focal_relpose, ImagePair, p35pf, relpose_6pt_shared_focal, solve_focal_batch, relpose_5pt, relpose_7pt, set_device, set_lm_mode, solve_batch, undistort_points,
)

__version__ = "0.1.0"
