"""k_sfocal_solve phase by phase: builds that return after phase n (sfocal.hip PL_SFOCAL_STOP), timed with HIP events around
pl_solve_focal_batch's launch sequence is not possible from Python - so the kernel time is taken from the call time of a LARGE batch
with the transfers subtracted by the phase-0 build (load only).   python scripts/exp/sfocal_phases.py <lib> [count]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import poselib_amd as P
from poselib_amd import synth
count = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(3)
r = synth.relative_pose_scene(4000, 0.0, 78, noise_px=0.5)
fr, cx, cy = r["camera1"]["params"]
def unit(p):
    b = np.c_[(np.asarray(p) - [cx, cy]) / 800.0, np.ones(len(p))]
    return b / np.linalg.norm(b, axis=1)[:, None]
b1, b2 = unit(r["x1"]), unit(r["x2"])
idx = np.array([rng.choice(4000, 6, replace=False) for _ in range(count)])
six = np.concatenate([b1[idx].reshape(count, 18), b2[idx].reshape(count, 18)], axis=1)
P.solve_focal_batch("relpose_6pt_shared_focal", six)
ts = []
for _ in range(7):
    t = time.perf_counter(); P.solve_focal_batch("relpose_6pt_shared_focal", six); ts.append(time.perf_counter() - t)
print(f"{os.path.basename(os.environ.get('POSELIB_AMD_LIB', 'default'))}: {count} samples, {1e3 * min(ts):.2f} ms per call (min of 7)")
