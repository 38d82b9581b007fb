#!/usr/bin/env python
"""configs[0] under the profiler: estimate_absolute_pose on 200 correspondences, default options, one call after the other.
    rocprofv3 --kernel-trace --stats -- python scripts/p3p200_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P
from poselib_amd import synth
ds = [synth.absolute_pose_scene(200, 0.5, 4200 + k) for k in range(8)]
run = lambda j: P.estimate_absolute_pose(ds[j % 8]["p2d"], ds[j % 8]["p3d"], ds[j % 8]["camera"], {"ransac": {"seed": j}})
for j in range(8):
    run(j)
t0 = time.perf_counter()
outs = [run(j) for j in range(64)]
dt = time.perf_counter() - t0
print(f"p3p_200_default: {1e3 * dt / 64:.3f} ms per call, mean iterations {sum(o[1]['iterations'] for o in outs) / 64:.0f}, refinements {sum(o[1]['refinements'] for o in outs) / 64:.1f}")
