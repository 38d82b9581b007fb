#!/bin/bash
R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_zz_gpu_focal.py tests/test_zz_gpu_shared_focal.py tests/test_zz_gpu_focal_group.py tests/test_golden_focal.py -x -q 2>&1 | tail -4
python - <<'PY'
import time, sys
sys.path.insert(0,'/root/repo')
import poselib_amd as P
from poselib_amd import synth
n=2000
da=[synth.absolute_pose_scene(n,0.4,7100+k) for k in range(4)]
dr=[synth.relative_pose_scene(n,0.4,7000+k) for k in range(4)]
def ra(j):
    d=da[j%4]; return P.estimate_absolute_pose(d["p2d"],d["p3d"],d["camera"],{"max_error":4.0,"estimate_focal_length":True,"ransac":{"seed":j}})
def rr(j):
    d=dr[j%4]; return P.estimate_shared_focal_relative_pose(d["x1"],d["x2"],d["camera1"]["params"][1:3],{"max_error":2.0,"ransac":{"seed":j}})
for name,f in (("pnpf",ra),("shared_focal",rr)):
    for j in range(4): f(j)
    t0=time.perf_counter()
    for j in range(100): f(j)
    print(name, "ms per problem", 1e3*(time.perf_counter()-t0)/100)
PY
for e in pnpf shared_focal; do timeout 300 python $R/scripts/focal_batch_trace.py $e 1024 2000 4 8 2>&1 | tail -2; done
