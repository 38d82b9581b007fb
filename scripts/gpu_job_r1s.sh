cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1s
timeout 300 python scripts/latency_probe.py 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import poselib_amd as P
from poselib_amd import synth
d = synth.absolute_pose_scene(2000, 0.5, 5)
for i in range(6):
    P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], {"ransac": {"seed": i}})
PY
timeout 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $R/gpurun_out/r1s/tr -o t -- python /tmp/one.py $R > $R/gpurun_out/r1s/tr.log 2>&1
ls $R/gpurun_out/r1s/tr
