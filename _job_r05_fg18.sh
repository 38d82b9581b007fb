#!/bin/bash
R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_zz_gpu_focal.py tests/test_zz_gpu_shared_focal.py tests/test_zz_gpu_focal_group.py -x -q 2>&1 | tail -4
for e in pnpf; do timeout 300 python $R/scripts/focal_batch_trace.py $e 1024 2000 5 8 2>&1 | tail -3; done
cd /tmp; export TMPDIR=/tmp
for e in pnpf; do
POSELIB_AMD_FOCAL_GROUP=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$e -- python $R/scripts/focal_batch_trace.py $e 1024 2000 3 1 > /dev/null 2>&1
f=$(find /tmp/p_$e -name "*kernel_stats.csv" | head -1); mkdir -p $R/gpurun_out/fg18; cp $f $R/gpurun_out/fg18/kernel_stats_1worker_$e.csv; head -8 $f | cut -d, -f1-4
done
