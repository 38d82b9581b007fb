#!/bin/bash
R=/root/repo
O=$R/gpurun_out/fg6
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_zz_gpu_focal.py tests/test_zz_gpu_shared_focal.py tests/test_zz_gpu_focal_group.py -x -q 2>&1 | tail -25 > $O/tests.log
tail -6 $O/tests.log
cd /tmp
for e in pnpf shared_focal; do
  POSELIB_AMD_FOCAL_GROUP=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$e -- python $R/scripts/focal_batch_trace.py $e 1024 2000 3 1 > $O/trace1_$e.log 2>&1
  f=$(find /tmp/p_$e -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_1worker_$e.csv
  head -6 $f | cut -d, -f1-4
  timeout 300 python $R/scripts/focal_batch_trace.py $e 1024 2000 4 8 2>&1 | tail -4
done
