#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/fg3
export TMPDIR=/tmp
for e in pnpf shared_focal; do
  (cd /tmp && POSELIB_AMD_FOCAL_GROUP=64 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$e -- python /root/repo/scripts/focal_batch_trace.py $e 1024 2000 3 1 > /root/repo/gpurun_out/fg3/trace_$e.log 2>&1)
  f=$(find /tmp/prof_$e -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/fg3/kernel_stats_1worker_$e.csv
  head -12 "$f" | cut -c1-200
  tail -4 gpurun_out/fg3/trace_$e.log
done
