#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/fg2
export TMPDIR=/tmp
for e in pnpf shared_focal; do
  POSELIB_AMD_GROUP_TIMING=1 timeout 300 python scripts/focal_batch_trace.py $e 1024 2000 3 > gpurun_out/fg2/plain_$e.log 2>&1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$e -- python /root/repo/scripts/focal_batch_trace.py $e 1024 2000 3 > /root/repo/gpurun_out/fg2/trace_$e.log 2>&1)
  f=$(find /tmp/prof_$e -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/fg2/kernel_stats_$e.csv
  head -25 "$f"
  cat gpurun_out/fg2/plain_$e.log | tail -8
done
