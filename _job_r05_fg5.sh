#!/bin/bash
R=/root/repo
O=$R/gpurun_out/fg5
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in base fstop1 fstop2; do
  if [ $v = base ]; then unset POSELIB_AMD_LIB; else export POSELIB_AMD_LIB=$R/scripts/exp/variants/$v/libposelib_amd.so; fi
  POSELIB_AMD_FOCAL_GROUP=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -- python $R/scripts/focal_batch_trace.py pnpf 512 2000 2 1 > $O/$v.log 2>&1
  f=$(find /tmp/p_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep solve_g $f | cut -d, -f1-4)"
done
for v in base sstop1 sstop2 sstop3; do
  if [ $v = base ]; then unset POSELIB_AMD_LIB; else export POSELIB_AMD_LIB=$R/scripts/exp/variants/$v/libposelib_amd.so; fi
  POSELIB_AMD_FOCAL_GROUP=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$v -- python $R/scripts/focal_batch_trace.py shared_focal 512 2000 2 1 > $O/s_$v.log 2>&1
  f=$(find /tmp/q_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep solve_g $f | cut -d, -f1-4)"
done
