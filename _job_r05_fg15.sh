#!/bin/bash
R=/root/repo
O=$R/gpurun_out/fg15
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for e in pnpf shared_focal; do
  B1="python $R/scripts/focal_batch_trace.py $e 512 2000 2 1"
  POSELIB_AMD_FOCAL_GROUP=64 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_$e -o p -- $B1 > $O/pmc_sq_$e.log 2>&1
  POSELIB_AMD_FOCAL_GROUP=64 timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq2_$e -o p -- $B1 > $O/pmc_sq2_$e.log 2>&1
  find $O -name "*kernel_trace.csv" -delete
done
ls -la $O/*/ | head -20
