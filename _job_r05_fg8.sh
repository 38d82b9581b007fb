#!/bin/bash
R=/root/repo
O=$R/gpurun_out/fg8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_gpu_focal.py tests/test_zz_gpu_shared_focal.py tests/test_zz_gpu_focal_group.py -x -q 2>&1 | tail -5
timeout 300 python scripts/focal_batch_bench.py 1024 2000 > $O/focal_batch.md 2>$O/focal_batch.err; cat $O/focal_batch.md
for e in pnpf shared_focal; do timeout 300 python $R/scripts/focal_batch_trace.py $e 1024 2000 4 8 2>&1 | tail -4; done
