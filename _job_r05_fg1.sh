#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/fg1
timeout 900 python -m pytest tests/test_zz_gpu_focal_group.py -x -q 2>&1 | tail -40 > gpurun_out/fg1/test_group.log
timeout 900 python -m pytest tests/test_zz_gpu_focal.py tests/test_zz_gpu_shared_focal.py -x -q 2>&1 | tail -15 > gpurun_out/fg1/test_focal.log
timeout 600 python scripts/focal_batch_bench.py 1024 2000 > gpurun_out/fg1/focal_batch.md 2> gpurun_out/fg1/focal_batch.err
tail -5 gpurun_out/fg1/test_group.log; tail -3 gpurun_out/fg1/test_focal.log; cat gpurun_out/fg1/focal_batch.md; tail -5 gpurun_out/fg1/focal_batch.err
