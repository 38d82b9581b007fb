#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/fg16
timeout 1500 python scripts/soak_focal_gpu.py 250 > gpurun_out/fg16/soak_focal_estimators.md 2> gpurun_out/fg16/soak.err
cat gpurun_out/fg16/soak_focal_estimators.md | tail -8; tail -3 gpurun_out/fg16/soak.err
