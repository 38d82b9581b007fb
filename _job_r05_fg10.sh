#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/fg10
timeout 1200 python scripts/soak_focal_group.py 600 > gpurun_out/fg10/soak_focal_group.md 2> gpurun_out/fg10/soak.err
cat gpurun_out/fg10/soak_focal_group.md | tail -12; tail -5 gpurun_out/fg10/soak.err
