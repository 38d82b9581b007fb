#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_zz_gpu_shared_focal.py -x -q -k "both_focal_solvers" 2>&1 | tail -8
