#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_zz_gpu_focal_group.py -x -q 2>&1 | tail -12
