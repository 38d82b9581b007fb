cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded" 2>&1 | tail -15
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
