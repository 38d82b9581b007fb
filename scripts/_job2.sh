cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
