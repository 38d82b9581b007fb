cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python bench_batch.py --problems 4096 2>&1 | tail -1 | cut -c1-120
