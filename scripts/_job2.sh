cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 300 python scripts/latency_probe.py 2>&1 | tail -3
timeout 300 python bench_batch.py --problems 4096 2>&1 | tail -1 | cut -c1-120
