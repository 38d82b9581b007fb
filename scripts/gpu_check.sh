# quick GPU regression: parity suite, latency probe, batch bench (used during development)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/check
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
timeout 300 python scripts/latency_probe.py 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/check/bench.json 2>&1; tail -1 gpurun_out/check/bench.json | cut -c1-200
