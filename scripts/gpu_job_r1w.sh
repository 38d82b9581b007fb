cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1w
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r1w/bench.json 2>&1; tail -1 gpurun_out/r1w/bench.json | cut -c1-330
