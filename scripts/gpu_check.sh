cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/check
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 300 python scripts/latency_probe.py 2>&1 | tail -3
for S in 8 16; do timeout 300 python bench_batch.py --problems 2048 --streams $S --no-cpu-baseline > gpurun_out/check/batch_s$S.json 2>&1; echo "S=$S $(tail -1 gpurun_out/check/batch_s$S.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['problems_per_s'], d['ms_per_step'])")"; done
