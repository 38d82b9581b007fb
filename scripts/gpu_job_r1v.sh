cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1v
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for S in 1 16; do timeout 200 python bench.py --workload relpose_5000 --streams $S --steps 5 --no-cpu-baseline > gpurun_out/r1v/bench_s${S}_rel.json 2>&1; echo "S=$S $(tail -1 gpurun_out/r1v/bench_s${S}_rel.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], '%.3f'%d['ms_per_step'])")"; done
