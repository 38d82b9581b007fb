cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1u
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
for w in hom_10000 fund_10000; do
  POSELIB_AMD_LATENCY_MODE=1 timeout 200 python bench.py --workload $w --streams 1 --steps 5 --no-cpu-baseline > gpurun_out/r1u/bench_s1_lat_$w.json 2>&1
  timeout 200 python bench.py --workload $w --streams 1 --steps 5 --no-cpu-baseline > gpurun_out/r1u/bench_s1_$w.json 2>&1
done
for f in gpurun_out/r1u/bench_*.json; do echo "$(basename $f) $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], '%.3f'%d['ms_per_step'])")"; done
