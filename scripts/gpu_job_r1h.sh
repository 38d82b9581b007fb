set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1h
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r1h/pytest_gpu.log
for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
  timeout 200 python bench.py --workload $w --streams 1 --no-cpu-baseline > gpurun_out/r1h/bench_s1_$w.json 2>&1
  timeout 200 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r1h/bench_s8_$w.json 2>&1
done
cat gpurun_out/r1h/pytest_gpu.log
for f in gpurun_out/r1h/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'])"; done
