cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1n
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r1n/pytest_gpu.log
cat gpurun_out/r1n/pytest_gpu.log
for w in p3p_5000 hom_10000; do
  timeout 200 python bench.py --workload $w --streams 1 --no-cpu-baseline > gpurun_out/r1n/bench_s1_$w.json 2>&1
done
timeout 200 python bench.py > gpurun_out/r1n/bench_default.json 2>gpurun_out/r1n/bench_default.err
for f in gpurun_out/r1n/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
timeout 300 python bench_batch.py --problems 2048 > gpurun_out/r1n/batch.json 2>&1; tail -1 gpurun_out/r1n/batch.json | cut -c1-700
