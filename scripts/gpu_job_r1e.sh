set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1e
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r1e/pytest_gpu.log
for P in 5 4 6 3; do
  POSELIB_AMD_PF_P=$P timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1e/bench_s1_P$P.json 2>&1
  POSELIB_AMD_PF_P=$P timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r1e/bench_s8_P$P.json 2>&1
done
cat gpurun_out/r1e/pytest_gpu.log
for f in gpurun_out/r1e/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'])"; done
