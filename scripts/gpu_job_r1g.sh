set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1g
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r1g/pytest_gpu.log
timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1g/bench_s1.json 2>&1
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r1g/bench_s8.json 2>&1
POSELIB_AMD_PF_P=4 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r1g/bench_s8_P4.json 2>&1
POSELIB_AMD_PF_P=4 timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1g/bench_s1_P4.json 2>&1
timeout 200 python bench.py --streams 16 --no-cpu-baseline > gpurun_out/r1g/bench_s16.json 2>&1
cat gpurun_out/r1g/pytest_gpu.log
for f in gpurun_out/r1g/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'])"; done
