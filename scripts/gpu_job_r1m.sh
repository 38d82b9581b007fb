cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1m
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r1m/pytest_gpu.log
cat gpurun_out/r1m/pytest_gpu.log
for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
  timeout 200 python bench.py --workload $w --streams 1 --no-cpu-baseline > gpurun_out/r1m/bench_s1_$w.json 2>&1
  timeout 200 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r1m/bench_s16_$w.json 2>&1
done
for f in gpurun_out/r1m/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
timeout 300 python bench_batch.py --problems 2048 > gpurun_out/r1m/batch.json 2>&1; tail -1 gpurun_out/r1m/batch.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
for w in hom_10000 fund_10000; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r1m/kt_$w -o k -- python $R/bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --streams 1 > $R/gpurun_out/r1m/kt_$w.log 2>&1
done
cd $R
for w in hom_10000 fund_10000; do python scripts/timeline.py $(find gpurun_out/r1m/kt_$w -name "*kernel_trace.csv") | grep -v copyBuffer; done
