cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1t
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
for w in hom_10000 fund_10000 p3p_5000; do
  timeout 200 python bench.py --workload $w --streams 1 --steps 5 --no-cpu-baseline > gpurun_out/r1t/bench_s1_$w.json 2>&1
  POSELIB_AMD_NO_LM2=1 timeout 200 python bench.py --workload $w --streams 1 --steps 5 --no-cpu-baseline > gpurun_out/r1t/bench_s1_nolm2_$w.json 2>&1
  timeout 200 python bench.py --workload $w --steps 5 --no-cpu-baseline > gpurun_out/r1t/bench_s16_$w.json 2>&1
  POSELIB_AMD_NO_LM2=1 timeout 200 python bench.py --workload $w --steps 5 --no-cpu-baseline > gpurun_out/r1t/bench_s16_nolm2_$w.json 2>&1
done
for f in gpurun_out/r1t/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r1t/kt_hom -o k -- python $R/bench.py --workload hom_10000 --steps 4 --warmup 2 --no-cpu-baseline --streams 1 > $R/gpurun_out/r1t/kt_hom.log 2>&1
cd $R
python scripts/timeline.py $(find gpurun_out/r1t/kt_hom -name "*kernel_trace.csv") | grep -v copyBuffer | awk '/k_lm2/{n++; if(n<=4||n>50) print; next} {print}'
