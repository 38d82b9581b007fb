set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1d
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r1d/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/r1d/bench_default.json 2> gpurun_out/r1d/bench_default.err
timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1d/bench_s1.json 2>&1
for w in relpose_5000 fund_10000 hom_10000; do timeout 200 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r1d/bench_$w.json 2>&1; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r1d/prof_s1 -o r -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1d/prof_s1.log 2>&1
for w in relpose_5000 fund_10000 hom_10000; do timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r1d/prof_$w -o r -- python $GRAFT_REPO_ROOT/bench.py --workload $w --streams 1 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1d/prof_$w.log 2>&1; done
cd $GRAFT_REPO_ROOT
for d in prof_s1 prof_relpose_5000 prof_fund_10000 prof_hom_10000; do f=$(find gpurun_out/r1d/$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > gpurun_out/r1d/$d.md; done
find gpurun_out/r1d -name "*.db" -size +20M -delete
cat gpurun_out/r1d/pytest_gpu.log gpurun_out/r1d/bench_default.json
