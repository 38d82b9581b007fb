# Evidence of the last third of round 3 (the focal-length estimators on top of the final build): the full GPU suite, the bench
# line as the driver runs it, a kernel trace of the two estimators and their soak.  Outputs under gpurun_out/r3b/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --detail-file $O/bench_detail.json > $O/bench_default.json 2> $O/bench_default.err
timeout 120 python scripts/time_focal_estimators.py 5 > $O/focal_timing.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof_focal -o r -- python scripts/time_focal_estimators.py 3 > $O/prof_focal.log 2>&1
f=$(find $O/prof_focal -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/prof_focal.md
timeout 200 python scripts/soak_focal_gpu.py 150 > $O/soak_focal.md 2> $O/soak_focal.err
find $O -name "*.db" -delete
cat $O/pytest_gpu.log; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err; cat $O/focal_timing.log; head -14 $O/prof_focal.md | cut -c1-150; tail -6 $O/soak_focal.md
