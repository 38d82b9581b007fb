cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1o
cd /tmp && export TMPDIR=/tmp
for w in p3p_5000; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r1o/kt_$w -o k -- python $R/bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --streams 1 > $R/gpurun_out/r1o/kt_$w.log 2>&1
done
cd $R
for w in p3p_5000; do python scripts/timeline.py $(find gpurun_out/r1o/kt_$w -name "*kernel_trace.csv") | grep -v copyBuffer; done
