set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1k
cd /tmp && export TMPDIR=/tmp
for S in 8 16; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r1k/kt_s$S -o k -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams $S > $R/gpurun_out/r1k/kt_s$S.log 2>&1
done
cd $R
for S in 8 16; do python scripts/busy.py $(find gpurun_out/r1k/kt_s$S -name "*kernel_trace.csv") > gpurun_out/r1k/busy_s$S.txt; cat gpurun_out/r1k/busy_s$S.txt; tail -1 gpurun_out/r1k/kt_s$S.log | cut -c1-200; done
