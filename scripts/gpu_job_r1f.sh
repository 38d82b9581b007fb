set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1f
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r1f/pmc1 -o p -- $B > $R/gpurun_out/r1f/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $R/gpurun_out/r1f/pmc2 -o p -- $B > $R/gpurun_out/r1f/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $R/gpurun_out/r1f/pmc3 -o p -- $B > $R/gpurun_out/r1f/pmc3.log 2>&1
cd $R
timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1f/bench_normal.json 2>&1
cp poselib_amd/lib/libposelib_amd.so /tmp/lib_normal.so
cp build/exp/libposelib_amd_exp.so poselib_amd/lib/libposelib_amd.so
timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1f/bench_noexact.json 2>&1
cp /tmp/lib_normal.so poselib_amd/lib/libposelib_amd.so
for f in gpurun_out/r1f/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
python scripts/pmc_summary.py $(find gpurun_out/r1f -name "*counter_collection.csv") > gpurun_out/r1f/pmc_summary.md 2>&1
cat gpurun_out/r1f/pmc_summary.md | head -30
