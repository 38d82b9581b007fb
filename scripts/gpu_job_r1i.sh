set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1i
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1"
run_pmc () { # $1 tag
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r1i/pmc1_$1 -o p -- $B > $R/gpurun_out/r1i/pmc1_$1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/r1i/pmc2_$1 -o p -- $B > $R/gpurun_out/r1i/pmc2_$1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r1i/pmc3_$1 -o p -- $B > $R/gpurun_out/r1i/pmc3_$1.log 2>&1
}
run_pmc packed
cd $R
timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1i/bench_packed.json 2>&1
POSELIB_AMD_SCORER=stream timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1i/bench_packed_stream.json 2>&1
cp poselib_amd/lib/libposelib_amd.so /tmp/lib_normal.so
cp build/exp/libposelib_amd_scalar.so poselib_amd/lib/libposelib_amd.so
timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1i/bench_scalar.json 2>&1
POSELIB_AMD_SCORER=stream timeout 200 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r1i/bench_scalar_stream.json 2>&1
timeout 200 python bench.py --workload fund_10000 --streams 1 --no-cpu-baseline > gpurun_out/r1i/bench_scalar_fund.json 2>&1
cd /tmp; run_pmc scalar; cd $R
cp /tmp/lib_normal.so poselib_amd/lib/libposelib_amd.so
for f in gpurun_out/r1i/bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
for t in packed scalar; do python scripts/pmc_summary.py $(find gpurun_out/r1i -name "*counter_collection.csv" | grep _$t/) | grep "kernel\|k_score_queue" > gpurun_out/r1i/pmc_$t.md; cat gpurun_out/r1i/pmc_$t.md; done
