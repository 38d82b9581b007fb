# Round-3 evidence run (MI355X box): GPU parity suite, the contract bench line (+ the complete per-workload reports),
# rocprofv3 kernel-trace summaries of the same commands, PMC passes (instruction / busy counters, FETCH_SIZE, WRITE_SIZE in
# separate passes, never combined with other trace domains), the single-problem timeline, the grouped batch (configs[4]),
# the 5-point generator on a full device.  Summarised into profiles/r03_* by scripts/make_profiles_r03.py.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r03
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --detail-file $O/detail_default.json > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --mode streams --streams 1 --no-secondary --no-cpu-baseline --steps 10 --detail-file $O/detail_s1.json > $O/bench_s1.json 2> $O/bench_s1.err
cd /tmp && export TMPDIR=/tmp
Q="--no-parity --no-cpu-baseline --no-secondary --detail-file /tmp/_detail.json"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_default -o r -- python $R/bench.py $Q --steps 5 > $O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- python $R/bench.py $Q --mode streams --streams 1 --steps 5 > $O/prof_s1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_default -o k -- python $R/bench.py $Q --steps 3 > $O/kt_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_s1 -o k -- python $R/bench.py $Q --mode streams --streams 1 --steps 3 > $O/kt_s1.log 2>&1
for w in relpose_5000 fund_10000 hom_10000; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o r -- python $R/bench.py $Q --workload $w --mode streams --streams 1 --steps 3 > $O/prof_$w.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/profg_$w -o r -- python $R/bench.py $Q --workload $w --steps 3 > $O/profg_$w.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_batch -o r -- python $R/bench_batch.py --problems 4096 --streams 8 --steps 2 --warmup 3 --no-cpu-baseline > $O/prof_batch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_batch -o k -- python $R/bench_batch.py --problems 4096 --streams 8 --steps 2 --warmup 3 --no-cpu-baseline > $O/kt_batch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o r -- $R/scripts/exp/genbench 1600000 16 2 > $O/genbench.txt 2>&1
for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
  B1="python $R/bench.py $Q --workload $w --mode streams --streams 1 --steps 2 --warmup 1"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_$w -o p -- $B1 > $O/pmc_sq_$w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2_$w -o p -- $B1 > $O/pmc_sq2_$w.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm_$w -o p -- $B1 > $O/pmc_grbm_$w.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$w -o p -- $B1 > $O/pmc_fetch_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$w -o p -- $B1 > $O/pmc_write_$w.log 2>&1
done
cd $R
for d in prof_default prof_s1 prof_relpose_5000 prof_fund_10000 prof_hom_10000 profg_relpose_5000 profg_fund_10000 profg_hom_10000 prof_batch prof_gen; do f=$(find $O/$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/$d.md; done
python scripts/busy.py $(find $O/kt_default -name "*kernel_trace.csv") > $O/busy_default.txt
python scripts/busy.py $(find $O/kt_batch -name "*kernel_trace.csv") > $O/busy_batch.txt
python scripts/timeline.py $(find $O/kt_s1 -name "*kernel_trace.csv") k_sample_delta 30 > $O/timeline_s1.txt
for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
  python scripts/pmc_summary.py $(find $O/pmc_sq_$w $O/pmc_sq2_$w $O/pmc_grbm_$w $O/pmc_fetch_$w $O/pmc_write_$w -name "*counter_collection.csv") > $O/pmc_$w.md
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log; tail -c 700 $O/bench_default.json; head -6 $O/prof_default.md | cut -c1-160; head -3 $O/busy_default.txt; head -4 $O/pmc_hom_10000.md | cut -c1-400
