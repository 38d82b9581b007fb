# Round-1 evidence run: parity suite, the contract bench line, rocprofv3 kernel-trace summaries of the same
# commands, PMC passes (VALU/issue counters, FETCH_SIZE, WRITE_SIZE in separate passes), multi-stream occupancy.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --streams 1 > $O/bench_s1.json 2> $O/bench_s1.err
for w in relpose_5000 fund_10000 hom_10000; do
  timeout 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 300 python bench_batch.py --problems 4096 > $O/bench_batch.json 2> $O/bench_batch.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_default -o r -- python $R/bench.py --no-cpu-baseline > $O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- python $R/bench.py --streams 1 --no-cpu-baseline > $O/prof_s1.log 2>&1
for w in relpose_5000 fund_10000 hom_10000; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o r -- python $R/bench.py --workload $w --streams 1 --steps 3 --no-cpu-baseline > $O/prof_$w.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_default -o k -- python $R/bench.py --no-cpu-baseline > $O/kt_default.log 2>&1
B1="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o p -- $B1 > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- $B1 > $O/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm -o p -- $B1 > $O/pmc_grbm.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B1 > $O/pmc_write.log 2>&1
for w in fund_10000; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$w -o p -- $B1 --workload $w > $O/pmc_fetch_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$w -o p -- $B1 --workload $w > $O/pmc_write_$w.log 2>&1
done
cd $R
for d in prof_default prof_s1 prof_relpose_5000 prof_fund_10000 prof_hom_10000; do f=$(find $O/$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/$d.md; done
python scripts/busy.py $(find $O/kt_default -name "*kernel_trace.csv") > $O/busy_default.txt
python scripts/pmc_summary.py $(find $O/pmc_sq $O/pmc_sq2 $O/pmc_grbm $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_p3p.md
python scripts/pmc_summary.py $(find $O/pmc_fetch_fund_10000 $O/pmc_write_fund_10000 -name "*counter_collection.csv") > $O/pmc_fund.md
find $O -name "*.db" -size +8M -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
cat $O/pytest_gpu.log; cat $O/bench_default.json; head -8 $O/prof_default.md; cat $O/busy_default.txt | head -8
