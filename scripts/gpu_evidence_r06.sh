# Round-6 evidence run (MI355X box).  Outputs under gpurun_out/evidence_r06; scripts/make_profiles_r06.py turns them into profiles/r06_*.
#   a: GPU suite, the contract bench line + detail, one-problem-at-a-time line, configs[4] at 256 ... 4096 per call, the new group members
#      (OPENCV / PROSAC / warm starts), the focal-length estimators' timings
#   b: rocprofv3 kernel traces of the bench command (grouped and one problem at a time), of relpose_5000, of the focal estimators
#   p: PMC passes of the four scorers (-> profiles/pmc_traffic.json; before a)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r06
mkdir -p $O
PART=${1:-a}
cd /tmp && export TMPDIR=/tmp
Q="--no-parity --no-cpu-baseline --no-secondary --detail-file /tmp/_detail.json"
if [ "$PART" = "a" ]; then
  cd $R
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log
  timeout 900 python bench.py --steps 20 --warmup 5 --detail-file $O/detail_default.json > $O/bench_default.json 2> $O/bench_default.err
  timeout 300 python bench.py --mode streams --streams 1 --no-secondary --no-cpu-baseline --steps 10 --detail-file $O/detail_s1.json > $O/bench_s1.json 2> $O/bench_s1.err
  timeout 300 python scripts/batch_sweep.py 4096 8:0:3 10:0:3 12:0:3 > $O/batch_sweep.log 2>&1
  for n in 256 512 1024 2048; do timeout 200 python scripts/batch_sweep.py $n 9:0:3 2>&1 | grep threads >> $O/batch_sizes.log; done
  POSELIB_AMD_GROUP_JUMP=0 timeout 200 python scripts/batch_sweep.py 512 9:0:3 2>&1 | grep threads > $O/batch_512_doubling.log
  timeout 300 python scripts/batch_overlap.py 512 10 > $O/batch_overlap.log 2>&1
  POSELIB_AMD_GROUP_TIMING=1 timeout 200 python scripts/batch_sweep.py 512 9:0:3 2>&1 | tail -13 > $O/batch_timing_512.log
  timeout 400 python scripts/batch_cameras.py 1024 > $O/batch_cameras.md 2> $O/batch_cameras.err
  timeout 200 python scripts/time_focal_estimators.py 5 > $O/focal_timing.log 2>&1
  timeout 600 python scripts/focal_batch_bench.py 1024 2000 > $O/focal_batch.md 2> $O/focal_batch.err
  cat $O/pytest_gpu.log; tail -c 1200 $O/bench_default.json; tail -3 $O/bench_default.err; cat $O/batch_sizes.log $O/batch_512_doubling.log; cat $O/batch_cameras.md; tail -8 $O/focal_batch.md
elif [ "$PART" = "p" ]; then
  # PMC passes of the four scorers (one problem at a time); python scripts/make_profiles_r06.py pmc turns them into profiles/pmc_traffic.json,
  # which the bench of part a prices its roofline with: run p, refresh, then a
  cd /tmp
  for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
    B1="python $R/bench.py $Q --workload $w --mode streams --streams 1 --steps 2 --warmup 1"
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_$w -o p -- $B1 > $O/pmc_sq_$w.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2_$w -o p -- $B1 > $O/pmc_sq2_$w.log 2>&1
    timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm_$w -o p -- $B1 > $O/pmc_grbm_$w.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$w -o p -- $B1 > $O/pmc_fetch_$w.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$w -o p -- $B1 > $O/pmc_write_$w.log 2>&1
  done
  cd $R
  for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
    python scripts/pmc_summary.py $(find $O/pmc_sq_$w $O/pmc_sq2_$w $O/pmc_grbm_$w $O/pmc_fetch_$w $O/pmc_write_$w -name "*counter_collection.csv") > $O/pmc_$w.md
    head -4 $O/pmc_$w.md | cut -c1-330
  done
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
else
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_default -o r -- python $R/bench.py $Q --steps 5 > $O/prof_default.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- python $R/bench.py $Q --mode streams --streams 1 --steps 5 > $O/prof_s1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_default -o k -- python $R/bench.py $Q --steps 3 > $O/kt_default.log 2>&1
  for w in relpose_5000; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o r -- python $R/bench.py $Q --workload $w --mode streams --streams 1 --steps 3 > $O/prof_$w.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/profg_$w -o r -- python $R/bench.py $Q --workload $w --steps 3 > $O/profg_$w.log 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_focal -o r -- python $R/scripts/focal_threads.py 1 > $O/prof_focal.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b512 -o r -- python $R/scripts/batch_sweep.py 512 9:0:3 > $O/prof_b512.log 2>&1
  cd $R
  for d in prof_default prof_s1 prof_relpose_5000 profg_relpose_5000 prof_focal prof_b512; do f=$(find $O/$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/$d.md; done
  python scripts/busy.py $(find $O/kt_default -name "*kernel_trace.csv") > $O/busy_default.txt
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
  head -8 $O/prof_default.md | cut -c1-170; head -3 $O/busy_default.txt; head -12 $O/prof_focal.md | cut -c1-170
fi
