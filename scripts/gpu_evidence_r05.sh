# Round-5 evidence run (MI355X box).  Outputs under gpurun_out/evidence_r05; scripts/make_profiles_r05.py turns them into profiles/r05_*.
#   part 1 (this file, argument "a"): GPU suite, the contract bench line + detail, batch trace (configs[4]) with occupancy, LM timings in
#           both summation modes, the micro-experiments (VALU issue table is committed separately: profiles/r05_valu_issue.md)
#   part 2 (argument "b"): kernel traces of the four throughput workloads (grouped and one problem at a time), PMC passes of the dominant kernels
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence_r05
mkdir -p $O
PART=${1:-a}
cd /tmp && export TMPDIR=/tmp
Q="--no-parity --no-cpu-baseline --no-secondary --detail-file /tmp/_detail.json"
if [ "$PART" = "a" ]; then
  cd $R
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log
  timeout 900 python bench.py --steps 20 --warmup 5 --detail-file $O/detail_default.json > $O/bench_default.json 2> $O/bench_default.err
  timeout 300 python bench.py --mode streams --streams 1 --no-secondary --no-cpu-baseline --steps 10 --detail-file $O/detail_s1.json > $O/bench_s1.json 2> $O/bench_s1.err
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_batch -o r -- python $R/bench_batch.py --problems 4096 --streams 10 --steps 4 --warmup 3 --no-cpu-baseline > $O/prof_batch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_batch -o k -- python $R/bench_batch.py --problems 4096 --streams 10 --steps 4 --warmup 3 --no-cpu-baseline > $O/kt_batch.log 2>&1
  cd $R
  timeout 300 python scripts/batch_sweep.py 4096 8:0:3 10:0:3 12:0:3 > $O/batch_sweep.log 2>&1
  for n in 256 512 1024 2048; do timeout 200 python scripts/batch_sweep.py $n 10:0:3 2>&1 | grep threads >> $O/batch_sizes.log; done
  timeout 300 python scripts/batch_overlap.py 512 10 > $O/batch_overlap.log 2>&1
  timeout 900 python scripts/soak_focal_device_vs_reference.py 300 > $O/soak_focal_device_vs_reference.md 2> $O/soak_focal.err
  (cd scripts/exp && for v in v3 v1; do if [ $v = v1 ]; then export POSELIB_AMD_REL_ROOTS_V1=1; else unset POSELIB_AMD_REL_ROOTS_V1; fi; timeout 120 ./genbench 1600000 16 3; done > $O/genbench_roots_ab.log 2>&1; unset POSELIB_AMD_REL_ROOTS_V1)
  (cd /tmp; for v in v3 v1; do if [ $v = v1 ]; then export POSELIB_AMD_REL_ROOTS_V1=1; else unset POSELIB_AMD_REL_ROOTS_V1; fi; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_gen_$v -o r -- $R/scripts/exp/genbench 1600000 16 3 > $O/prof_gen_$v.log 2>&1; f=$(find $O/prof_gen_$v -name "*.db" | head -1); [ -n "$f" ] && python $R/scripts/rocprof_summary.py $f > $O/prof_gen_$v.md; done; unset POSELIB_AMD_REL_ROOTS_V1)
  LM_PROFILE_VARIANT=lmprof timeout 300 python scripts/lm_profile.py > $O/lm_profile.md 2>/dev/null
  POSELIB_AMD_LM_ORDERED=1 timeout 300 python scripts/batch_sweep.py 4096 10:0:3 > $O/batch_sweep_ordered.log 2>&1
  POSELIB_AMD_GROUP_TIMING=1 timeout 200 python scripts/batch_sweep.py 4096 10:0:3 2>&1 | tail -3 > $O/batch_timing.log
  for l in truncated cauchy; do
    timeout 300 python scripts/time_lm.py $l > $O/time_lm_tree_$l.log 2>&1
    POSELIB_AMD_LM_ORDERED=1 timeout 300 python scripts/time_lm.py $l > $O/time_lm_ordered_$l.log 2>&1
  done
  timeout 100 scripts/exp/select_bench > $O/select_bench.log 2>&1
  timeout 120 python scripts/time_focal_estimators.py 5 > $O/focal_timing.log 2>&1
  timeout 300 python scripts/focal_threads.py 1 4 8 16 24 > $O/focal_threads.log 2>&1
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_focal -o r -- python $R/scripts/focal_threads.py 1 > $O/prof_focal.log 2>&1)
  f=$(find $O/prof_focal -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/prof_focal.md
  f=$(find $O/prof_batch -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/prof_batch.md
  python scripts/busy.py $(find $O/kt_batch -name "*kernel_trace.csv") 0.45 > $O/busy_batch.txt
  python scripts/chain_view.py $(find $O/kt_batch -name "*kernel_trace.csv") > $O/chain_batch.txt
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
  cat $O/pytest_gpu.log; tail -c 900 $O/bench_default.json; tail -3 $O/bench_default.err; head -3 $O/busy_batch.txt; cat $O/batch_sweep.log
else
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_default -o r -- python $R/bench.py $Q --steps 5 > $O/prof_default.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- python $R/bench.py $Q --mode streams --streams 1 --steps 5 > $O/prof_s1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_default -o k -- python $R/bench.py $Q --steps 3 > $O/kt_default.log 2>&1
  for w in relpose_5000 fund_10000 hom_10000; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o r -- python $R/bench.py $Q --workload $w --mode streams --streams 1 --steps 3 > $O/prof_$w.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/profg_$w -o r -- python $R/bench.py $Q --workload $w --steps 3 > $O/profg_$w.log 2>&1
  done
  for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
    B1="python $R/bench.py $Q --workload $w --mode streams --streams 1 --steps 2 --warmup 1"
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_$w -o p -- $B1 > $O/pmc_sq_$w.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2_$w -o p -- $B1 > $O/pmc_sq2_$w.log 2>&1
    timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm_$w -o p -- $B1 > $O/pmc_grbm_$w.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$w -o p -- $B1 > $O/pmc_fetch_$w.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$w -o p -- $B1 > $O/pmc_write_$w.log 2>&1
  done
  cd $R
  for d in prof_default prof_s1 prof_relpose_5000 prof_fund_10000 prof_hom_10000 profg_relpose_5000 profg_fund_10000 profg_hom_10000; do f=$(find $O/$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/$d.md; done
  python scripts/busy.py $(find $O/kt_default -name "*kernel_trace.csv") > $O/busy_default.txt
  for w in p3p_5000 relpose_5000 fund_10000 hom_10000; do
    python scripts/pmc_summary.py $(find $O/pmc_sq_$w $O/pmc_sq2_$w $O/pmc_grbm_$w $O/pmc_fetch_$w $O/pmc_write_$w -name "*counter_collection.csv") > $O/pmc_$w.md
  done
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
  head -8 $O/prof_default.md | cut -c1-170; head -3 $O/busy_default.txt; head -5 $O/pmc_p3p_5000.md | cut -c1-300
fi
# part "c" (separate call): steady-state occupancy of configs[4] - pl_estimate_batch calls back to back (scripts/batch_sweep.py: 3 warm-up
# + 4 timed calls, nothing else on the device), the last half of the kernel span; bench_batch.py's own trace ends with Python-side
# record marshalling and the gather, which is not the library's time
if [ "$PART" = "c" ]; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_sweep -o k -- python $R/scripts/batch_sweep.py 4096 10:0:3 > $O/kt_sweep.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sweep -o r -- python $R/scripts/batch_sweep.py 4096 10:0:3 > $O/prof_sweep.log 2>&1
  cd $R
  f=$(find $O/kt_sweep -name "*kernel_trace.csv" | head -1)
  # the window = the four TIMED calls at the end of the run (the warm-up calls grow the workers' arenas and can take seconds under the tracer):
  # the last 4 x (median call time printed by batch_sweep.py) of the kernel span
  FR=$(python - "$f" "$O/kt_sweep.log" <<'PY'
import csv, re, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
m = re.search(r"median of 4: ([0-9.]+) ms", open(sys.argv[2]).read())
ms = float(m.group(1)) if m else 52.0
print(f"{max(0.0, 1.0 - 4.0 * ms * 1e6 / (t1 - t0)):.6f}")
PY
)
  echo "window: last $(python -c "print(round((1-$FR)*100,2))") % of the kernel span = the four timed calls" > $O/busy_sweep.txt
  python scripts/busy.py $f $FR >> $O/busy_sweep.txt
  export PL_SHARES_CUT=$FR
  python scripts/chain_view.py $f > $O/chain_sweep.txt
  python - "$f" > $O/shares_sweep.txt <<'PY'
import csv, collections, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pl::", "")
import os
t0 = min(r["s"] for r in rows); t1 = max(r["e"] for r in rows); cut = t0 + float(os.environ.get("PL_SHARES_CUT", "0.5")) * (t1 - t0)
sel = [r for r in rows if r["s"] >= cut]
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in sel:
    tot[r["n"]] += (r["e"] - r["s"]) / 1e3; cnt[r["n"]] += 1
T = sum(tot.values())
print(f"the four timed calls at the end of the kernel span: {(t1 - cut) / 1e6:.1f} ms, {len(sel)} dispatches, summed kernel time {T / 1e3:.1f} ms")
print("| kernel | dispatches | summed ms | share of GPU time | avg us |\n|---|---|---|---|---|")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"| `{n}` | {cnt[n]} | {v / 1e3:.2f} | {100 * v / T:.1f} % | {v / cnt[n]:.1f} |")
PY
  f=$(find $O/prof_sweep -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/prof_sweep.md
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
  head -3 $O/busy_sweep.txt; grep -i "copyBuffer" $O/shares_sweep.txt; head -12 $O/shares_sweep.txt
fi
