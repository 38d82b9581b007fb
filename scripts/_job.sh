cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cp poselib_amd/lib/libposelib_amd.so /tmp/lib_base.so
cd /tmp && export TMPDIR=/tmp
for T in base q; do
  if [ $T != base ]; then cp $R/poselib_amd/lib/variants/lib_$T.so $R/poselib_amd/lib/libposelib_amd.so; fi
  for w in fund_10000 relpose_5000 hom_10000; do
  rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o p -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 2 --workload $w > /dev/null 2>&1
  echo "== $T $w"; python $R/scripts/rocprof_summary.py $(find /tmp/pv -name "*.db" | head -1) | grep -E "k_score_queue"
  done
done
cp /tmp/lib_base.so $R/poselib_amd/lib/libposelib_amd.so
