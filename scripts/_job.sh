cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cp poselib_amd/lib/libposelib_amd.so /tmp/lib_base.so
cd /tmp && export TMPDIR=/tmp
for T in base pipe base pipe; do
  if [ $T != base ]; then cp $R/poselib_amd/lib/variants/lib_$T.so $R/poselib_amd/lib/libposelib_amd.so; else cp /tmp/lib_base.so $R/poselib_amd/lib/libposelib_amd.so; fi
  rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o p -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 3 > /dev/null 2>&1
  echo "== $T"; python $R/scripts/rocprof_summary.py $(find /tmp/pv -name "*.db" | head -1) | grep -E "k_score_mfma"
done
cp /tmp/lib_base.so $R/poselib_amd/lib/libposelib_amd.so
