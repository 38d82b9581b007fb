cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fundamental or two_view or single_solver or chunk" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o p -- python $R/bench.py --no-cpu-baseline --workload fund_10000 --streams 1 --steps 2 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/pv -name "*.db" | head -1) | grep -E "k_generate|k_rel"
