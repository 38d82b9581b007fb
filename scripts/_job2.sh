cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -k "soak or redraws or absolute_pose_parity or score_and_refine" 2>&1 | tail -2
timeout 300 python scripts/latency_probe.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o p -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 3 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/pv -name "*.db" | head -1) | grep k_score_seq
