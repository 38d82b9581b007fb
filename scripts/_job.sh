cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "relative or two_view or single_solver or chunk" 2>&1 | tail -3
for S in 1 16; do
timeout 300 python bench.py --no-cpu-baseline --workload relpose_5000 --streams $S --steps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('relpose streams $S value %.4g'%d['value'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o p -- python $R/bench.py --no-cpu-baseline --workload relpose_5000 --streams 1 --steps 2 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/pv -name "*.db" | head -1) | grep -E "k_generate|k_rel"
