cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --workload hom_10000 --streams 16 --steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('hom streams 16 value %.4g launch_ms %.3f'%(d['value'], d['roofline']['avg_launch_ms']))"
done
