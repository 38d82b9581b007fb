cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('p3p 16 streams value %.4g'%d['value'])"
done
timeout 300 python bench.py --no-cpu-baseline --streams 1 --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('p3p 1 stream value %.4g'%d['value'])"
