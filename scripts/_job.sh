cd $GRAFT_REPO_ROOT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash scripts/gpu_check.sh
