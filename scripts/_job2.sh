cd $GRAFT_REPO_ROOT
SOAK_NMIN=7 SOAK_NMAX=40 timeout 1500 python scripts/parity_soak.py 150 31 2>&1 | tail -14
timeout 1500 python scripts/parity_soak.py 100 32 2>&1 | tail -6
