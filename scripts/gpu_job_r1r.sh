cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1r
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
for S in 8 16 32; do timeout 300 python bench_batch.py --problems 2048 --streams $S --no-cpu-baseline > gpurun_out/r1r/batch_s$S.json 2>&1; echo "S=$S $(tail -1 gpurun_out/r1r/batch_s$S.json | cut -c1-420)"; done
