cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1l
for Q in 4 8 16; do for S in 8 16 32; do
  GPU_MAX_HW_QUEUES=$Q timeout 200 python bench.py --streams $S --steps 6 --no-cpu-baseline > gpurun_out/r1l/b_q${Q}_s$S.json 2>&1
  echo "Q=$Q S=$S $(tail -1 gpurun_out/r1l/b_q${Q}_s$S.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done; done
