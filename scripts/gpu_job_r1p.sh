cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1p
for pps in 16 32 64 128; do timeout 200 python bench.py --no-cpu-baseline --problems-per-step $pps > gpurun_out/r1p/b_$pps.json 2>&1; echo "pps=$pps $(tail -1 gpurun_out/r1p/b_$pps.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"; done
for S in 8 24; do timeout 200 python bench.py --no-cpu-baseline --streams $S --problems-per-step 64 > gpurun_out/r1p/b_s$S.json 2>&1; echo "S=$S $(tail -1 gpurun_out/r1p/b_s$S.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"; done
