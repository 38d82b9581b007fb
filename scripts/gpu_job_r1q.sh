cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1q
for i in 1 2 3 4; do timeout 200 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r1q/b_$i.json 2>&1; echo "run=$i $(tail -1 gpurun_out/r1q/b_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"; done
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --steps 10 --problems-per-step 128 > gpurun_out/r1q/c_$i.json 2>&1; echo "pps128 run=$i $(tail -1 gpurun_out/r1q/c_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"; done
