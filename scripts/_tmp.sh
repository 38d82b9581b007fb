cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mfma
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 200 python bench.py --streams 1 --steps 5 --no-cpu-baseline > gpurun_out/mfma/s1.json 2>&1; tail -1 gpurun_out/mfma/s1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mfma s1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
POSELIB_AMD_NO_MFMA=1 timeout 200 python bench.py --streams 1 --steps 5 --no-cpu-baseline > gpurun_out/mfma/s1_no.json 2>&1; tail -1 gpurun_out/mfma/s1_no.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queue s1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/mfma/s16.json 2>&1; tail -1 gpurun_out/mfma/s16.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mfma s16', d['value'], d['ms_per_step'])"
