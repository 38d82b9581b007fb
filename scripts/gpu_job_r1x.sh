cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1x
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r1x/bench1.json 2>&1; tail -1 gpurun_out/r1x/bench1.json | cut -c1-200
BENCH_DIST_BACKEND=gloo BENCH_SHARE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r1x/bench2.json 2>&1; tail -2 gpurun_out/r1x/bench2.json | cut -c1-400
