cd $GRAFT_REPO_ROOT
export BENCH_DIST_BACKEND=gloo BENCH_SHARE_DEVICE=1
timeout 300 python bench.py --no-cpu-baseline --streams 1 --steps 3 2>&1 | tail -1 | cut -c1-330
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --shard-problem --no-cpu-baseline 2>&1 | tail -2 | cut -c1-900
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
