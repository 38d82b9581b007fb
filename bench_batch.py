#!/usr/bin/env python
"""bench_batch.py — BASELINE.json configs[4]: a batch of independent image-pair problems (mixed P3P / 5-point /
homography, N ~ U{500..5000}, 30-70 % outliers, default options = the reference's ~10^3-iteration regime),
sharded round-robin over the GPUs of one node (poselib_amd/sharding.py); each rank hands its whole shard to ONE
call of the batched C-ABI entry point pl_estimate_batch (S problems in flight per GPU, host threads inside the
library); RCCL only for the final gather of the result records.

    python bench_batch.py --problems 4096 --gpus 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench_batch.py --gpus 8

Unlike bench.py the correspondences start in HOST memory: every problem goes through the complete front-end
(pl_estimate_absolute_pose / _relative_pose / _homography: un-projection or normalisation, upload, RANSAC, LO,
final polish), so the figure is PCIe-inclusive.  One JSON line, same shape as bench.py's.
"""
import argparse
import json
import os
import sys
import time

# one HIP stream per worker of pl_estimate_batch: 16 hardware queues instead of the runtime's default 4 (read when HIP initialises -
# torch does that below, before the library's own default at load time could apply); see bench.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KINDS = ("abs", "rel", "hom")


def make_problem(i):
    from poselib_amd import synth

    rs = synth.Stream(900000 + i)
    n = int(rs.uniform(1, 500, 5001)[0])
    outl = float(rs.uniform(1, 0.3, 0.7)[0])
    kind = KINDS[i % 3]
    if kind == "abs":
        d = synth.absolute_pose_scene(n, outl, 2000 + i)
    elif kind == "rel":
        d = synth.relative_pose_scene(n, outl, 2000 + i)
    else:
        d = synth.homography_scene(n, outl, 2000 + i, noise_px=0.3)
    return kind, n, d


def solve(P, i, kind, d):
    opt = {"ransac": {"seed": i}}
    if kind == "abs":
        img, info = P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
        model = np.r_[img.pose.q, img.pose.t]
    elif kind == "rel":
        pose, info = P.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
        model = np.r_[pose.q, pose.t]
    else:
        H, info = P.estimate_homography(d["x1"], d["x2"], opt)
        model = H.reshape(-1)
    return info, model


def solve_oracle(O, i, kind, d):
    opt = {"ransac": {"seed": i}}
    if kind == "abs":
        return O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], opt)
    if kind == "rel":
        return O.estimate_relative_pose(d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
    return O.estimate_homography(d["x1"], d["x2"], opt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--problems", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=3, help="untimed calls (the workers' device arenas grow during the first two)")
    ap.add_argument("--streams", type=int, default=10, help="host threads inside pl_estimate_batch")
    ap.add_argument("--cpu-sample", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from concurrent.futures import ThreadPoolExecutor

    import poselib_amd as P
    from poselib_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist

        dist.init_process_group("nccl", rank=rank, world_size=world)

    mine = sharding.owned(args.problems, rank, world)
    problems = {i: make_problem(i) for i in mine}  # synthetic data, generated outside the timed region
    S = max(1, args.streams)
    P.set_device(local_rank)

    def batch_args(i):
        kind, n, d = problems[i]
        opt = {"ransac": {"seed": i}}
        if kind == "abs":
            return ("abs", d["p2d"], d["p3d"], d["camera"], opt)
        if kind == "rel":
            return ("rel", d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
        return ("hom", d["x1"], d["x2"], opt)

    shard_args = [batch_args(i) for i in mine]

    # the descriptors (pl_batch_item: host pointers, options, output buffers) are marshalled once, outside the timed
    # region; a timed step is the C-ABI call pl_estimate_batch and nothing else - host-resident inputs, PCIe-inclusive
    batch = P.Batch(shard_args)

    def run_shard():
        """the whole shard in ONE library call: problems of the same kind advance in groups through one launch sequence,
        S host threads inside the library work on groups concurrently"""
        batch.run(max_in_flight=S)

    def records():
        out = []
        for i, (model, info) in zip(mine, batch.results()):
            kind, n, d = problems[i]
            if kind == "abs":
                flat = np.r_[model.pose.q, model.pose.t]
            elif kind == "rel":
                flat = np.r_[model.q, model.t]
            else:
                flat = model.reshape(-1)
            out.append((sharding.pack_record(i, info, flat), info["hypotheses"], n))
        return out

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_shard()
    sync()
    t0 = time.perf_counter()
    hyp = 0
    res = None
    for _ in range(args.steps):  # (nothing but the C-ABI call in the timed loop)
        run_shard()
    sync()
    elapsed = time.perf_counter() - t0
    hyp = args.steps * int(batch.stats()[2].sum())  # every step runs the same problems with the same seeds
    res = records()

    local = np.stack([r[0] for r in res]) if res else np.zeros((0, sharding.RECORD_DOUBLES))
    table = sharding.gather_records(local, args.problems, device="cuda")  # the final gather (RCCL)
    tot = torch.tensor([elapsed, float(hyp)], dtype=torch.float64, device="cuda")
    if use_dist:
        alls = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(alls, tot)
        alls = torch.stack(alls).cpu().numpy()
    else:
        alls = tot.cpu().numpy()[None]

    if rank == 0:
        t_max = float(alls[:, 0].max())
        total_hyp = float(alls[:, 1].sum())
        out = {
            "metric": "scored RANSAC hypotheses/sec", "value": total_hyp / t_max, "unit": "hypotheses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batch_mixed", "problem": "BASELINE configs[4]: P3P / 5-point / homography cycling, "
                       "N in [500,5000], 30-70 % outliers, default options, host-resident inputs (PCIe-inclusive)",
                       "problems": args.problems, "problems_in_flight_per_gpu": S,
                       "problems_per_s": args.problems * args.steps / t_max,
                       "mean_iterations": float(table[:, 1].mean()), "mean_inliers": float(table[:, 3].mean())},
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as O

            sample = mine[: args.cpu_sample]
            t1 = time.perf_counter()
            ok = 0
            for i in sample:
                kind, n, d = problems[i]
                model, mask, st = solve_oracle(O, i, kind, d)
                ok += int(st["iterations"] == int(table[i, 1]) and st["num_inliers"] == int(table[i, 3]))
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": len(sample) / cpu_s, "unit": "problems/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle estimate_* on the first {len(sample)} problems of the batch "
                                             f"({cpu_s:.1f} s, 1 of {os.cpu_count()} host cores); "
                                             f"{ok}/{len(sample)} agree with the GPU result in iterations and inliers"}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
