#!/usr/bin/env python
"""bench.py — scored RANSAC hypotheses / second on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload = "p3p_5000"): BASELINE.json configs[1] — P3P LO-RANSAC on 5000 synthetic 2D-3D
correspondences, 70 % outliers, max_iterations = 100000, with min_iterations = max_iterations so that the
loop really evaluates 100000 iterations (with default options PoseLib stops after ~10^3).
One "step" = one batch of 8 x S (default 128) independent, complete ransac_pnp problems (sample -> P3P -> score all N -> LO ->
final refinement -> inlier mask; different RANSAC seeds) worked through by S (= --streams, default 16) host
threads with one HIP stream each, i.e. S problems in flight on the GPU, on correspondences that are already
resident in HBM.  A hypothesis = one minimal-solver model scored
against all N correspondences (ransac_impl.h:112-113).  Multi-GPU: independent image pairs, one per rank
(weak scaling, no data-path collective); RCCL is used only for the barrier and the final gather.

The JSON line also carries
  roofline     : dominant kernel k_score_queue<ABS,5>; achieved = algorithmic bytes (hypotheses x N x 40 B, i.e. as if
                 every hypothesis streamed the fp64 correspondence set) / HIP-event duration of the launches.
                 NOTE the set is register/LDS resident, so physical HBM traffic (roofline.traffic, from the committed
                 PMC passes) is orders of magnitude lower and the real bound is the vector ALU;
                 frac may therefore exceed 1.
  cpu_baseline : the CPU oracle (port of the reference path, single thread like the reference) timed on the
                 same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys

# one HIP stream per in-flight problem: let the runtime map the 16 streams onto 16 hardware queues instead of the
# default 4 (must be set before the HIP runtime initialises).  Measured on MI355X, whole-job throughput of the
# default workload: 4 queues 3.4e8, 8 queues 3.9e8, 12 queues 4.0e8, 16 queues 4.2e8 hypotheses/s - two streams sharing
# a queue block each other behind their long single-CU kernels (LM, sampler orbit).  (bench_batch.py keeps the
# runtime default of 4, which is best for its short default-option problems.)
# (stress-tested with 32 queues and 40 streams in one process: no resource failures)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# throughput benchmark: every LO task on one workgroup (k_lm) instead of spread over several with one launch per LM
# iteration (k_lm2, the library's default for large homography / fundamental problems: 1.5-2.2x shorter single
# problems, but -10..-25 % throughput with 16 problems in flight).  No effect on the default workload.
if not ("--streams" in sys.argv and sys.argv[sys.argv.index("--streams") + 1:][:1] == ["1"]):  # (one at a time: keep it)
    os.environ.setdefault("POSELIB_AMD_LATENCY_MODE", "0")
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ITERATIONS = 100000
FOCAL = 1000.0
HBM_PEAK_GBS = 8000.0
# workload -> (problem kind, N, outlier ratio, max_error [px], data seed, bytes per correspondence, description)
WORKLOADS = {
    "p3p_5000": (0, 5000, 0.7, 12.0, 1001, 40, "P3P LO-RANSAC (ransac_pnp), BASELINE configs[1]"),
    "relpose_5000": (1, 5000, 0.5, 1.0, 1002, 32, "5-point LO-RANSAC (ransac_relpose), BASELINE configs[2]"),
    "fund_10000": (2, 10000, 0.5, 1.0, 1004, 32, "7-point LO-RANSAC (ransac_fundamental), BASELINE configs[3]"),
    "hom_10000": (3, 10000, 0.5, 1.0, 1003, 32, "4-point homography LO-RANSAC (ransac_homography), BASELINE configs[3]"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=16,
                    help="independent problems in flight per GPU (one host thread + HIP stream each)")
    ap.add_argument("--problems-per-step", type=int, default=0,
                    help="independent problems per step and GPU (default 8 x streams): the batch one step works through")
    ap.add_argument("--workload", default="p3p_5000", choices=sorted(WORKLOADS))
    ap.add_argument("--shard-problem", action="store_true",
                    help="strong-scaling mode (SURVEY 8e-ii): ONE problem at a time, its iterations sharded over the "
                         "ranks (pl_ransac_run_sharded, one all-gather per batch); default is independent problems per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iterations", type=int, default=ITERATIONS)
    args = ap.parse_args()

    import torch

    import poselib_amd as P
    from poselib_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BENCH_DIST_BACKEND=gloo + BENCH_SHARE_DEVICE=1: rehearsal of the multi-rank path on a box with ONE GPU (all
    # ranks use device 0, the collectives run on CPU tensors).  Never used for reported numbers.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    device_index = 0 if os.environ.get("BENCH_SHARE_DEVICE") else local_rank
    torch.cuda.set_device(device_index)
    P.set_device(device_index)
    use_dist = world > 1
    coll_device = "cuda" if backend == "nccl" else "cpu"
    if use_dist:
        import torch.distributed as dist

        dist.init_process_group(backend, rank=rank, world_size=world)

    KIND, N_POINTS, OUTLIER_RATIO, MAX_ERROR_PX, DATA_SEED, BYTES_PER_CORR, DESCR = WORKLOADS[args.workload]
    # one image pair per rank (independent problems; data seed + rank); pixels -> normalised image plane
    pp = np.array([500.0, 500.0])
    shard_problem = bool(args.shard_problem)
    data_rank = 0 if shard_problem else rank  # sharded problem: every rank holds the same correspondences
    if KIND == 0:
        scene = synth.absolute_pose_scene(N_POINTS, OUTLIER_RATIO, DATA_SEED + data_rank)
        A, Bpts = (scene["p2d"] - pp) / FOCAL, scene["p3d"]
    else:
        gen = {1: synth.relative_pose_scene, 2: synth.fundamental_scene, 3: synth.homography_scene}[KIND]
        scene = gen(N_POINTS, OUTLIER_RATIO, DATA_SEED + data_rank)
        A, Bpts = (scene["x1"] - pp) / FOCAL, (scene["x2"] - pp) / FOCAL
    # the front-end's O(N) pre-processing (robust.cc:40-46) is done once, outside the timed region
    from concurrent.futures import ThreadPoolExecutor

    thr = MAX_ERROR_PX / FOCAL
    S = 1 if shard_problem else max(1, args.streams)  # collectives of one process group must be issued in one order
    # every worker thread selects this rank's GPU before its first call (per-thread context: HIP stream + scratch)
    pool = ThreadPoolExecutor(max_workers=S, initializer=lambda: P.set_device(device_index))

    def make_problem(_):
        return P.Problem(KIND, A, Bpts)  # SoA in HBM, resident from here on

    probs = list(pool.map(make_problem, range(S)))

    exchange = None
    if shard_problem and use_dist:
        from poselib_amd import sharding

        exchange = sharding.dist_allgather(device=(f"cuda:{device_index}" if backend == "nccl" else None))

    def run_one(args_):
        prob, seed = args_
        opt = {"max_error": thr, "ransac": {"max_iterations": ITERATIONS, "min_iterations": ITERATIONS, "seed": seed}}
        if exchange is not None:
            return prob.run_sharded(opt, rank, world, exchange)
        return prob.run(opt)

    PPS = args.problems_per_step if args.problems_per_step > 0 else 8 * S

    def step(seed):
        """One step = a batch of PPS independent ransac_pnp problems (different RANSAC seeds) worked through by S
        host threads / HIP streams, i.e. S problems in flight on this GPU at any time."""
        return list(pool.map(run_one, [(probs[j % S], seed * PPS + j) for j in range(PPS)]))

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        step(1000 + w)
    # the dominant kernel with the device to itself (4 problems one after the other, outside the timed region): the
    # per-launch time the timed region reports is that of S launches sharing the device
    solo_ms, solo_launches = 0.0, 0
    for j in range(4):
        _, info = pool.submit(run_one, (probs[0], 7000 + j)).result()
        solo_ms += info["score_kernel_ms"]
        solo_launches += info["score_kernel_launches"]
    sync()
    t0 = time.perf_counter()
    hyp = 0
    kern_ms = 0.0
    launches = 0
    last = None
    for s in range(args.steps):
        for pose, info in step(s):
            hyp += info["hypotheses"]
            kern_ms += info["score_kernel_ms"]
            launches += info["score_kernel_launches"]
            last = (pose, info)
    sync()
    elapsed = time.perf_counter() - t0

    # final gather over RCCL: [elapsed, hypotheses, kernel ms, launches, inliers, pose(7)]
    model_flat = (list(last[0].q) + list(last[0].t) + [0.0, 0.0]) if KIND in (0, 1) else list(np.asarray(last[0]).reshape(-1))
    rec = torch.tensor([elapsed, float(hyp), kern_ms, float(launches), float(last[1]["num_inliers"])] + model_flat,
                       dtype=torch.float64, device=coll_device)
    if use_dist:
        allrec = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(allrec, rec)
        allrec = torch.stack(allrec).cpu().numpy()
    else:
        allrec = rec.cpu().numpy()[None]

    if rank == 0:
        t_max = float(allrec[:, 0].max())
        # sharded problem: every rank reports the job's total; independent problems: the ranks' sums add up
        total_hyp = float(allrec[0, 1]) if shard_problem else float(allrec[:, 1].sum())
        value = total_hyp / t_max
        k_ms = float(allrec[0, 2])
        k_launch = int(allrec[0, 3])
        hyp0 = float(allrec[0, 1])
        avg_launch_s = (k_ms / max(k_launch, 1)) * 1e-3
        alg_bytes_per_launch = (hyp0 / max(k_launch, 1)) * N_POINTS * BYTES_PER_CORR
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic, traffic_src, kernel_name, valu_busy = None, None, f"k_score_queue<{KIND}, P>", None
        try:  # PMC-measured HBM bytes per launch (rocprofv3 passes, committed under profiles/)
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload)
            if tr:
                traffic, traffic_src = tr["traffic_bytes_per_launch"], tr["source"]
                kernel_name, valu_busy = tr.get("kernel", kernel_name), tr.get("valu_busy")
        except Exception:
            pass
        out = {
            "metric": "scored RANSAC hypotheses/sec",
            "value": value,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if shard_problem else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "problem": DESCR, "correspondences": N_POINTS,
                       "outlier_ratio": OUTLIER_RATIO, "max_iterations": ITERATIONS, "min_iterations": ITERATIONS,
                       "max_error_px": MAX_ERROR_PX, "problems_per_gpu_per_step": PPS, "problems_in_flight_per_gpu": S,
                       "hypotheses_per_step": hyp0 / args.steps,
                       "iterations_per_s": (1 if shard_problem else world) * PPS * args.steps * ITERATIONS / t_max,
                       "sharding": "one problem over the ranks (iteration ranges, one all-gather per batch)" if shard_problem else "independent problems per rank",
                       "inliers_found": int(allrec[0, 4])},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "avg_launch_ms": 1e3 * avg_launch_s, "launches": k_launch,
                         "launches_in_flight": S, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "solo_avg_launch_ms": solo_ms / max(solo_launches, 1),
                         "solo_frac": (alg_bytes_per_launch / (1e-3 * solo_ms / max(solo_launches, 1)) / 1e9 / HBM_PEAK_GBS)
                         if solo_ms > 0 else None,
                         "point_hypotheses_per_s": value * N_POINTS, "valu_busy_pmc": valu_busy,
                         "note": f"algorithmic bytes = hypotheses x N x {BYTES_PER_CORR} B (SURVEY 8d); the set is "
                                 "register/LDS-resident, so frac > 1 is expected: the binding unit is the vector ALU "
                                 "(fp32 filter + fp64 exact pass, DESIGN.md 4); avg_launch_ms is the HIP-event time of "
                                 "one launch while launches_in_flight problems share the device (throughput-optimal, but every "
                                 "launch takes longer); solo_* is the same kernel with the device to itself, measured "
                                 "before the timed region"},
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as O

            # bounded sample of the same workload: whole problems (different RANSAC seeds) until ~12 s of CPU time
            cpu_fn = {0: O.ransac_pnp, 1: O.ransac_relpose, 2: O.ransac_fundamental, 3: O.ransac_homography}[KIND]
            t1 = time.perf_counter()
            cpu_hyp, cpu_sec, cpu_n = 0, 0.0, 0
            while cpu_n < 64 and time.perf_counter() - t1 < 12.0:
                o = {"max_error": thr, "ransac": {"max_iterations": args.cpu_iterations,
                                                  "min_iterations": args.cpu_iterations, "seed": cpu_n}}
                _, _, cst = cpu_fn(A, Bpts, o)
                cpu_hyp += cst["hypotheses"]
                cpu_sec += cst["seconds"]
                cpu_n += 1
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": cpu_hyp / cpu_sec, "unit": "hypotheses/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle {cpu_fn.__name__}, {cpu_n} problems of the same workload "
                                             f"({args.cpu_iterations} iterations each, {cpu_hyp} hypotheses, "
                                             f"{cpu_s:.1f} s wall), g++ -O3 no -march, 1 of {os.cpu_count()} host cores"}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    for pr in probs:
        pr.close()
    pool.shutdown()


if __name__ == "__main__":
    main()
