#!/usr/bin/env python
"""bench.py — scored RANSAC hypotheses / second on MI355X (BASELINE.json metric: P3P@5k corrs, 5pt@5k corrs).

    python bench.py --gpus N --steps K --warmup W
        N > 1 and no WORLD_SIZE in the environment: bench.py starts the N ranks ITSELF (re-exec under
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), one process per
        GPU; launched under torchrun by someone else it uses the ranks it finds.  It never prints n_gpus = 1 for
        --gpus 8: too few devices is an error.

Primary workload (value / ms_per_step; config.workload = "p3p_5000"): BASELINE.json configs[1] - P3P LO-RANSAC on
5000 synthetic 2D-3D correspondences, 70 % outliers, max_iterations = min_iterations = 100000 (with default options
PoseLib stops after ~10^3 iterations).  One "step" = a batch of 1536 independent, complete ransac_pnp
problems (sample -> P3P -> score all N -> LO -> final refinement -> inlier mask; different RANSAC seeds) on
correspondences already resident in HBM, handed to the library as ONE C-ABI call, pl_ransac_batch: the problems advance
in lock-step groups of 16 (one launch sequence per group and batch of iterations - the problem index is a grid dimension
of every kernel), 8 groups in flight on separate streams.  Every result is that of the single-problem call pl_ransac_run,
bit for bit (tests/test_gpu_group.py).  --mode streams is round 1's form of the same work: pl_ransac_run from 16 host
threads with one HIP stream each (about 15 % slower: ~15 small launches per problem compete on 16 hardware queues).
A hypothesis = one minimal-solver model scored against all N correspondences (ransac_impl.h:112-113).
The same K steps are then run for the metric's second half and the other BASELINE configs (config.secondary:
relpose_5000 = configs[2], fund_10000 / hom_10000 = configs[3]), each with its own value / roofline / parity / CPU
baseline.  Multi-GPU: independent image pairs, one set per rank (weak scaling, no data-path collective); RCCL is used
for the barrier and the final gather only.

The JSON line also carries
  parity       : the GPU results of RANSAC seeds 0..7 of the first timed step against the CPU oracle's runs of the same
                 seeds (the runs that also give cpu_baseline): iterations, refinements, hypotheses, inlier count and
                 mask must be identical, models within 1e-6.  A mismatch makes the exit code non-zero.
  roofline     : dominant kernel (k_score_mfma / k_score_mfma2 / k_score_queue).  bound = "valu_issue": the correspondences are
                 register/LDS-resident, so the kernel is limited by vector-ALU issue, not by HBM.
                 achieved = VALU wave-instructions per launch (PMC-measured instructions per (hypothesis, point chunk),
                 committed in profiles/pmc_traffic.json, x hypotheses x chunks of the launch) / launch duration (HIP
                 events, launching stream); peak = 1024 SIMDs x clock / the SIMD cycles per VALU instruction at the kernel's
                 own mix, its MFMAs charged at their 32.6 cycles in the matrix pipe (profiles/valu_mix.json, r06_overlap2.md;
                 frac_mfma_at_issue_cost: rounds 4 - 5's pricing at 8.4 issue cycles); frac < 1 by construction.  The SURVEY 8d "as if streamed" byte count is kept as algorithmic_hbm_x_peak, the PMC
                 traffic as hbm_frac_physical.
  cpu_baseline : oracle/_ref (the reference's own sources, kind "reference") and the oracle restatement ("port") timed
                 on the same workload on this box's host cores (rank 0, N = 1 only; single-threaded per problem like the
                 reference, several problems on separate cores at once, rate quoted per core).
"""
import argparse
import json
import os
import socket
import subprocess
import sys

# one HIP stream per in-flight problem: let the runtime map the 16 streams onto 16 hardware queues instead of the
# default 4 (must be set before the HIP runtime initialises).  Measured on MI355X, whole-job throughput of the
# default workload: 4 queues 3.4e8, 8 queues 3.9e8, 12 queues 4.0e8, 16 queues 4.2e8 hypotheses/s - two streams sharing
# a queue block each other behind their long single-CU kernels (LM, sampler orbit).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ITERATIONS = 100000
FOCAL = 1000.0
HBM_PEAK_GBS = 8000.0
SIMDS = 1024            # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
PEAK_CLOCK_GHZ = 2.4    # peak engine clock
# A wave64 vector instruction occupies its SIMD's issue port for 2.3 cycles (full-rate class: v_mul/add_f32, v_add/sub_u32, v_and/or/xor,
# v_mov, v_fma_f32 with <= 1 VGPR source) or 4.2 cycles (everything else: packed, fp64, DPP, 3-operand integer, compares, conversions);
# a v_mfma_f32_32x32x16_f16 costs 8.4 issue cycles (measured: profiles/r04_valu_issue.md, scripts/exp/valu_issue.cc, overlap.cc).  The
# peak of a kernel is priced at ITS instruction mix (profiles/valu_mix.json, scripts/valu_mix.py); the all-half-rate and all-full-rate
# readings are reported next to it.
VALU_PEAK_GINST_S = SIMDS * PEAK_CLOCK_GHZ / 4.0  # 614.4 G wave-instructions / s: every instruction at the nominal half rate (rounds 1-3)
VALU_PEAK_FULL_RATE_GINST_S = SIMDS * PEAK_CLOCK_GHZ / 2.0


def kernel_issue_cycles(kernel_name, per_hc, mfma_cost="mfma_pipe"):
    """SIMD cycles per VALU instruction of `kernel_name` at its own mix, the kernel's MFMAs included in the numerator - at their
    32.6 cycles in the matrix pipe (`mfma_pipe`, the default since round 6: profiles/r06_overlap2.md measured that the pipe's time
    does NOT run under the vector instructions that consume its results) or at the 8.4 cycles they hold the issue port
    (`mfma_issue`, the pricing of rounds 4 - 5) -, and how it was derived; None when profiles/valu_mix.json has no entry"""
    try:
        vm = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
    except Exception:
        return None, None
    k = vm.get(kernel_name) or vm.get(kernel_name.replace(" ", ""))
    cyc = vm.get("_class_cycles", {})
    if not k or not cyc:
        return None, None
    price = lambda m: (m["full"] * cyc["full"] + m["half"] * cyc["half"] + m["quarter"] * cyc["quarter"] + m["f64_trans"] * cyc["f64_trans"])
    whole = k["whole_kernel_static"]
    if per_hc and "hot_loop_hypotheses_per_iteration" in k and "hot_loop" in k:
        hl, rest = k["hot_loop"], k["outside_hot_loop_static"]
        f = k["hot_loop_iterations_per_chunk"] / k["hot_loop_hypotheses_per_iteration"]
        loop_valu = hl["valu"] * f
        loop_cycles = (price(hl) + hl["mfma"] * cyc.get(mfma_cost, cyc["mfma_issue"])) * f
        rest_valu = max(0.0, per_hc - loop_valu)
        total = loop_cycles + rest_valu * price(rest) / max(1, rest["valu"])
        return total / per_hc, "tile loop counted exactly (%.1f of %.1f instructions per hypothesis and chunk), the rest at the static mix outside the loop" % (loop_valu, per_hc)
    return (price(whole) + whole["mfma"] * cyc.get(mfma_cost, cyc["mfma_issue"])) / max(1, whole["valu"]), "whole-kernel static mix"
PARITY_SEEDS = 64   # problems of the first timed step checked against the oracle (seed j on scene j; VERDICT r4: 8 was 0.03 % of a step)
BASELINE_SEEDS = 8  # of those, the runs that are timed as the CPU baseline (oracle and reference sources: ~12 core-seconds each)
POSE_TOL = 1e-6
# workload -> (kind, N, outlier ratio, max_error [px], data seed, bytes per correspondence, problems per step and
#              in-flight stream, description)
WORKLOADS = {
    "p3p_5000": (0, 5000, 0.7, 12.0, 1001, 40, 96, "P3P LO-RANSAC (ransac_pnp), BASELINE configs[1]"),
    "relpose_5000": (1, 5000, 0.5, 1.0, 1002, 32, 24, "5-point LO-RANSAC (ransac_relpose), BASELINE configs[2]"),
    "fund_10000": (2, 10000, 0.5, 1.0, 1004, 32, 8, "7-point LO-RANSAC (ransac_fundamental), BASELINE configs[3]"),
    "hom_10000": (3, 10000, 0.5, 1.0, 1003, 32, 32, "4-point homography LO-RANSAC (ransac_homography), BASELINE configs[3]"),
}
SECONDARY = ["relpose_5000", "fund_10000", "hom_10000"]


def cpu_info():
    """host CPU of the box the baselines were timed on"""
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run ourselves."""
    if not os.environ.get("BENCH_SHARE_DEVICE") and "--rehearse-distributed" not in sys.argv:
        import torch

        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible; refusing to run a smaller job "
                  "under that label", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def model_vec(kind, model):
    if kind in (0, 1):
        return np.r_[np.asarray(model.q, dtype=np.float64), np.asarray(model.t, dtype=np.float64)]
    return np.asarray(model, dtype=np.float64).reshape(-1)


def model_diff(kind, got, ref):
    """rotation + translation difference for poses (BASELINE: 1e-6 each), sign-SENSITIVE normalised difference for F / H"""
    return max(model_diff_parts(kind, got, ref).values())


def model_diff_parts(kind, got, ref):
    """The components of the model difference.  Poses: dR = |R - R'|_F and dt = |t - t'|; relative poses additionally
    split dt into the direction d(t/|t|) and the length d|t|: the LM of the reference moves t in its tangent plane
    without renormalising (optim/relative.h:94-152), so |t| is a gauge the reference does not reproduce across its OWN
    builds (-O2 vs -O3 -march=native: up to 7e-5, tests/test_golden_vs_reference.py) while R and the direction agree to
    1e-13."""
    if kind in (0, 1):
        from poselib_amd import synth

        parts = {"dR": float(np.linalg.norm(synth.quat_to_rotmat(got[:4]) - synth.quat_to_rotmat(ref[:4]))),
                 "dt": float(np.linalg.norm(got[4:] - ref[4:]))}
        if kind == 1:
            ng, nr = float(np.linalg.norm(got[4:])), float(np.linalg.norm(ref[4:]))
            parts["dt_dir"] = float(np.linalg.norm(got[4:] / ng - ref[4:] / nr)) if ng > 0 and nr > 0 else float("inf")
            parts["dt_len"] = abs(ng - nr)
        return parts
    a, b = got / np.linalg.norm(got), ref / np.linalg.norm(ref)
    return {"dM": float(np.linalg.norm(a - b))}


class Ranks:
    """rank bookkeeping + the two collectives the bench needs (barrier, final gather)"""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # BENCH_DIST_BACKEND=gloo + BENCH_SHARE_DEVICE=1: rehearsal of the multi-rank path on a box with ONE GPU (all
        # ranks use device 0, the collectives run on CPU tensors).  Never used for reported numbers.
        self.backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        self.device_index = 0 if os.environ.get("BENCH_SHARE_DEVICE") else self.local_rank
        self.coll_device = "cuda" if self.backend == "nccl" else "cpu"
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self, cuda=True):
        if cuda:
            import torch

            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            if cuda:
                import torch

                torch.cuda.synchronize()

    def gather(self, values):
        """every rank contributes a vector of doubles; returns the (world, len) table on every rank"""
        import torch

        rec = torch.tensor(values, dtype=torch.float64, device=self.coll_device)
        if self.dist is None:
            return rec.cpu().numpy()[None]
        out = [torch.zeros_like(rec) for _ in range(self.world)]
        self.dist.all_gather(out, rec)
        return torch.stack(out).cpu().numpy()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def rehearse(args):
    """--rehearse-distributed: the launch / rendezvous / barrier / gather path of a multi-rank run WITHOUT any GPU work
    (tests/test_bench_distributed.py runs it with the gloo backend on CPU).  Prints a line that cannot be mistaken for a
    measurement: no metric, no value."""
    ranks = Ranks(args)
    ranks.barrier(cuda=False)
    table = ranks.gather([float(ranks.rank), float(os.getpid()), 1.0])
    ranks.barrier(cuda=False)
    if ranks.rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": ranks.world, "backend": ranks.backend,
                          "ranks_seen": [int(r) for r in table[:, 0]], "distinct_processes": len(set(table[:, 1])),
                          "metric": None, "value": None}))
    ranks.close()
    return 0


def run_workload(name, args, ranks, P, synth, pool_factory, primary):
    """K timed steps of one workload on this rank; returns the per-rank record and (rank 0) the data the report needs"""
    KIND, N_POINTS, OUTLIER_RATIO, MAX_ERROR_PX, DATA_SEED, BYTES_PER_CORR, PPS_PER_STREAM, DESCR = WORKLOADS[name]
    shard_problem = bool(args.shard_problem)
    pp = np.array([500.0, 500.0])
    data_rank = 0 if shard_problem else ranks.rank  # its own image pairs per rank; sharded problem: the same on every rank

    def make_scene(k):
        """scene k of this rank (normalised image coordinates): data seed DATA_SEED + 7919 k + rank"""
        if KIND == 0:
            sc = synth.absolute_pose_scene(N_POINTS, OUTLIER_RATIO, DATA_SEED + 7919 * k + data_rank)
            return (sc["p2d"] - pp) / FOCAL, sc["p3d"]
        gen = {1: synth.relative_pose_scene, 2: synth.fundamental_scene, 3: synth.homography_scene}[KIND]
        sc = gen(N_POINTS, OUTLIER_RATIO, DATA_SEED + 7919 * k + data_rank)
        return (sc["x1"] - pp) / FOCAL, (sc["x2"] - pp) / FOCAL

    thr = MAX_ERROR_PX / FOCAL
    S = 1 if shard_problem else max(1, args.streams)  # collectives of one process group must be issued in one order
    pool = pool_factory(S)
    # DISTINCT image pairs: problem j of a step works on scene j mod NS (the parity sample - RANSAC seeds 0..7 of the first
    # timed step - therefore covers scenes 0..7), so the 16 problems of a lock-step group are 16 different scenes, not 16
    # copies of one cache-resident set.
    # The front-end's O(N) pre-processing (robust.cc:40-46) and the upload are done once, outside the timed region.
    NS = 1 if shard_problem else max(1, args.scenes)
    scenes = [make_scene(k) for k in range(NS)]
    A, Bpts = scenes[0]
    probs = list(pool.map(lambda k: P.Problem(KIND, scenes[k][0], scenes[k][1]), range(NS)))  # SoA in HBM, resident from here on

    def prob_of(j):
        return probs[j % NS]  # (the parity seeds 0..7 too: seed j on scene j mod NS - the oracle below runs the same pairing)

    exchange = None
    if shard_problem and ranks.dist is not None:
        from poselib_amd import sharding

        exchange = sharding.dist_allgather(device=(f"cuda:{ranks.device_index}" if ranks.backend == "nccl" else None))

    def options(seed):
        return {"max_error": thr, "ransac": {"max_iterations": ITERATIONS, "min_iterations": ITERATIONS, "seed": seed}}

    def run_one(a):
        prob, seed = a
        if exchange is not None:
            return prob.run_sharded(options(seed), ranks.rank, ranks.world, exchange)
        return prob.run(options(seed))

    if primary and args.problems_per_step > 0:
        PPS = args.problems_per_step
    else:
        PPS = max(PARITY_SEEDS, PPS_PER_STREAM * S if primary else max(1, int(PPS_PER_STREAM * S * args.secondary_scale)))
    grouped = args.mode == "groups" and not shard_problem
    G, T = max(1, args.group_size), max(1, args.group_threads)

    # grouped mode (default): one C-ABI call per step, pl_ransac_batch - the PPS problems advance in lock-step groups of
    # G through ONE launch sequence per group and batch of iterations, T host threads inside the library work on
    # different groups.  The descriptors (options structs, output buffers) of every step are marshalled before the
    # timed region: what is timed is the library call.
    batches = {}
    if grouped:
        for sd in [1000 + w for w in range(args.warmup)] + list(range(args.steps)):
            batches[sd] = P.RansacBatch([prob_of(j) for j in range(PPS)], [options(sd * PPS + j) for j in range(PPS)])

    def step(seed):
        """a batch of PPS independent problems (RANSAC seeds seed * PPS + j).  Grouped: see above; --mode streams: worked
        through by S host threads / HIP streams, i.e. S problems in flight on this GPU at any time.
        Returns the pl_ransac_stats of the problems"""
        if grouped:
            batches[seed].run(T, G)
            return batches[seed].stats()
        return [info for _, info in pool.map(run_one, [(prob_of(j), seed * PPS + j) for j in range(PPS)])]

    def field(st, name):
        return st[name] if isinstance(st, dict) else getattr(st, name)

    for w in range(args.warmup):
        step(1000 + w)
    # the dominant kernel with the device to itself (4 problems one after the other, outside the timed region)
    ranks.barrier()
    solo_ms, solo_launches, solo_hyp = 0.0, 0, 0
    # (ONE worker thread for all twelve: every host thread has a context of its own, and a context's first launch runs on freshly
    # allocated buffers - 250 instead of 125 us for k_score_mfma<10>, scripts/exp/solo_launch_probe.py; taken from the pool's S
    # threads as they came, the four measured launches were first launches of some threads: 0.17 - 0.23 ms in round 6's first line)
    solo_pool = pool_factory(1)
    for j in range(12):  # the first eight only bring the clocks up (an idle device starts these launches ~10 % slower)
        _, info = solo_pool.submit(run_one, (probs[0], 7000 + j)).result()
        if j < 8:
            continue
        solo_ms += info["score_kernel_ms"]
        solo_launches += info["score_kernel_launches"]
        solo_hyp += info["hypotheses"]
    solo_pool.shutdown()
    ranks.barrier()
    t0 = time.perf_counter()
    hyp = nan_hyp = launches = 0
    kern_ms = 0.0
    last_inliers = 0
    first_streams = None
    for s in range(args.steps):
        res = step(s)
        if s == 0 and not grouped:
            first_streams = res[:PARITY_SEEDS]
        for st in res:
            hyp += field(st, "hypotheses")
            nan_hyp += field(st, "nan_hypotheses")
            kern_ms += field(st, "score_kernel_ms")
            launches += field(st, "score_kernel_launches")
        last_inliers = field(res[-1], "num_inliers")
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    # RANSAC seeds 0..7 of the first timed step: the ones the oracle runs below
    if grouped:
        first = batches[0].results(PARITY_SEEDS)
    else:
        first = [pool.submit(run_one, (prob_of(j), j)).result() for j in range(PARITY_SEEDS)] if first_streams is not None else None
    batches.clear()
    for pr in probs:
        pr.close()
    pool.shutdown()
    rec = [elapsed, float(hyp), kern_ms, float(launches), float(last_inliers), float(nan_hyp), solo_ms,
           float(solo_launches), float(solo_hyp)]
    ctx = {"A": A, "B": Bpts, "parity_scenes": [scenes[j % NS] for j in range(PARITY_SEEDS)], "thr": thr, "first": first, "PPS": PPS, "S": S, "kind": KIND, "n": N_POINTS,
           "bytes_per_corr": BYTES_PER_CORR, "descr": DESCR, "outliers": OUTLIER_RATIO, "max_error_px": MAX_ERROR_PX,
           "shard_problem": shard_problem, "grouped": grouped, "G": G, "T": T, "scenes": NS}
    return rec, ctx


def cpu_runs(kind, scenes, thr, iterations, nseeds, use_reference):
    """RANSAC seed j of the workload on scene j (j = 0..nseeds-1: what problem j of the first timed step ran) on the host:
    single-threaded per problem like the reference, the problems on separate cores (ctypes releases the GIL).  use_reference: oracle/_ref/libposelib_ref.so (the reference's
    own sources) instead of the oracle restatement.  Test infrastructure used as the CHECKER and the CPU baseline only."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib as O

    fn_name = {0: "ransac_pnp", 1: "ransac_relpose", 2: "ransac_fundamental", 3: "ransac_homography"}[kind]

    def one(seed):
        o = {"max_error": thr, "ransac": {"max_iterations": iterations, "min_iterations": iterations, "seed": seed}}
        return getattr(O, fn_name)(scenes[seed][0], scenes[seed][1], o)

    def run_all():
        workers = max(1, min(nseeds, (os.cpu_count() or 2) // 2))
        with ThreadPoolExecutor(max_workers=workers) as ex:
            return list(ex.map(one, range(nseeds))), workers

    t1 = time.perf_counter()
    if use_reference:
        import ref_lib

        with ref_lib.reference():
            out, workers = run_all()
    else:
        out, workers = run_all()
    return out, time.perf_counter() - t1, workers, fn_name


def parity_block(kind, gpu_first, cpu_out):
    checked = min(len(gpu_first), len(cpu_out))
    same = {"iterations": 0, "refinements": 0, "hypotheses": 0, "num_inliers": 0, "masks": 0}
    worst = {}
    for (gm, gi), (cm, cmask, cst) in zip(gpu_first[:checked], cpu_out[:checked]):
        for k in ("iterations", "refinements", "hypotheses", "num_inliers"):
            same[k] += int(gi[k] == cst[k])
        same["masks"] += int(bool((np.array(gi["inliers"], dtype=bool) == np.asarray(cmask, dtype=bool)).all()))
        ref = np.asarray(cm, dtype=np.float64).reshape(-1)
        for k, v in model_diff_parts(kind, model_vec(kind, gm), ref).items():
            worst[k] = max(worst.get(k, 0.0), v)
    top = max(worst.values()) if worst else 0.0
    ok = all(v == checked for v in same.values()) and top <= POSE_TOL and checked > 0
    out = {"checked": checked, "identical_iterations": same["iterations"], "identical_refinements": same["refinements"],
           "identical_hypotheses": same["hypotheses"], "identical_inlier_counts": same["num_inliers"],
           "identical_masks": same["masks"], "max_model_diff": top, "tolerance": POSE_TOL, "ok": bool(ok),
           "against": "oracle, RANSAC seeds 0..%d of the first timed step (seed j on scene j), %d iterations" % (checked - 1, ITERATIONS)}
    for k, v in worst.items():
        out["max_" + k] = v
    return out


def report_workload(name, table, ctx, args, world):
    """rank 0: value, roofline, parity, cpu baseline of one workload from the gathered per-rank records"""
    kind, n_points = ctx["kind"], ctx["n"]
    shard_problem = ctx["shard_problem"]
    t_max = float(table[:, 0].max())
    total_hyp = float(table[0, 1]) if shard_problem else float(table[:, 1].sum())
    value = total_hyp / t_max
    hyp0, k_ms, k_launch = float(table[0, 1]), float(table[0, 2]), int(table[0, 3])
    nan0 = float(table[0, 5])
    solo_ms, solo_launches, solo_hyp = float(table[0, 6]), int(table[0, 7]), float(table[0, 8])
    avg_launch_s = (k_ms / max(k_launch, 1)) * 1e-3
    solo_launch_s = (solo_ms / max(solo_launches, 1)) * 1e-3
    hyp_per_launch = hyp0 / max(k_launch, 1)
    alg_bytes_per_launch = hyp_per_launch * n_points * ctx["bytes_per_corr"]
    pmc = {}
    try:  # PMC-measured constants of the dominant kernel (rocprofv3 passes, committed under profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(name) or {}
    except Exception:
        pass
    kernel_name = pmc.get("kernel", "k_score_mfma<PG>" if kind == 0 else f"k_score_queue<{kind}, P>")
    traffic = pmc.get("traffic_bytes_per_launch")
    roof = {"bound": "valu_issue", "unit": "G wave-instructions/s", "peak": VALU_PEAK_GINST_S,
            "peak_basis": f"{SIMDS} SIMDs x {PEAK_CLOCK_GHZ} GHz / 4 cycles per wave64 VALU instruction",
            # the timed region runs the GROUP form of the kernel (blockIdx.z = problem of the group, same body); `achieved` /
            # `frac` are priced on the solo launches of the single-problem form measured right before the timed region
            "kernel": (kernel_name.replace("<", "_g<", 1) if ctx["grouped"] else kernel_name), "solo_kernel": kernel_name,
            "launches": k_launch,
            "launches_in_flight": ctx["T"] if ctx["grouped"] else ctx["S"],
            "avg_launch_ms": 1e3 * avg_launch_s, "solo_avg_launch_ms": 1e3 * solo_launch_s,
            "hypotheses_per_launch": hyp_per_launch,
            "solo_hypotheses_per_launch": solo_hyp / max(solo_launches, 1), "point_hypotheses_per_s": value * n_points,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "algorithmic_hbm_x_peak": (alg_bytes_per_launch / solo_launch_s / 1e9 / HBM_PEAK_GBS) if solo_launch_s > 0 else None,
            "traffic": traffic, "traffic_source": pmc.get("source"),
            "hbm_frac_physical": (traffic / solo_launch_s / 1e9 / HBM_PEAK_GBS) if (traffic and solo_launch_s > 0) else None}
    per_hc = pmc.get("valu_insts_per_hypothesis_chunk")
    chunk_pts = pmc.get("points_per_chunk")
    if per_hc and chunk_pts and solo_launch_s > 0:
        chunks = (n_points + chunk_pts - 1) // chunk_pts
        insts_solo = per_hc * (solo_hyp / max(solo_launches, 1)) * chunks
        achieved = insts_solo / solo_launch_s / 1e9
        insts_all = per_hc * hyp0 * chunks  # every scoring launch of the timed region on rank 0
        cyc_per_inst, cyc_basis = kernel_issue_cycles(kernel_name, per_hc)
        cyc_issue_only, _ = kernel_issue_cycles(kernel_name, per_hc, "mfma_issue")
        peak = SIMDS * PEAK_CLOCK_GHZ / cyc_per_inst if cyc_per_inst else VALU_PEAK_GINST_S
        roof.update({"achieved": achieved, "peak": peak, "frac": achieved / peak,
                     "issue_cycles_per_instruction": cyc_per_inst or 4.0,
                     "peak_basis": (f"{SIMDS} SIMDs x {PEAK_CLOCK_GHZ} GHz / {cyc_per_inst:.2f} issue cycles per wave64 VALU instruction at this kernel's mix "
                                    f"({cyc_basis}; class costs measured: profiles/r04_valu_issue.md; an MFMA at its 32.6 cycles in the matrix pipe: profiles/r06_overlap2.md)") if cyc_per_inst else roof["peak_basis"],
                     # rounds 4 - 5 priced an MFMA at the 8.4 cycles it holds the issue port (scripts/exp/overlap.cc: vector instructions
                     # that do not depend on it run under the rest of its 32.6 cycles); the scorers' vector instructions CONSUME the
                     # products, and profiles/r06_overlap2.md measured 203 cycles for what costs 123 + 120 apart - `frac` charges the pipe's time
                     "frac_mfma_at_issue_cost": (achieved * cyc_issue_only / (SIMDS * PEAK_CLOCK_GHZ)) if cyc_issue_only else None,
                     "frac_if_all_half_rate": achieved / VALU_PEAK_GINST_S, "frac_if_all_full_rate": achieved / VALU_PEAK_FULL_RATE_GINST_S,
                     "frac_basis": "the kernel with the device to itself (4 problems one after the other right before "
                                   "the timed region; HIP events on the launching stream) - in the timed region "
                                   "several launches share the device (grouped mode: every launch serves a group of "
                                   "problems, avg_launch_ms is one problem's share), see device_frac_timed_region",
                     "valu_insts_per_hypothesis_chunk": per_hc, "points_per_chunk": chunk_pts,
                     "valu_insts_per_launch": insts_solo,
                     "device_frac_timed_region": insts_all / 1e9 / peak / float(table[0, 0]),
                     "valu_busy_pmc": pmc.get("valu_busy"), "mfma_busy_pmc": pmc.get("mfma_busy")})
    else:
        roof.update({"achieved": None, "frac": None,
                     "frac_basis": "no PMC instruction count committed for this kernel (profiles/pmc_traffic.json)"})
    roof["note"] = ("the correspondences are register/LDS-resident: physical HBM traffic per launch (traffic, PMC) is "
                    "orders of magnitude below the SURVEY 8d 'as if every hypothesis streamed the set' count "
                    "(algorithmic_bytes_per_launch), so the binding roof is vector-ALU issue, not HBM")
    out = {"value": value, "unit": "hypotheses/s", "ms_per_step": 1e3 * t_max / args.steps,
           "problem": ctx["descr"], "correspondences": n_points, "outlier_ratio": ctx["outliers"],
           "max_iterations": ITERATIONS, "min_iterations": ITERATIONS, "max_error_px": ctx["max_error_px"],
           "problems_per_gpu_per_step": ctx["PPS"], "distinct_scenes": ctx["scenes"],
           "problems_in_flight_per_gpu": (ctx["G"] * ctx["T"]) if ctx["grouped"] else ctx["S"],
           "execution": (f"pl_ransac_batch: lock-step groups of {ctx['G']} problems (one launch sequence per group and batch "
                         f"of iterations), {ctx['T']} groups in flight") if ctx["grouped"] else
                        f"pl_ransac_run from {ctx['S']} host threads, one HIP stream each",
           "timed_region_s": t_max, "hypotheses_per_step": hyp0 / args.steps,
           "iterations_per_s": (1 if shard_problem else world) * ctx["PPS"] * args.steps * ITERATIONS / t_max,
           "nan_model_share": (nan0 / hyp0) if hyp0 else None,
           "nan_model_note": "share of the counted hypotheses whose model has a NaN entry (the reference's P3P emits "
                             "them for inconsistent samples; ransac_impl.h:112-113 counts them, utils.cc:36-65 scores "
                             "them at full price on the CPU, the device scorer skips them: no inliers possible)",
           "inliers_found": int(table[0, 4]), "roofline": roof}
    # ---- the oracle on RANSAC seeds 0..7 of the same problem: parity of the timed configuration + CPU baseline ----
    ok = True
    if not args.no_parity and not shard_problem:
        cpu_out, wall, workers, fn_name = cpu_runs(kind, ctx["parity_scenes"], ctx["thr"], ITERATIONS, PARITY_SEEDS, False)
        out["parity"] = parity_block(kind, ctx["first"], cpu_out)
        ok = out["parity"]["ok"]
        if world == 1 and not args.no_cpu_baseline:
            cpu_base = cpu_out[:BASELINE_SEEDS]
            hyp_c = sum(c[2]["hypotheses"] for c in cpu_base)
            sec_c = sum(c[2]["seconds"] for c in cpu_base)
            port = {"value": hyp_c / sec_c, "unit": "hypotheses/s", "cores": 1, "kind": "port",
                    "sample_short": f"oracle {fn_name}, seeds 0..{BASELINE_SEEDS - 1}, {ITERATIONS} it each, {hyp_c} hyp, "
                                    f"{sec_c:.1f} core-s, rate per core",
                    "sample": f"oracle {fn_name}, RANSAC seeds 0..{BASELINE_SEEDS - 1} of the same workload ({ITERATIONS} "
                              f"iterations each, {hyp_c} hypotheses, {sec_c:.1f} core-seconds; {workers} problems at a "
                              f"time on separate cores, {wall:.1f} s wall), g++ -O3 no -march, rate per core, "
                              f"{os.cpu_count()} host cores"}
            base = port
            if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libposelib_ref.so")):
                try:
                    ref_out, rwall, rworkers, _ = cpu_runs(kind, ctx["parity_scenes"], ctx["thr"], ITERATIONS, BASELINE_SEEDS, True)
                    # (ransac_* does not report its model count; the oracle's run of the same seed counts the same
                    # sample stream and agrees in iterations / inliers / mask, see agrees_with_port)
                    hyp_r = hyp_c
                    sec_r = sum(c[2]["seconds"] for c in ref_out)
                    same = sum(int(r[2]["iterations"] == c[2]["iterations"] and r[2]["num_inliers"] == c[2]["num_inliers"]
                                   and bool((r[1] == c[1]).all())) for r, c in zip(ref_out, cpu_base))
                    base = {"value": hyp_r / sec_r, "unit": "hypotheses/s", "cores": 1, "kind": "reference",
                            "sample_short": f"oracle/_ref (reference sources, g++ -O3, eigen shim) {fn_name}, seeds 0.."
                                            f"{BASELINE_SEEDS - 1}, {ITERATIONS} it each, {hyp_r} hyp, {sec_r:.1f} core-s, "
                                            f"rate per core; {same}/{len(ref_out)} runs = port",
                            "sample": f"oracle/_ref/libposelib_ref.so = the reference's own sources (robust/ransac.cc, "
                                      f"ransac_impl.h, estimators, solvers, utils.cc, bundle.cc, ...) compiled in place "
                                      f"against oracle/eigen_shim (real Eigen is not in this image): {fn_name}, RANSAC "
                                      f"seeds 0..{BASELINE_SEEDS - 1} of the same workload ({ITERATIONS} iterations each, "
                                      f"{hyp_r} hypotheses as counted by the oracle's runs of the same seeds, {sec_r:.1f} core-seconds; {rworkers} problems at a time on "
                                      f"separate cores, {rwall:.1f} s wall), g++ -O3 (the reference's Release flags, CMakeLists.txt:18-33), rate per core, {os.cpu_count()} host cores",
                            "agrees_with_port": f"{same}/{len(ref_out)} runs identical in iterations / inliers / mask",
                            "port": port}
                except Exception as e:  # the prebuilt file did not travel / does not load: the port alone
                    port["reference_unavailable"] = repr(e)
            out["cpu_baseline"] = base
    return out, ok


def run_batch_mixed(args, ranks, P, synth):
    """BASELINE configs[4]: a batch of independent image pairs (P3P / 5-point / homography cycling, N ~ U{500..5000},
    30-70 % outliers, DEFAULT options = the reference's ~10^3-iteration regime), host-resident inputs, one
    pl_estimate_batch call per step (problems of a kind advance in groups through one launch sequence; PCIe- and
    front-end-inclusive).  Problem i of the global batch lives on rank i mod world (sharding.owned)."""
    from poselib_amd import sharding

    # two ways to size the global batch: --batch-total T (BASELINE configs[4] as worded: 4096 problems SHARDED over the
    # N ranks - strong scaling, rank r gets the problems i = r mod N, 512 each at N = 8) or --batch-problems per rank
    # (weak scaling: the default of the N = 1 line, where both mean 4096 problems per call)
    total = args.batch_total if args.batch_total > 0 else args.batch_problems * ranks.world
    per_rank = (total + ranks.world - 1) // ranks.world
    mine = sharding.owned(total, ranks.rank, ranks.world)
    kinds = ("abs", "rel", "hom")
    problems = {}
    for i in mine:
        rs = synth.Stream(900000 + i)
        n = int(rs.uniform(1, 500, 5001)[0])
        outl = float(rs.uniform(1, 0.3, 0.7)[0])
        kind = kinds[i % 3]
        opt = {"ransac": {"seed": i}}
        if kind == "abs":
            d = synth.absolute_pose_scene(n, outl, 2000 + i)
            problems[i] = ("abs", d["p2d"], d["p3d"], d["camera"], opt)
        elif kind == "rel":
            d = synth.relative_pose_scene(n, outl, 2000 + i)
            problems[i] = ("rel", d["x1"], d["x2"], d["camera1"], d["camera2"], opt)
        else:
            d = synth.homography_scene(n, outl, 2000 + i, noise_px=0.3)
            problems[i] = ("hom", d["x1"], d["x2"], opt)
    batch = P.Batch([problems[i] for i in mine])  # descriptors marshalled once, outside the timed region
    # (three untimed calls at least: the workers' device arenas grow to the largest group each of them has served, and
    # which worker serves which group differs from call to call - measured: calls 1-2 take 200-400 ms, then 40 ms each)
    for _ in range(max(3, args.warmup)):
        batch.run(max_in_flight=args.batch_threads)
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):  # (nothing but the C-ABI call in the timed loop: reading 4096 stats records from Python costs ~5 ms)
        batch.run(max_in_flight=args.batch_threads)
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    hyp = args.steps * int(batch.stats()[2].sum())  # every step runs the same problems with the same seeds: the same hypotheses
    # what ONE of eight ranks sees when 4096 problems are sharded over a node: 512 problems per call (the first 512 of this
    # rank's list; N = 1 line only - VERDICT r4 next 4)
    small = None
    if ranks.world == 1 and len(mine) > 512:
        b512 = P.Batch([problems[i] for i in mine[:512]])
        for _ in range(max(3, args.warmup)):
            b512.run(max_in_flight=args.batch_threads)
        ts = time.perf_counter()
        reps512 = 4 * args.steps
        for _ in range(reps512):
            b512.run(max_in_flight=args.batch_threads)
        small = 512.0 * reps512 / (time.perf_counter() - ts)
        # ... and the same shard size with FOUR calls in flight from four host threads (pl_estimate_batch is re-entrant and leases
        # one of its worker pools per call since round 5): a rank that works through a stream of 512-problem shards
        import threading

        kf = 4
        shards = [P.Batch([problems[i] for i in mine[j * 512:(j + 1) * 512]]) for j in range(kf) if len(mine) >= (j + 1) * 512]
        small_inflight = None
        if len(shards) == kf:
            wk = max(2, args.batch_threads * 2 // 5)

            def spin(b, reps):
                for _ in range(reps):
                    b.run(max_in_flight=wk)

            for reps in (4, reps512):  # (an untimed round first: every pool's workers grow their arenas)
                ts = time.perf_counter()
                th = [threading.Thread(target=spin, args=(b, reps)) for b in shards]
                [t.start() for t in th]
                [t.join() for t in th]
                small_inflight = 512.0 * kf * reps / (time.perf_counter() - ts)
    table = ranks.gather([elapsed, float(hyp), float(len(mine))])
    if ranks.rank != 0:
        return None, True
    t_max = float(table[:, 0].max())
    out = {"value": float(table[:, 1].sum()) / t_max, "unit": "hypotheses/s",
           "problems_per_s": float(table[:, 2].sum()) * args.steps / t_max, "ms_per_step": 1e3 * t_max / args.steps,
           "problem": "BASELINE configs[4]: P3P / 5-point / homography cycling, N in [500,5000], 30-70 % outliers, default "
                      "options, host-resident inputs (PCIe- and front-end-inclusive), one pl_estimate_batch call per step",
           "problems_per_gpu_per_step": per_rank, "host_threads_per_gpu": args.batch_threads, "timed_region_s": t_max,
           "global_batch": total, "batch_scaling": "strong (--batch-total)" if args.batch_total > 0 else "weak (--batch-problems per rank)",
           "sharding": "problem i on rank i mod world; RCCL for the barrier and the final gather only"}
    if small is not None:
        out["problems_per_s_at_512_per_call"] = small
        if small_inflight is not None:
            out["problems_per_s_at_512_per_call_4_calls_in_flight"] = small_inflight
    ok = True
    if not args.no_parity:  # a sample of rank 0's problems against the oracle's front-ends
        from concurrent.futures import ThreadPoolExecutor

        import oracle_lib as O

        res = batch.results()
        want = 240  # (>= 200 problems; ~17 ms each on one core)
        sample = list(range(0, len(mine), max(1, len(mine) // want)))[:want]

        def one(j):
            pr = problems[mine[j]]
            t = time.perf_counter()
            if pr[0] == "abs":
                ref, mask, st = O.estimate_absolute_pose(pr[1], pr[2], pr[3], pr[4])
            elif pr[0] == "rel":
                ref, mask, st = O.estimate_relative_pose(pr[1], pr[2], pr[3], pr[4], pr[5])
            else:
                ref, mask, st = O.estimate_homography(pr[1], pr[2], pr[3])
            return mask, st, time.perf_counter() - t

        def run_all():  # single-threaded per problem like the reference, the problems on separate cores
            workers = max(1, min(len(sample), (os.cpu_count() or 2) // 2, 32))
            with ThreadPoolExecutor(max_workers=workers) as ex:
                return list(ex.map(one, sample)), workers

        cpu, workers = run_all()
        same = 0
        for j, (mask, st, _) in zip(sample, cpu):
            info = res[j][1]
            same += int(info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"]
                        and bool((np.array(info["inliers"], dtype=bool) == mask).all()))
        ok = same == len(sample)
        out["parity"] = {"checked": len(sample), "identical_iterations_inliers_masks": same, "ok": ok,
                         "against": "oracle estimate_* (complete front-ends) on a sample of the batch"}
        if ranks.world == 1 and not args.no_cpu_baseline:
            core_s = sum(c[2] for c in cpu)
            out["cpu_baseline"] = {"value": len(sample) / core_s, "unit": "problems/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle estimate_* on {len(sample)} problems of the batch, {workers} at a time on "
                                             f"separate cores, {core_s:.1f} core-seconds, rate per core ({os.cpu_count()} host cores)"}
            if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libposelib_ref.so")):
                try:
                    import ref_lib

                    with ref_lib.reference():
                        rcpu, rworkers = run_all()
                    rcore_s = sum(c[2] for c in rcpu)
                    agree = sum(int(a[1]["iterations"] == b[1]["iterations"] and a[1]["num_inliers"] == b[1]["num_inliers"]
                                    and bool((a[0] == b[0]).all())) for a, b in zip(rcpu, cpu))
                    out["cpu_baseline"] = {"value": len(sample) / rcore_s, "unit": "problems/s", "cores": 1, "kind": "reference",
                                           "port_value": len(sample) / core_s,
                                           "sample": f"oracle/_ref (the reference's own sources, g++ -O3, eigen shim) estimate_* on "
                                                     f"{len(sample)} problems of the batch, {rworkers} at a time on separate cores, "
                                                     f"{rcore_s:.1f} core-seconds, rate per core; {agree}/{len(sample)} runs = port"}
                except Exception as e:  # the reference build is optional: the port's figure stays
                    out["cpu_baseline"]["reference_error"] = str(e)[:200]
    return out, ok


def run_undistort_stage(args, ranks, P, synth):
    """BASELINE configs[3] "OPENCV camera model": the pre-processing stage in front of the homography / 7-point
    estimators (which take no camera) - 2 x 10000 pixels of an OPENCV camera through Camera::unproject on the device
    (pl_undistort_points; host-resident in and out, PCIe-inclusive), checked against the oracle's unproject."""
    cam = {"model": "OPENCV", "width": 1000, "height": 1000, "params": [1000.0, 1010.0, 500.0, 505.0, 0.04, -0.015, 8e-4, -6e-4]}
    d = synth.homography_scene(10000, 0.5, 1003 + ranks.rank)
    fx, fy, cx, cy = cam["params"][:4]
    pix = [synth.opencv_distort_pixels(np.c_[fx * (d[k][:, 0] - 500.0) / FOCAL + cx, fy * (d[k][:, 1] - 500.0) / FOCAL + cy],
                                       cam["params"]) for k in ("x1", "x2")]
    for _ in range(max(1, args.warmup)):
        out = [P.undistort_points(cam, p) for p in pix]
    ranks.barrier()
    t0 = time.perf_counter()
    reps = 10 * args.steps
    for _ in range(reps):
        out = [P.undistort_points(cam, p) for p in pix]
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    table = ranks.gather([elapsed])
    if ranks.rank != 0:
        return None, True
    t_max = float(table[:, 0].max())
    rep = {"value": ranks.world * reps * 2 * 10000 / t_max, "unit": "points/s", "calls": reps * 2, "points_per_call": 10000,
           "ms_per_call": 1e3 * t_max / (reps * 2),
           "problem": "configs[3] pre-processing: pixels of an OPENCV camera -> Camera::unproject (iterative inverse, "
                      "camera_models.cc:972-990) -> pixels of the distortion-free camera; host-resident in / out"}
    ok = True
    if not args.no_parity:
        import oracle_lib as O

        worst = 0.0
        for p, o in zip(pix, out):
            un = O.unproject(cam, p)
            worst = max(worst, float(np.abs(o - np.c_[fx * un[:, 0] + cx, fy * un[:, 1] + cy]).max()))
        ok = worst == 0.0
        rep["parity"] = {"checked_points": 20000, "max_abs_diff_px": worst, "ok": ok, "against": "oracle unproject + fx u + cx"}
    return rep, ok


def run_p3p_200_default(args, ranks, P, synth):
    """BASELINE configs[0]: estimate_absolute_pose (P3P, SIMPLE_PINHOLE) on 200 synthetic 2D-3D correspondences, 50 % outliers,
    DEFAULT options (robust.cc:36-126; the reference stops after ~1000 iterations) - the one BASELINE config that is a LATENCY
    workload: one front-end call after the other from host-resident inputs, milliseconds per call, next to the reference's own
    sources (oracle/_ref) on one host core.  A chain of ~15 small dispatches and three synchronisations does not beat one CPU
    core on a problem this small - the number is reported as it is."""
    n, calls = 200, max(64, 8 * args.steps)
    ds = [synth.absolute_pose_scene(n, 0.5, 4200 + 16 * ranks.rank + k) for k in range(8)]
    opt = lambda j: {"ransac": {"seed": j}}
    run = lambda j: P.estimate_absolute_pose(ds[j % 8]["p2d"], ds[j % 8]["p3d"], ds[j % 8]["camera"], opt(j))
    for j in range(8):
        run(j)
    ranks.barrier()
    t0 = time.perf_counter()
    outs = [run(j) for j in range(calls)]
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    table = ranks.gather([elapsed])
    if ranks.rank != 0:
        return None, True
    t_max = float(table[:, 0].max())
    rep = {"ms_per_call": 1e3 * t_max / calls, "calls": calls, "correspondences": n, "outlier_ratio": 0.5,
           "mean_iterations": float(np.mean([o[1]["iterations"] for o in outs])),
           "problem": "BASELINE configs[0]: estimate_absolute_pose, SIMPLE_PINHOLE, default options, one call at a time, host-resident inputs"}
    ok = True
    if not args.no_parity:
        import oracle_lib as O

        def cpu_all():
            t1 = time.perf_counter()
            res = [O.estimate_absolute_pose(ds[j % 8]["p2d"], ds[j % 8]["p3d"], ds[j % 8]["camera"], opt(j)) for j in range(calls)]
            return res, time.perf_counter() - t1

        res, t_port = cpu_all()
        same, worst = 0, 0.0
        for (img, info), (pose, mask, st) in zip(outs, res):
            same += int(info["iterations"] == st["iterations"] and info["num_inliers"] == st["num_inliers"] and info["refinements"] == st["refinements"]
                        and bool((np.array(info["inliers"], dtype=bool) == mask).all()))
            worst = max(worst, model_diff(0, np.r_[img.pose.q, img.pose.t], np.asarray(pose, dtype=np.float64)))
        ok = same == calls and worst <= POSE_TOL
        rep["parity"] = {"checked": calls, "identical_iterations_refinements_inliers_masks": same, "max_model_diff": worst, "ok": ok,
                         "against": "oracle estimate_absolute_pose, every call of the timed region"}
        if ranks.world == 1 and not args.no_cpu_baseline:
            rep["cpu_port_ms_per_call"] = 1e3 * t_port / calls
            try:
                import ref_lib

                if ref_lib.available():
                    with ref_lib.reference():
                        rres, t_ref = cpu_all()
                    rep["cpu_reference_ms_per_call"] = 1e3 * t_ref / calls
                    rep["cpu_reference_agrees"] = sum(int(a[2]["iterations"] == b[2]["iterations"] and a[2]["num_inliers"] == b[2]["num_inliers"]
                                                          and bool((a[1] == b[1]).all())) for a, b in zip(rres, res))
            except Exception as e:
                rep["reference_error"] = str(e)[:200]
    return rep, ok


def run_focal_estimators(args, ranks, P, synth):
    """the two focal-length estimators through their front-ends - estimate_absolute_pose with
    estimate_focal_length (ransac_pnpf, P3.5Pf) and estimate_shared_focal_relative_pose (6-point shared-focal solver) - on
    2000 correspondences, 40 % outliers, one problem after the other (a first device path: no problems in flight, no
    grouping).  Host-resident inputs, PCIe-inclusive.  Parity: every problem of the step against the oracle (decisions, mask,
    focal length bit for bit); CPU baseline: the oracle on the same problems (its solvers restate the reference's templates)."""
    n, reps = 2000, max(4, args.steps)
    da = [synth.absolute_pose_scene(n, 0.4, 7100 + 8 * ranks.rank + k) for k in range(4)]
    dr = [synth.relative_pose_scene(n, 0.4, 7000 + 8 * ranks.rank + k) for k in range(4)]
    oa = lambda s: {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": s}}
    orl = lambda s: {"max_error": 2.0, "ransac": {"seed": s}}
    pp = lambda d: d["camera1"]["params"][1:3]

    def run_abs(j):
        d = da[j % 4]
        return P.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], oa(j))

    def run_rel(j):
        d = dr[j % 4]
        return P.estimate_shared_focal_relative_pose(d["x1"], d["x2"], pp(d), orl(j))

    rep, ok = {}, True
    for name, run in (("pnpf_2000", run_abs), ("shared_focal_2000", run_rel)):
        for j in range(2):
            run(j)
        ranks.barrier()
        t0 = time.perf_counter()
        outs = [run(j) for j in range(reps)]
        ranks.barrier()
        elapsed = time.perf_counter() - t0
        # the same problems from 8 and from 16 host threads (one HIP stream and context each): what a caller's thread pool gets.
        # Every thread runs two problems before the clock starts (its context, stream and buffers exist); 16 streams are what the
        # device's hardware queues take - more threads get less (scripts/focal_threads.py, scripts/exp/focal_threads.cc).
        import threading

        def threaded(T, N):
            nxt, lock, bar = [0], threading.Lock(), threading.Barrier(T + 1)

            def work(i):
                run(i), run(i + 1)
                bar.wait()
                while True:
                    with lock:
                        j = nxt[0]
                        nxt[0] += 1
                    if j >= N:
                        break
                    run(j)

            th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
            for t in th:
                t.start()
            bar.wait()
            ranks.barrier()
            t1 = time.perf_counter()
            for t in th:
                t.join()
            ranks.barrier()
            return time.perf_counter() - t1

        n8, n16 = max(256, 16 * reps), max(768, 32 * reps)  # (0.1 - 0.2 s each: a shorter run measures the threads' start-up)
        elapsed8, elapsed16 = threaded(8, n8), threaded(16, n16)

        # the same problems through pl_estimate_batch (round 5, driver_focal_group.inc): groups of problems advance in lock-step
        # through ONE launch sequence.  Descriptors marshalled outside the clock (the camera of a pnpf item is in / out: a fresh batch
        # per run); the first `reps` results are compared with the single calls above - every field, bit for bit.
        def item(j):
            if name == "pnpf_2000":
                d = da[j % 4]
                return ("abs", d["p2d"], d["p3d"], d["camera"], oa(j))
            d = dr[j % 4]
            return ("shared_focal", d["x1"], d["x2"], pp(d), orl(j))

        nb = max(1024, 48 * reps)
        P.Batch([item(j) for j in range(nb)]).run(8)  # (the workers' buffers exist)
        bt = P.Batch([item(j) for j in range(nb)])
        ranks.barrier()
        t1 = time.perf_counter()
        bt.run(8)
        ranks.barrier()
        elapsed_b = time.perf_counter() - t1
        outs_b = bt.results()
        same_b = 0
        for (m1, i1), (m2, i2) in zip(outs, outs_b):
            c1, c2 = (m1.camera, m2.camera) if name == "pnpf_2000" else (m1.camera1, m2.camera1)
            same_b += bool(all(i1[k] == i2[k] for k in ("iterations", "refinements", "num_inliers", "hypotheses", "model_score"))
                           and np.array_equal(np.asarray(i1["inliers"]), np.asarray(i2["inliers"])) and list(c1.params) == list(c2.params)
                           and np.array_equal(np.r_[m1.pose.q, m1.pose.t], np.r_[m2.pose.q, m2.pose.t]))
        table = ranks.gather([elapsed, float(sum(o[1]["hypotheses"] for o in outs)), elapsed8, elapsed16, elapsed_b, float(same_b)])
        if ranks.rank != 0:
            continue
        t_max = float(table[:, 0].max())
        r = {"problems_per_s": ranks.world * reps / t_max, "ms_per_problem": 1e3 * t_max / reps,
             "problems_per_s_8_threads": ranks.world * n8 / float(table[:, 2].max()),
             "problems_per_s_16_threads": ranks.world * n16 / float(table[:, 3].max()),
             "problems_per_s_batch": ranks.world * nb / float(table[:, 4].max()), "batch_problems": nb,
             "batch_identical_to_single_calls": f"{int(table[:, 5].min())}/{reps}",
             "hyp_per_s": float(table[:, 1].sum()) / t_max, "problems": reps, "correspondences": n}
        ok = ok and int(table[:, 5].min()) == reps
        if not args.no_parity:
            import oracle_lib as O

            good, t_cpu = 0, 0.0
            for j, (model, info) in enumerate(outs):
                t1 = time.perf_counter()
                if name == "pnpf_2000":
                    d = da[j % 4]
                    pose, mask, st, cam = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], oa(j), return_camera=True)
                    # (k_lm_cam sums cost and normal equations in the reference's order at every n since round 4: bit for bit)
                    same = cam[0] == model.camera.params[0] and np.array_equal(pose, np.r_[model.pose.q, model.pose.t])
                else:
                    d = dr[j % 4]
                    pose, focal, mask, st = O.estimate_shared_focal_relative_pose(d["x1"], d["x2"], pp(d), orl(j))
                    same = focal == model.camera1.params[0] and np.array_equal(pose, np.r_[model.pose.q, model.pose.t])
                t_cpu += time.perf_counter() - t1
                good += bool(same and st["iterations"] == info["iterations"] and st["refinements"] == info["refinements"]
                             and np.array_equal(mask, np.asarray(info["inliers"], dtype=bool)))
            r["parity"] = {"problems": reps, "identical": good, "ok": good == reps,
                           "what": "against the ORACLE (whose minimal solvers are this project's formulations, not the reference's templates): "
                                   "iterations, refinements and inlier mask equal; pose and focal length bit for bit (both estimators: every sum of "
                                   "their refinements runs in the reference's order).  Agreement with the reference's own sources: "
                                   "device_vs_reference_sources"}
            r["cpu_port_problems_per_s"] = reps / t_cpu
            if ranks.world == 1 and not args.no_cpu_baseline:
                # the reference's own sources (oracle/_ref; the oracle's and the device's solvers restate its elimination templates since
                # round 6): every problem of the step - the CPU baseline and the measured agreement of the DEVICE with the reference
                # (decisions and mask; DESIGN 5: 600 of 600 random problems in the soak)
                import ref_lib

                if ref_lib.available():
                    agree, t_ref = 0, 0.0
                    with ref_lib.reference():
                        for j, (model, info) in enumerate(outs):
                            t1 = time.perf_counter()
                            if name == "pnpf_2000":
                                _, rmask, rst = O.estimate_absolute_pose(da[j % 4]["p2d"], da[j % 4]["p3d"], da[j % 4]["camera"], oa(j))
                            else:
                                _, _, rmask, rst = O.estimate_shared_focal_relative_pose(dr[j % 4]["x1"], dr[j % 4]["x2"], pp(dr[j % 4]), orl(j))
                            t_ref += time.perf_counter() - t1
                            agree += bool(rst["iterations"] == info["iterations"] and rst["refinements"] == info["refinements"]
                                          and np.array_equal(rmask, np.asarray(info["inliers"], dtype=bool)))
                    r["cpu_reference_problems_per_s"] = reps / t_ref
                    r["identical_to_reference"] = agree
                    r["parity"]["device_vs_reference_sources"] = f"{agree}/{reps} problems: same iterations, refinements and inlier mask as oracle/_ref"
            ok = ok and good == reps
        rep[name] = r
    return (rep if ranks.rank == 0 else None), ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="groups", choices=["groups", "streams"],
                    help="groups (default): every step is ONE pl_ransac_batch call, problems advance in lock-step groups; "
                         "streams: pl_ransac_run from --streams host threads (round 1's form)")
    ap.add_argument("--group-size", type=int, default=16, help="problems per lock-step group (pl_ransac_batch)")
    ap.add_argument("--group-threads", type=int, default=8, help="groups in flight (host threads inside pl_ransac_batch)")
    ap.add_argument("--streams", type=int, default=16,
                    help="--mode streams: independent problems in flight per GPU (one host thread + HIP stream each); "
                         "also sets the default step size (96 x streams problems)")
    ap.add_argument("--problems-per-step", type=int, default=0,
                    help="independent problems per step and GPU of the primary workload (default 96 x streams = 1536)")
    ap.add_argument("--scenes", type=int, default=64,
                    help="distinct synthetic image pairs per workload and GPU (problem j of a step works on scene j mod this; "
                         "1 = round 2's form: every problem of a step on the same correspondences)")
    ap.add_argument("--workload", default="p3p_5000", choices=sorted(WORKLOADS))
    ap.add_argument("--no-secondary", action="store_true", help="only the primary workload")
    ap.add_argument("--secondary-scale", type=float, default=1.0, help="scales the secondary workloads' step size")
    ap.add_argument("--shard-problem", action="store_true",
                    help="strong-scaling mode (SURVEY 8e-ii): ONE problem at a time, its iterations sharded over the "
                         "ranks (pl_ransac_run_sharded, one all-gather per batch); default is independent problems per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle runs (profiling)")
    ap.add_argument("--batch-problems", type=int, default=4096,
                    help="configs[4] leg (\"batch of 4096 independent image pairs\"): problems per GPU and step of the mixed "
                         "default-options batch (0: skip)")
    ap.add_argument("--batch-total", type=int, default=0,
                    help="BASELINE configs[4] as worded: this many problems in all, sharded over the ranks (problem i on rank i mod N; "
                         "strong scaling).  0 (default): --batch-problems per rank (weak scaling)")
    ap.add_argument("--batch-threads", type=int, default=10, help="host threads inside pl_estimate_batch (8 - 12 measure alike)")
    ap.add_argument("--detail-file", default="", help="complete per-workload reports (default gpurun_out/bench_detail.json)")
    ap.add_argument("--rehearse-distributed", action="store_true",
                    help="launch / rendezvous / gather path only, no GPU work, prints no measurement (CPU test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.rehearse_distributed:
        sys.exit(rehearse(args))

    import torch
    from concurrent.futures import ThreadPoolExecutor

    import poselib_amd as P
    from poselib_amd import synth

    ranks = Ranks(args)
    torch.cuda.set_device(ranks.device_index)
    P.set_device(ranks.device_index)

    def pool_factory(S):
        # every worker thread selects this rank's GPU before its first call (per-thread context: HIP stream + scratch)
        return ThreadPoolExecutor(max_workers=S, initializer=lambda: P.set_device(ranks.device_index))

    names = [args.workload] + ([] if (args.no_secondary or args.shard_problem) else [w for w in SECONDARY if w != args.workload])
    reports, all_ok = {}, True
    for i, name in enumerate(names):
        rec, ctx = run_workload(name, args, ranks, P, synth, pool_factory, primary=(i == 0))
        table = ranks.gather(rec)  # final gather over RCCL
        if ranks.rank == 0:
            reports[name], ok = report_workload(name, table, ctx, args, ranks.world)
            all_ok = all_ok and ok

    if not args.no_secondary and not args.shard_problem:
        rep, ok = run_undistort_stage(args, ranks, P, synth)
        if ranks.rank == 0:
            reports["opencv_undistort"] = rep
            names = names + ["opencv_undistort"]
            all_ok = all_ok and ok
    if not args.no_secondary and not args.shard_problem:
        rep, ok = run_p3p_200_default(args, ranks, P, synth)
        if ranks.rank == 0:
            reports["p3p_200_default"] = rep
            names = names + ["p3p_200_default"]
            all_ok = all_ok and ok
    focal_rep = None
    if not args.no_secondary and not args.shard_problem:
        focal_rep, ok = run_focal_estimators(args, ranks, P, synth)
        if ranks.rank == 0:
            reports["focal_estimators"] = focal_rep
            all_ok = all_ok and ok
    if args.batch_problems > 0 and not args.no_secondary and not args.shard_problem:
        rep, ok = run_batch_mixed(args, ranks, P, synth)
        if ranks.rank == 0:
            reports["batch_mixed"] = rep
            names = names + ["batch_mixed"]
            all_ok = all_ok and ok

    if ranks.rank == 0:
        prim = reports[args.workload]
        # ONE line, short enough to survive a truncated log (the driver keeps the tail of stdout and flat scalars of
        # `config`): every workload's numbers as SCALARS in `config`; the prose (what a NaN model is, how frac is
        # measured, what the CPU sample was) lives in DESIGN.md section 6, the complete per-workload reports go to
        # --detail-file.
        cfg = {"workload": args.workload,
               **{k: prim[k] for k in ("correspondences", "outlier_ratio", "max_iterations", "min_iterations", "max_error_px",
                                       "problems_per_gpu_per_step", "problems_in_flight_per_gpu", "distinct_scenes",
                                       "timed_region_s", "hypotheses_per_step", "iterations_per_s", "nan_model_share")},
               "sharding": "one problem over the ranks" if args.shard_problem else "independent problems per rank, RCCL: barrier + final gather",
               "lm_sums": {"1": "reference order at every n (k_lm_ordered)", "2": "tree beyond 256 correspondences, every estimator"}.get(
                   os.environ.get("POSELIB_AMD_LM_ORDERED", "0"), "poses / H: reference order up to 256 correspondences, tree beyond; F: reference order at every n (default)")}
        for n in names[1:]:
            r = reports[n]
            if n in WORKLOADS:
                cfg[n + "_hyp_per_s"] = r["value"]
                cfg[n + "_ms_per_step"] = r["ms_per_step"]
                cfg[n + "_frac"] = r["roofline"].get("frac")
                cfg[n + "_device_frac"] = r["roofline"].get("device_frac_timed_region")
                cfg[n + "_kernel"] = r["roofline"].get("kernel")
                cfg[n + "_issue_cycles_per_instruction"] = r["roofline"].get("issue_cycles_per_instruction")
                cfg[n + "_frac_if_all_half_rate"] = r["roofline"].get("frac_if_all_half_rate")
                cfg[n + "_counters"] = r["roofline"].get("traffic_source")
                if "parity" in r:
                    cfg[n + "_parity_ok"] = r["parity"]["ok"]
                    cfg[n + "_max_model_diff"] = r["parity"]["max_model_diff"]
                    for k in ("max_dR", "max_dt_dir", "max_dt_len"):
                        if k in r["parity"]:
                            cfg[n + "_" + k] = r["parity"][k]
                if "cpu_baseline" in r:
                    cfg[n + "_cpu_" + r["cpu_baseline"]["kind"] + "_hyp_per_s"] = r["cpu_baseline"]["value"]
            elif n == "p3p_200_default":
                cfg["p3p_200_default_ms_per_call"] = r["ms_per_call"]
                cfg["p3p_200_default_parity_ok"] = r.get("parity", {}).get("ok")
                for k in ("cpu_reference_ms_per_call", "cpu_port_ms_per_call"):
                    if k in r:
                        cfg["p3p_200_default_" + k] = r[k]
            elif n == "opencv_undistort":
                cfg["opencv_undistort_points_per_s"] = r["value"]
                cfg["opencv_undistort_parity_ok"] = r.get("parity", {}).get("ok")
            elif n == "batch_mixed":
                cfg["batch_mixed_problems_per_s"] = r["problems_per_s"]
                cfg["batch_mixed_global_batch"] = r["global_batch"]
                cfg["batch_mixed_scaling"] = r["batch_scaling"]
                if "problems_per_s_at_512_per_call" in r:  # what one of eight ranks sees of a 4096-problem batch
                    cfg["batch_mixed_512_problems_per_s"] = r["problems_per_s_at_512_per_call"]
                    if "problems_per_s_at_512_per_call_4_calls_in_flight" in r:
                        cfg["batch_mixed_512_x4_in_flight_problems_per_s"] = r["problems_per_s_at_512_per_call_4_calls_in_flight"]
                cfg["batch_mixed_hyp_per_s"] = r["value"]
                cfg["batch_mixed_parity_ok"] = r.get("parity", {}).get("ok")
                if "cpu_baseline" in r:
                    cfg["batch_mixed_cpu_" + r["cpu_baseline"]["kind"] + "_problems_per_s"] = r["cpu_baseline"]["value"]
        for n, r in (focal_rep or {}).items():
            cfg[n + "_problems_per_s"] = r["problems_per_s"]
            cfg[n + "_ms_per_problem"] = r["ms_per_problem"]
            cfg[n + "_problems_per_s_8_threads"] = r["problems_per_s_8_threads"]
            cfg[n + "_problems_per_s_16_threads"] = r["problems_per_s_16_threads"]
            cfg[n + "_problems_per_s_batch"] = r["problems_per_s_batch"]
            cfg[n + "_batch_identical_to_single_calls"] = r["batch_identical_to_single_calls"]
            cfg[n + "_parity_ok"] = r.get("parity", {}).get("ok")
            if "cpu_port_problems_per_s" in r:
                cfg[n + "_cpu_port_problems_per_s"] = r["cpu_port_problems_per_s"]
            if "cpu_reference_problems_per_s" in r:
                cfg[n + "_cpu_reference_problems_per_s"] = r["cpu_reference_problems_per_s"]
            if "identical_to_reference" in r:
                cfg[n + "_decisions_identical_to_reference"] = f"{r['identical_to_reference']}/{r['problems']}"
        if focal_rep:
            cfg["focal_oracle_vs_reference_soak"] = "589/600 random problems: identical decisions (profiles/r03_soak_focal_oracle_vs_reference.md)"
        short = lambda d, drop: {k: v for k, v in d.items() if k not in drop}
        out = {
            "metric": "scored RANSAC hypotheses/sec (P3P@5k corrs, 5pt@5k corrs)",
            "value": prim["value"],
            "unit": "hypotheses/s",
            "n_gpus": ranks.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": prim["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if args.shard_problem else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": cfg,
            "roofline": short(prim["roofline"], ("peak_basis", "frac_basis", "note", "traffic_source")),
            "parity": short(prim["parity"], ("against",)) if prim.get("parity") else None,
        }
        if "cpu_baseline" in prim:
            cb = prim["cpu_baseline"]
            out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": cb.get("sample_short", cb["sample"][:160]), **cpu_info()}
            if "port" in cb:
                out["cpu_baseline"]["port_value"] = cb["port"]["value"]
        detail = args.detail_file or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(detail), exist_ok=True)
            with open(detail, "w") as f:
                json.dump({"line": out, "reports": reports}, f, indent=1)
        except OSError:
            pass
        print(json.dumps(out))
    ranks.close()
    if ranks.rank == 0 and not all_ok:
        print("bench.py: PARITY MISMATCH between the GPU results and the oracle (see the parity blocks)", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
