"""Multi-process path on CPU: world_size 2 over gloo.  Each rank solves its round-robin shard of a small
batch of independent problems (with the CPU oracle standing in for the GPU — there is none here), the
records are all-gathered exactly as bench.py does over RCCL, and every rank must end up with the table a
single process produces."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

NUM_PROBLEMS = 7


def solve_problem(i):
    import oracle_lib as O
    from poselib_amd import sharding, synth

    d = synth.absolute_pose_scene(150 + 10 * i, 0.4, 3000 + i)
    pose, mask, st = O.estimate_absolute_pose(d["p2d"], d["p3d"], d["camera"], {"ransac": {"seed": i}})
    st["hypotheses"] = 0
    return sharding.pack_record(i, st, pose)


def worker(rank, world, port, q):
    from poselib_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.owned(NUM_PROBLEMS, rank, world)
    local = np.stack([solve_problem(i) for i in mine]) if mine else np.zeros((0, sharding.RECORD_DOUBLES))
    table = sharding.gather_records(local, NUM_PROBLEMS)
    dist.barrier()
    q.put((rank, mine, table))
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_process():
    from poselib_amd import sharding

    assert sharding.owned(7, 0, 2) == [0, 2, 4, 6] and sharding.owned(7, 1, 2) == [1, 3, 5]
    assert sorted(sharding.owned(4096, 3, 8) + sharding.owned(4096, 5, 8))[:4] == [3, 5, 11, 13]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = np.stack([solve_problem(i) for i in range(NUM_PROBLEMS)])
    single[:, 5] = 0
    for rank, mine, table in results:
        table = table.copy()
        table[:, 5] = 0  # wall time differs between runs
        assert (table == single).all(), rank


def exchange_worker(rank, world, port, q):
    from poselib_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ag = sharding.dist_allgather()
    out = []
    for nbytes in (8, 6928, 40000):  # the three message sizes of pl_ransac_run_sharded (count, header + 32, full lists)
        send = ((np.arange(nbytes) * (rank + 3)) % 251).astype(np.uint8)
        recv = np.zeros(world * nbytes, dtype=np.uint8)
        ag(send, recv)
        out.append(recv)
    q.put((rank, out))
    dist.destroy_process_group()


def test_exchange_step_of_the_within_problem_sharding_over_gloo():
    """sharding.dist_allgather is the collective Problem.run_sharded calls once per batch (RCCL on the GPU nodes):
    world 2 over gloo, every rank must receive rank r's bytes at offset r * len."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in results:
        for recv, nbytes in zip(out, (8, 6928, 40000)):
            for r in range(2):
                want = ((np.arange(nbytes) * (r + 3)) % 251).astype(np.uint8)
                assert (recv[r * nbytes:(r + 1) * nbytes] == want).all(), (rank, nbytes, r)


def test_thread_allgather():
    import threading

    from poselib_amd import sharding

    world = 3
    ag_for = sharding.thread_allgather(world)
    got = [None] * world

    def run(rank):
        for rnd in range(4):
            send = np.full(16, 10 * rnd + rank, dtype=np.uint8)
            recv = np.zeros(16 * world, dtype=np.uint8)
            ag_for(rank)(send, recv)
            got[rank] = recv
            assert all((recv[r * 16:(r + 1) * 16] == 10 * rnd + r).all() for r in range(world))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert all(g is not None for g in got)


def test_configs4_strong_split_of_4096_problems_over_a_node():
    """BASELINE configs[4] as worded (bench.py --batch-total 4096): ONE batch of 4096 problems sharded over the N ranks of a
    node, problem i on rank i mod N - every problem on exactly one rank, 512 per rank at N = 8, and the three problem kinds of
    the mix (i mod 3) evenly on every rank."""
    from poselib_amd import sharding

    for world in (1, 2, 4, 8):
        parts = [sharding.owned(4096, r, world) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(4096))
        assert all(len(p) == 4096 // world for p in parts)
        for p in parts:
            kinds = [sum(1 for i in p if i % 3 == k) for k in range(3)]
            assert max(kinds) - min(kinds) <= 1
