"""Sharding of independent estimation problems over the GPUs of one node.

Problems (image pairs) are independent, so the data path needs no collective: problem i is owned by rank
i mod world (round-robin), every rank runs its shard on its own GPU (one process per GPU), and a single
all-gather of fixed-size result records at the end hands rank 0 the complete answer — over RCCL ("nccl"
backend) on MI355X nodes, over gloo in the CPU tests.  Record layout (float64):
    [problem id, iterations, refinements, num_inliers, hypotheses, seconds, model (9 doubles, zero padded)]
Inlier masks stay on the owning rank unless asked for (they are N bytes per problem).
"""
from __future__ import annotations

import numpy as np


def dist_allgather(group=None, device=None):
    """The collective of Problem.run_sharded's exchange steps (ONE problem across the GPUs: every rank scores a share of each batch
    of iterations) as a torch.distributed all-gather: RCCL when `device` is this rank's GPU ("nccl" backend), gloo
    with device=None.  Returns allgather(send, recv) for numpy uint8 arrays, len(recv) == world * len(send)."""
    import torch
    import torch.distributed as dist

    def allgather(send, recv):
        world = dist.get_world_size(group)
        t = torch.from_numpy(np.array(send, dtype=np.uint8, copy=True))
        if device is not None:
            t = t.to(device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        recv[:] = torch.cat(out).cpu().numpy()

    return allgather


def thread_allgather(world: int):
    """In-process all-gather between `world` host threads (one per device, or several sharing one device in the
    tests): returns allgather_for(rank) -> allgather(send, recv)."""
    import threading

    barrier = threading.Barrier(world)
    slots = [None] * world

    def allgather_for(rank):
        def allgather(send, recv):
            slots[rank] = np.array(send, copy=True)
            barrier.wait()
            n = len(send)
            for r in range(world):
                recv[r * n:(r + 1) * n] = slots[r]
            barrier.wait()

        return allgather

    return allgather_for

RECORD_DOUBLES = 15


def owned(num_problems: int, rank: int, world: int):
    """Indices of the problems rank `rank` solves (round-robin)."""
    return list(range(rank, num_problems, world))


def pack_record(problem_id: int, info: dict, model) -> np.ndarray:
    r = np.zeros(RECORD_DOUBLES)
    r[0] = problem_id
    r[1] = info["iterations"]
    r[2] = info["refinements"]
    r[3] = info["num_inliers"]
    r[4] = info.get("hypotheses", 0)
    r[5] = info.get("seconds", 0.0)
    m = np.asarray(model, dtype=np.float64).reshape(-1)
    r[6:6 + m.size] = m
    return r


def gather_records(local: np.ndarray, num_problems: int, device=None):
    """All-gather the per-rank record arrays and return them ordered by problem id (every rank gets the
    full table).  `local` has shape (len(owned), RECORD_DOUBLES)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        table = local
    else:
        per_rank = (num_problems + world - 1) // world
        buf = torch.zeros((per_rank, RECORD_DOUBLES), dtype=torch.float64, device=device)
        buf[:, 0] = -1.0
        if local.shape[0]:
            buf[: local.shape[0]] = torch.as_tensor(local, dtype=torch.float64, device=device)
        out = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        table = torch.cat(out).cpu().numpy()
        table = table[table[:, 0] >= 0]
    order = np.argsort(table[:, 0], kind="stable")
    table = table[order]
    assert table.shape[0] == num_problems and (table[:, 0] == np.arange(num_problems)).all()
    return table
