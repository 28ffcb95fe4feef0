"""Multi-GPU proving of one execution: shards are independent proofs (`Shard::shard`,
/root/reference/src/lair/execute.rs:186-216), one process per GPU, shards dealt to the ranks by work
(`assign_shards_balanced`; `assign_shards` is the plain round robin).

The only cross-shard data: every shard's transcript observes every shard's main-trace root before
any challenge is drawn, and the verifier's grand-sum check needs the sum of all chips' cumulative sums.  Both are a
few dozen bytes per shard: one all-gather and one all-reduce per proof over `torch.distributed` (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  RCCL has no modular reduction: extension-field sums travel
as 4 x int64 (addends < 2^31, so thousands of shards cannot overflow) and are reduced mod p locally.
"""
from __future__ import annotations

import numpy as np

from .field import P


def assign_shards(n_shards: int, world: int, rank: int) -> list[int]:
    """Shard indices proven by `rank` (round robin, the layout the reference's shard loop would map onto ranks)."""
    return [s for s in range(n_shards) if s % world == rank]


def assign_shards_balanced(costs, world: int) -> list[list[int]]:
    """Shards to ranks by estimated work (longest processing time first; every rank gets the same number of shards, the
    all-gather's shape): `Shard::shard` cuts every chip at the same row count, so the first shards of an execution hold all
    its chips and the last ones only the tallest -- round robin would leave rank 0 with twice the average.  Returns
    [shard indices of rank 0, of rank 1, ...], each ascending; identical on every rank (ties broken by index).
    len(costs) must be a multiple of world."""
    n = len(costs)
    if n % world:
        raise ValueError(f"{n} shards do not divide over {world} ranks")
    per_rank = n // world
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for s in sorted(range(n), key=lambda i: (-float(costs[i]), i)):
        r = min((r for r in range(world) if len(out[r]) < per_rank), key=lambda r: (load[r], r))
        out[r].append(s)
        load[r] += float(costs[s])
    return [sorted(x) for x in out]


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def exchange_roots(local_roots, device="cpu", shard_indices=None):
    """local_roots: [k][8] roots of this rank's shards (every rank passes the same k; pad with zeros otherwise).
    Returns the roots of all shards ordered by shard index: the round-robin layout of `assign_shards` undone, or -- with
    `shard_indices` (this rank's shard numbers, same order as local_roots; any assignment, e.g. assign_shards_balanced) -- by the
    indices that travel with the roots."""
    import torch

    local = torch.tensor(np.asarray(local_roots, dtype=np.int64).reshape(-1, 8), device=device)
    dist = _dist()
    if shard_indices is not None:
        idx = torch.tensor(np.asarray(shard_indices, dtype=np.int64).reshape(-1, 1), device=device)
        rec = torch.cat([idx, local], dim=1).contiguous()  # [k][9]
        if dist is None:
            rows = rec.cpu().tolist()
        else:
            out = torch.zeros((dist.get_world_size(),) + tuple(rec.shape), dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(out.view(-1), rec.view(-1))
            rows = [r for per_rank in out.cpu().tolist() for r in per_rank]
        rows.sort(key=lambda r: r[0])
        if [r[0] for r in rows] != list(range(len(rows))):
            raise ValueError("shard indices of the ranks are not a partition of 0 .. n-1")
        return [[int(x) for x in r[1:]] for r in rows]
    if dist is None:
        return [[int(x) for x in r] for r in local.cpu().tolist()]
    world = dist.get_world_size()
    out = torch.zeros((world,) + tuple(local.shape), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out.view(-1), local.view(-1))
    gathered = out.cpu().tolist()  # [rank][k][8]
    k = local.shape[0]
    return [[int(x) for x in gathered[s % world][s // world]] for s in range(k * world)]


def reduce_cumulative_sums(local_sums, device="cpu"):
    """local_sums: iterable of extension-field elements (4 canonical lanes each): the cumulative sums of every chip of
    every shard this rank proved.  Returns the machine-wide total (4 lanes, reduced mod p) on every rank; the proof
    set is consistent iff it is zero."""
    import torch

    acc = np.zeros(4, dtype=np.int64)
    for s in local_sums:
        acc += np.asarray(s, dtype=np.int64)
        acc %= P
    t = torch.from_numpy(acc).to(device)
    dist = _dist()
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(int(x) % P for x in t.cpu().tolist())
