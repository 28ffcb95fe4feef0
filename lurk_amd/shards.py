"""Multi-GPU proving of one execution: shards are independent proofs (`Shard::shard`,
/root/reference/src/lair/execute.rs:186-216), one process per GPU, shard s on rank s mod world.

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


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def exchange_roots(local_roots, device="cpu"):
    """local_roots: [k][8] roots of this rank's shards (every rank passes the same k; pad with zeros otherwise).
    Returns the roots of all shards ordered by shard index (rank-major round robin undone)."""
    import torch

    local = torch.tensor(np.asarray(local_roots, dtype=np.int64).reshape(-1, 8), device=device)
    dist = _dist()
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
