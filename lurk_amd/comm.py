"""`Comm`: the RCCL communicator of a multi-GPU proof behind the lurkhip C ABI (csrc/comm.cpp, include/lurkhip.h).

The two collectives of a sharded proof -- the all-gather of (shard index, main-trace root) records every shard's transcript
observes, and the all-reduce of the chips' cumulative sums the verifier's grand-sum check needs
(/root/reference/src/lair/execute.rs:186-241, /root/reference/src/lair/lair_chip.rs:104-139) -- run inside the library, on the
context's stream.  The host only distributes the 128-byte communicator id: a Rust host over whatever started its ranks, this
mirror over `torch.distributed.broadcast_object_list` (any backend; the id is host data)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context, as_u32

ID_BYTES = 128
RECORD_WORDS = 9


def unique_id() -> bytes:
    buf = (C.c_uint8 * ID_BYTES)()
    N.check(N.lib.lurkhip_comm_unique_id(C.addressof(buf)))
    return bytes(buf)


def library() -> str:
    """The librccl the C ABI bound (LURKHIP_RCCL_LIB, else the copy the process has already mapped -- PyTorch's --, else the
    system's); raises with the loader's message when there is none."""
    p = N.lib.lurkhip_comm_library()
    if p is None:
        raise RuntimeError(N.last_error(None))
    return p.decode()


class Comm:
    def __init__(self, ctx: Context, uid: bytes, rank: int, world: int):
        if len(uid) != ID_BYTES:
            raise ValueError("a communicator id is 128 bytes")
        self.ctx, self.rank, self.world = ctx, rank, world
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        ctx.check(N.lib.lurkhip_comm_create(ctx.handle, C.addressof(buf), rank, world, C.byref(h)))
        self.handle = h

    @classmethod
    def from_process_group(cls, ctx: Context) -> "Comm":
        """One communicator over the ranks of the initialised torch.distributed process group (rank 0 draws the id)."""
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        box = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(ctx, box[0], rank, world)

    def close(self):
        if getattr(self, "handle", None):
            N.lib.lurkhip_comm_destroy(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exchange_roots(self, shard_indices, roots, n_shards=None):
        """This rank's (index, root[8]) pairs -> the roots of ALL shards in shard order (lists of 8 ints).  With `n_shards` (the
        number of shards of the execution, known to every rank) the ranks may hold different numbers of shards, or none
        (lurkhip_exchange_roots_var); without it every rank passes the same number."""
        idx = as_u32(np.asarray(shard_indices).reshape(-1))
        r = as_u32(np.asarray(roots).reshape(-1, 8))
        if len(idx) != len(r):
            raise ValueError("one index per root")
        if n_shards is None:
            out = np.zeros((len(idx) * self.world, 8), dtype=np.uint32)
            self.ctx.check(N.lib.lurkhip_exchange_roots(self.ctx.handle, self.handle, idx.ctypes.data, r.ctypes.data, len(idx), out.ctypes.data))
        else:
            out = np.zeros((max(int(n_shards), 1), 8), dtype=np.uint32)
            self.ctx.check(N.lib.lurkhip_exchange_roots_var(self.ctx.handle, self.handle, idx.ctypes.data if len(idx) else None,
                                                            r.ctypes.data if len(idx) else None, len(idx), int(n_shards), out.ctypes.data))
            out = out[: int(n_shards)]
        return [[int(x) for x in row] for row in out]

    def reduce_sums(self, local_sums):
        """Canonical extension-field elements (4 lanes each) of this rank -> the machine-wide sum, on every rank."""
        s = as_u32(np.asarray(list(local_sums), dtype=np.int64).reshape(-1, 4))
        out = np.zeros(4, dtype=np.uint32)
        self.ctx.check(N.lib.lurkhip_reduce_sums(self.ctx.handle, self.handle, s.ctypes.data, len(s), out.ctypes.data))
        return tuple(int(x) for x in out)

    def exchange_roots_dev(self, records_dev, n_local: int, gathered_dev):
        """Device-resident records ([n_local][9] words) -> [world * n_local][9] words in rank order; enqueued, not waited for."""
        from .context import _addr

        self.ctx.check(N.lib.lurkhip_exchange_roots_dev(self.ctx.handle, self.handle, _addr(records_dev), n_local, _addr(gathered_dev)))

    def reduce_sums_dev(self, lanes_dev, total_dev):
        from .context import _addr

        self.ctx.check(N.lib.lurkhip_reduce_sums_dev(self.ctx.handle, self.handle, _addr(lanes_dev), _addr(total_dev)))


def bring_up(ctx: Context, device="cuda"):
    """(Comm, None) when the C-ABI route works on EVERY rank of the torch process group, else (None, why): the decision is taken
    collectively, so that all ranks fall back to torch.distributed together -- a rank that cannot load librccl must not leave the
    others inside ncclCommInitRank, and a communicator that comes up wrong must not be found out inside the timed region.
    Three agreed steps: the loader binds a library; the communicator is created; one exchange of roots and one reduction of sums
    give the expected words."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()

    def agreed(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    why = None
    try:
        library()
    except Exception as e:  # noqa: BLE001
        why = f"loader: {e}" if str(e) else "loader: librccl could not be loaded (LURKHIP_RCCL_LIB, the copy the process has mapped, the system's: none bound)"
    if not agreed(why is None):
        return None, why or "another rank could not load librccl"
    c = None
    try:
        c = Comm.from_process_group(ctx)
    except Exception as e:  # noqa: BLE001
        why = f"lurkhip_comm_create: {e}"
    if not agreed(c is not None):
        if c is not None:
            c.close()
        return None, why or "another rank could not create its communicator"
    try:
        P = 2013265921
        roots = c.exchange_roots([rank], [[rank + 1] * 8], n_shards=world)
        total = c.reduce_sums([(rank + 1, 0, 0, 7)])
        good = roots == [[r + 1] * 8 for r in range(world)] and total == ((world * (world + 1) // 2) % P, 0, 0, (7 * world) % P)
        if not good:
            why = f"self-test: roots {roots[:2]}..., sums {total}"
    except Exception as e:  # noqa: BLE001
        good, why = False, f"self-test: {e}"
    if not agreed(good):
        c.close()
        return None, why or "another rank failed the communicator's self-test"
    return c, None
