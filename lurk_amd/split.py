"""One shard proved by G = 2, 4, 8 .. ranks together (include/lurkhip.h: lurkhip_split_comm, lurkhip_*_split; csrc/split.hip).

The reference proves a shard inside one process, and `Shard::shard` only cuts an execution above 2^22 rows
(/root/reference/src/lair/execute.rs:186-241): behind `machine.prove` (/root/reference/benches/fib.rs:124) anything smaller is
one shard.  `SplitProver` is `prover._ShardProver` with the commitments and the proof made by all ranks of a communicator; every
rank gets the words `Machine.prove_shard` returns on one GPU.

Two carriers for the four collectives: `RcclSplitComm` (the library's own RCCL communicator: device buffers stay on the device)
and `TorchSplitComm` (ctypes callbacks that stage through host memory and use a torch.distributed group -- gloo in the tests,
where several ranks share the one GPU of the box, which RCCL refuses)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context, _addr, as_u32

_A2A = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
_AGD = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
_AGH = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_ARH = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64)


class SplitCommStruct(C.Structure):
    """struct lurkhip_split_comm"""

    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("user", C.c_void_p), ("alltoallv_dev", _A2A), ("allgather_dev", _AGD),
                ("allgather_host", _AGH), ("allreduce_sum_u64_host", _ARH)]


class RcclSplitComm:
    """The collectives on a `comm.Comm` (RCCL behind the C ABI: ncclSend / ncclRecv pairs for the all-to-all)."""

    def __init__(self, ctx: Context, comm):
        self.ctx, self.comm = ctx, comm
        self.struct = SplitCommStruct()
        ctx.check(N.lib.lurkhip_comm_split_vtable(ctx.handle, comm.handle, C.byref(self.struct)))
        self.rank, self.world = int(self.struct.rank), int(self.struct.world)
        self.bytes_sent = 0  # (not counted on this route)

    def selftest(self):
        return carrier_selftest(self)


def carrier_selftest(self) -> str | None:
    """The four collectives of a carrier (`RcclSplitComm`, `TorchSplitComm`) once, with known words (ragged blocks: rank r sends
    100 + 7 d + r words to rank d): None when every word is where it belongs on THIS rank, else what was wrong.  Collective; the caller agrees on the outcome over its process group
    (bench.py: every rank falls back to the host route together)."""
    import torch

    w, r, st = self.world, self.rank, self.struct
    n_to = [100 + 7 * d + r for d in range(w)]
    n_from = [100 + 7 * r + s for s in range(w)]
    soff = (C.c_uint64 * (w + 1))(*np.concatenate([[0], np.cumsum(n_to)]).tolist())
    roff = (C.c_uint64 * (w + 1))(*np.concatenate([[0], np.cumsum(n_from)]).tolist())
    send = torch.cat([torch.arange(n, dtype=torch.int32) + (r * 1000 + d) * 10000 for d, n in enumerate(n_to)]).to(f"cuda:{self.ctx.device}")
    recv = torch.zeros(sum(n_from), dtype=torch.int32, device=send.device)
    torch.cuda.synchronize()
    try:
        if st.alltoallv_dev(st.user, send.data_ptr(), soff, recv.data_ptr(), roff, None) != 0:
            return "all-to-all: " + N.last_error(self.ctx.handle)
        torch.cuda.synchronize()
        want = torch.cat([torch.arange(n, dtype=torch.int32) + (s * 1000 + r) * 10000 for s, n in enumerate(n_from)])
        if not torch.equal(recv.cpu(), want):
            return "all-to-all: a block arrived with other words"
        mine = torch.full((5,), 17 + r, dtype=torch.int32, device=send.device)
        got = torch.zeros(5 * w, dtype=torch.int32, device=send.device)
        if st.allgather_dev(st.user, mine.data_ptr(), got.data_ptr(), 5, None) != 0:
            return "device all-gather: " + N.last_error(self.ctx.handle)
        torch.cuda.synchronize()
        if got.cpu().tolist() != [17 + s for s in range(w) for _ in range(5)]:
            return "device all-gather: wrong words"
        src, dst = bytes([r + 1] * 6), C.create_string_buffer(6 * w)
        if st.allgather_host(st.user, src, dst, 6) != 0 or dst.raw != b"".join(bytes([s + 1] * 6) for s in range(w)):
            return "host all-gather failed"
        lanes = (C.c_uint64 * 2)(r + 1, (1 << 40) + r)
        if st.allreduce_sum_u64_host(st.user, lanes, 2) != 0 or list(lanes) != [w * (w + 1) // 2, w * (1 << 40) + w * (w - 1) // 2]:
            return "64-bit all-reduce failed"
    except Exception as e:  # noqa: BLE001
        return repr(e)
    return None


class TorchSplitComm:
    """The collectives over a torch.distributed group through host memory (device blocks are read back on the context's stream,
    exchanged, written again): for the oversubscribed tests and as the fall-back carrier.  Counts the bytes this rank sends."""

    def __init__(self, ctx: Context, group=None):
        import torch.distributed as dist

        self.ctx, self.group = ctx, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.bytes_sent = 0
        self.alltoall_bytes = 0
        self.calls = {"alltoallv": 0, "allgather_dev": 0, "allgather_host": 0, "allreduce": 0}
        self.error = None
        self._cb = (_A2A(self._alltoallv_dev), _AGD(self._allgather_dev), _AGH(self._allgather_host), _ARH(self._allreduce))  # kept alive
        self.struct = SplitCommStruct(self.rank, self.world, None, *self._cb)
        self.turns = False
        self.segments = []

    # ---- taking turns (bench.py --split-turns): what a rank's work BETWEEN two collectives takes when it has the device to itself.
    # Several ranks on one device run their kernels against each other and their clocks say nothing; here a token goes round instead:
    # after a collective rank 0 works alone until it arrives at the next one, then rank 1 does, ... and the collective runs when the
    # last rank has arrived.  segments[k] = this rank's seconds (host and device, the stream drained) between collectives k - 1 and k.
    def _token(self, send: bool):
        import torch
        import torch.distributed as dist

        t = torch.zeros(1, dtype=torch.int32)
        peer = self.rank + 1 if send else self.rank - 1
        peer = dist.get_global_rank(self.group, peer) if self.group is not None else peer
        (dist.send if send else dist.recv)(t, peer, group=self.group)

    def _segment_end(self):
        import time

        self.ctx.sync()
        self.segments.append(time.perf_counter() - self._t_seg)
        if self.rank + 1 < self.world:
            self._token(True)

    def _segment_begin(self):
        import time

        if self.rank > 0:
            self._token(False)
        self._t_seg = time.perf_counter()

    def begin_turns(self):
        import torch.distributed as dist

        self.ctx.sync()
        dist.barrier(group=self.group)
        self.turns, self.segments = True, []
        self._segment_begin()

    def end_turns(self):
        """-> this rank's segment times of the proof(s) since begin_turns (the last one: from the last collective to here)"""
        import torch.distributed as dist

        self._segment_end()
        self.turns = False
        dist.barrier(group=self.group)
        return list(self.segments)

    def selftest(self):
        return carrier_selftest(self)

    def _guard(self, fn, *a):
        try:
            if self.turns:
                self._segment_end()
            fn(*a)
            if self.turns:
                self._segment_begin()
            return 0
        except BaseException as e:  # nothing unwinds into C; the library reports "collective failed", the cause is kept here
            self.error = e
            return -1

    def _d2h(self, dev, words):
        out = np.empty(words, dtype=np.uint32)
        if words:
            self.ctx.check(N.lib.lurkhip_memcpy_d2h(self.ctx.handle, out.ctypes.data, dev, words * 4))
        return out

    def _h2d(self, dev, arr):
        if arr.size:
            self.ctx.check(N.lib.lurkhip_memcpy_h2d(self.ctx.handle, dev, arr.ctypes.data, arr.size * 4))

    def _alltoallv_dev(self, user, send, soff, recv, roff, stream):
        def run():
            import torch
            import torch.distributed as dist

            w = self.world
            so, ro = [int(soff[i]) for i in range(w + 1)], [int(roff[i]) for i in range(w + 1)]
            s = self._d2h(send, so[w]).view(np.int32)
            ins = [torch.from_numpy(s[so[d]:so[d + 1]].copy()) for d in range(w)]
            outs = [torch.empty(ro[r + 1] - ro[r], dtype=torch.int32) for r in range(w)]
            # (gloo has no all_to_all for ragged lists on every version: w broadcasts-free rounds of send / recv pairs)
            reqs = []
            for k in range(1, w):
                to, frm = (self.rank + k) % w, (self.rank - k) % w
                reqs.append(dist.isend(ins[to], dist.get_global_rank(self.group, to) if self.group is not None else to, group=self.group))
                reqs.append(dist.irecv(outs[frm], dist.get_global_rank(self.group, frm) if self.group is not None else frm, group=self.group))
            outs[self.rank].copy_(ins[self.rank])
            for r in reqs:
                r.wait()
            self.calls["alltoallv"] += 1
            sent = 4 * (so[w] - (so[self.rank + 1] - so[self.rank]))
            self.bytes_sent += sent
            self.alltoall_bytes += sent
            self._h2d(recv, np.concatenate([o.numpy() for o in outs]).view(np.uint32) if ro[w] else np.zeros(0, dtype=np.uint32))

        return self._guard(run)

    def _allgather_dev(self, user, send, recv, words, stream):
        def run():
            import torch
            import torch.distributed as dist

            s = torch.from_numpy(self._d2h(send, int(words)).view(np.int32))
            outs = [torch.empty(int(words), dtype=torch.int32) for _ in range(self.world)]
            dist.all_gather(outs, s, group=self.group)
            self.calls["allgather_dev"] += 1
            self.bytes_sent += 4 * int(words) * (self.world - 1)
            self._h2d(recv, torch.cat(outs).numpy().view(np.uint32))

        return self._guard(run)

    def _allgather_host(self, user, send, recv, nbytes):
        def run():
            import torch
            import torch.distributed as dist

            n = int(nbytes)
            s = torch.frombuffer(bytearray(C.string_at(send, n)), dtype=torch.uint8) if n else torch.empty(0, dtype=torch.uint8)
            outs = [torch.empty(n, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(outs, s, group=self.group)
            self.calls["allgather_host"] += 1
            self.bytes_sent += n * (self.world - 1)
            data = torch.cat(outs).numpy().tobytes()
            C.memmove(recv, data, len(data))

        return self._guard(run)

    def _allreduce(self, user, buf, n):
        def run():
            import torch
            import torch.distributed as dist

            k = int(n)
            a = np.ctypeslib.as_array(buf, shape=(k,))
            t = torch.from_numpy(a.astype(np.int64))  # (addends below 2^32, at most 64 ranks: no overflow)
            dist.all_reduce(t, group=self.group)
            self.calls["allreduce"] += 1
            self.bytes_sent += 8 * k
            a[:] = t.numpy().astype(np.uint64)

        return self._guard(run)


class SplitProver:
    """setup / commit_shard / prove_shard of `prover.Machine` (or `StarkMachine`) by all ranks of `comm` together.  Every rank
    passes the same traces; chips of at least 2^min_log_n rows are cut across the ranks, the shorter ones proved whole by every rank."""

    def __init__(self, machine, comm, min_log_n: int = 10):
        self.m, self.comm, self.min_log_n = machine, comm, int(min_log_n)
        self.ctx = machine.ctx
        self.pk = None
        self.vk_root = None
        self._included = {}

    def _check(self, status):
        if status != N.OK and getattr(self.comm, "error", None) is not None:
            e, self.comm.error = self.comm.error, None
            raise RuntimeError(f"a collective of the split prover failed: {e!r}") from e
        self.ctx.check(status)

    def setup(self):
        """machine.setup over the ranks: the preprocessed traces' commitment (the byte chip's 2^16 x 6 table for a Lair machine)."""
        m = self.m
        if m.pk is None:
            m.setup()  # (builds the preprocessed traces on the device; its one-rank key is what the split root is compared with)
        prep = getattr(m, "_prep", None)
        h = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        if prep is not None:
            ptrs = (C.c_void_p * 1)(prep.data_ptr())
            lh = np.array([prep.shape[0].bit_length() - 1], dtype=np.uint32)
            ws = np.array([prep.shape[1]], dtype=np.uint32)
            self._check(N.lib.lurkhip_setup_split(self.ctx.handle, C.byref(self.comm.struct), self.min_log_n, 1, C.cast(ptrs, C.c_void_p), _addr(lh), _addr(ws), 1,
                                                  C.byref(h), _addr(root)))
        else:
            self._check(N.lib.lurkhip_setup_split(self.ctx.handle, C.byref(self.comm.struct), self.min_log_n, 0, None, None, None, 1, C.byref(h), _addr(root)))
        self.pk = h
        self.vk_root = [int(x) for x in root]
        return self.vk_root

    def block_buffers(self, prepared):
        """Device tensors for `run_prepared_blocks`: a cut chip's holds this rank's block of rows, the others their whole trace."""
        import torch

        g = self.comm.world
        out = []
        for _, air, lg, t, p in prepared:
            if p is None or lg < self.min_log_n:
                out.append(t if p is None else torch.empty((1 << lg, p.width), dtype=torch.int32, device=f"cuda:{self.ctx.device}"))
            else:
                out.append(torch.empty(((1 << lg) // g, p.width), dtype=torch.int32, device=f"cuda:{self.ctx.device}"))
        return out

    def run_prepared_blocks(self, prepared, buffers=None):
        """`Machine.run_prepared` for one shard over the ranks: every chip of at least 2^min_log_n rows generates only this rank's
        block of rows [rank N / G, (rank + 1) N / G) (trace generation is row by row: the same kernels on offset inputs), the
        shorter chips their whole trace.  Returns the list `commit_shard(.., row_blocks=True)` takes."""
        buffers = buffers if buffers is not None else self.block_buffers(prepared)
        g, r = self.comm.world, self.comm.rank
        out = []
        for (mi, air, lg, _, p), t in zip(prepared, buffers):
            if p is not None:
                if lg >= self.min_log_n:
                    rows = (1 << lg) // g
                    p.run_rows(r * rows, rows, t, repr=N.REPR_MONTY, ctx=self.ctx)
                else:
                    p.run(t, repr=N.REPR_MONTY, ctx=self.ctx)
            out.append((mi, air, lg, t))
        return out

    def commit_shard(self, traces, row_blocks: bool = False):
        """traces: [(machine index, air, log_height, device trace)] as `Machine.run_prepared` returns them -- whole, on every rank --,
        or, with row_blocks, as `run_prepared_blocks` does."""
        n = len(traces)
        airs = (C.c_void_p * n)(*[a.handle.value for _, a, _, _ in traces])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, _, _, t in traces])
        lh = np.array([lg for _, _, lg, _ in traces], dtype=np.uint32)
        pitches = np.array([t.stride(0) for _, _, _, t in traces], dtype=np.uint32)
        prep_idx = np.array([self.m._prep_index(mi) for mi, _, _, _ in traces], dtype=np.int32)
        h = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        self._check(N.lib.lurkhip_shard_commit_split(self.ctx.handle, C.byref(self.comm.struct), self.min_log_n, n, C.cast(airs, C.c_void_p), _addr(lh),
                                                     C.cast(ptrs, C.c_void_p), _addr(pitches), _addr(prep_idx), 1, 1 if row_blocks else 0, C.byref(h), _addr(root)))
        self._included[h.value] = [mi for mi, _, _, _ in traces]
        return h, [int(x) for x in root]

    def prove_shard(self, shard_handle, challenger, public_values, num_queries=100, pow_bits=16, parse=True):
        from .prover import parse_proof

        pv = as_u32(public_values)
        p = C.c_void_p()
        self._check(N.lib.lurkhip_shard_prove_split(self.ctx.handle, self.pk, shard_handle, challenger.handle, _addr(pv), len(pv), num_queries, pow_bits, C.byref(p)))
        n = int(N.lib.lurkhip_proof_words(p))
        words = np.zeros(n, dtype=np.uint32)
        N.check(N.lib.lurkhip_proof_read(p, _addr(words), n))
        N.lib.lurkhip_proof_free(p)
        included = self._included[shard_handle.value]
        for i in range(int(words[1])):
            words[10 + 11 * i] = included[int(words[10 + 11 * i])]
        return parse_proof(words) if parse else words

    def free_shard(self, shard_handle):
        self._included.pop(shard_handle.value, None)
        N.lib.lurkhip_shard_free(self.ctx.handle, shard_handle)

    def prove(self, traces, public_values, num_queries=100, pow_bits=16, parse=False, row_blocks: bool = False):
        """One shard from its traces to its proof: the transcript `Machine.prove` builds for a one-shard execution."""
        from .prover import Challenger

        if self.pk is None:
            self.setup()
        ch = Challenger(self.ctx)
        ch.observe(self.vk_root)
        ch.observe([0])
        handle, root = self.commit_shard(traces, row_blocks)
        try:
            ch.observe(root)
            ch.observe(public_values)
            return self.prove_shard(handle, ch, public_values, num_queries, pow_bits, parse=parse), root
        finally:
            self.free_shard(handle)

    def close(self):
        if self.pk:
            N.lib.lurkhip_pk_free(self.ctx.handle, self.pk)
            self.pk = None


def parse_plan(words):
    """lurkhip_split_plan's output (64-bit words) as a dict: the groups, this rank's tiles, and the job lists of the two exchanges."""
    w = [int(x) for x in words]
    pos = [0]

    def take(n=1):
        out = w[pos[0]:pos[0] + n]
        pos[0] += n
        return out if n > 1 else out[0]

    def jobs():
        out = []
        for _ in range(take()):
            buf, row0, col0, row_stride, lin_off, lin_pitch, width, rows = take(8)
            out.append(dict(buf=buf, row0=row0, col0=col0, row_stride=row_stride, lin_off=lin_off, lin_pitch=lin_pitch, width=width, rows=rows))
        return out

    def vec():
        n = take()
        return [take() for _ in range(n)]

    world = take()
    plan = {"world": world, "groups": []}
    for _ in range(take()):
        log_n, W, local_pitch, slab_w, n_mats, n_extras, W_local, sparse = take(8)
        mats = [tuple(take(2)) for _ in range(n_mats)]
        bounds = take(world + 1)
        extras = [tuple(take(4)) for _ in range(n_extras)]
        plan["groups"].append(dict(log_n=log_n, W=W, W_local=W_local, sparse=bool(sparse), local_pitch=local_pitch, slab_w=slab_w, mats=mats, bounds=bounds, extras=extras))
    plan["tiles"] = [dict(zip(("group", "mat", "c0", "w", "slab_col"), take(5))) for _ in range(take())]
    plan["my_extras"] = vec()
    plan["has_a"] = bool(take())
    plan["a_send_off"], plan["a_recv_off"] = vec(), vec()
    plan["a_pack"], plan["a_unpack"] = jobs(), jobs()
    plan["b_send_off"], plan["b_recv_off"] = vec(), vec()
    plan["b_pack"], plan["b_unpack"] = jobs(), jobs()
    assert pos[0] == len(w), "trailing words in the plan"
    return plan


def split_plan(world, rank, min_log_n, log_heights, widths, kinds, lqds=None, chunks=None, n_next=None, runs=None):
    n = len(widths)
    lh, ws = np.array(log_heights, dtype=np.uint32), np.array(widths, dtype=np.uint32)
    ks = np.array(kinds, dtype=np.int32)
    lq = np.array(lqds if lqds is not None else [0] * n, dtype=np.uint32)
    ck = np.array(chunks if chunks is not None else [0] * n, dtype=np.uint32)
    nx = np.array(n_next if n_next is not None else [0] * n, dtype=np.uint32)
    rc = np.array([len(r) for r in runs] if runs is not None else [0] * n, dtype=np.uint32)  # per matrix: its live (first column, width) runs
    rr = np.array([x for r in (runs or []) for pair in r for x in pair] + [0], dtype=np.uint32)
    need = N.lib.lurkhip_split_plan(world, rank, min_log_n, n, _addr(lh), _addr(ws), _addr(ks), _addr(lq), _addr(ck), _addr(nx), _addr(rc), _addr(rr), None, 0)
    if need < 0:
        raise N.LurkHipError(int(need), N.last_error(None))
    out = np.zeros(int(need), dtype=np.uint64)
    N.lib.lurkhip_split_plan(world, rank, min_log_n, n, _addr(lh), _addr(ws), _addr(ks), _addr(lq), _addr(ck), _addr(nx), _addr(rc), _addr(rr), out.ctypes.data, int(need))
    return parse_plan(out)
