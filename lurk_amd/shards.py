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
    """Shards to ranks by estimated work (longest processing time first; rank sizes differ by at most one shard):
    `Shard::shard` cuts every chip at the same row count (/root/reference/src/lair/execute.rs:186-216: ceil(rows / max_shard_size)
    shards -- any number, not a multiple of the rank count), so the first shards of an execution hold all its chips and the last
    ones only the tallest -- round robin would leave rank 0 with twice the average.  Returns [shard indices of rank 0, of rank
    1, ...], each ascending (a rank's list is empty when there are fewer shards than ranks); identical on every rank (ties
    broken by index)."""
    n = len(costs)
    most = -(-n // world)  # no rank holds more than ceil(n / world) shards: the device memory of a rank bounds what it can keep resident
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for s in sorted(range(n), key=lambda i: (-float(costs[i]), i)):
        r = min((r for r in range(world) if len(out[r]) < most), key=lambda r: (load[r], r))
        out[r].append(s)
        load[r] += float(costs[s])
    return [sorted(x) for x in out]


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def exchange_roots(local_roots, device="cpu", shard_indices=None, comm=None, n_shards=None, group=None):
    """local_roots: [k][8] roots of this rank's shards.  With `shard_indices` (this rank's shard numbers, same order as
    local_roots; any assignment, e.g. assign_shards_balanced) k may differ from rank to rank and may be 0: the ranks first
    all-gather their counts, then records padded to the largest count (a 9-word (index, root) record per shard, index -1 = padding).
    Without indices every rank passes the same k (the round-robin layout of `assign_shards`, undone on return).
    `comm` (lurk_amd.comm.Comm, with shard_indices): the all-gathers run behind the C ABI on RCCL (lurkhip_exchange_roots_var)
    instead of torch.distributed -- the route a Rust host takes; the gloo CPU tests and single-process runs keep the torch path.
    Returns the roots of all shards ordered by shard index.  `n_shards` (optional): the number of shards every rank expects.
    `group`: the torch process group to use (two machine proofs in flight on one rank issue their collectives from two threads:
    each thread has its own group, so that the order of collectives is the same on every rank within each group)."""
    if comm is not None:
        if shard_indices is None:
            raise ValueError("the C-ABI exchange carries the shard indices with the roots")
        return comm.exchange_roots(shard_indices, local_roots, n_shards=n_shards)
    import torch

    local = torch.tensor(np.asarray(local_roots, dtype=np.int64).reshape(-1, 8), device=device)
    dist = _dist()
    if shard_indices is not None:
        idx = torch.tensor(np.asarray(shard_indices, dtype=np.int64).reshape(-1, 1), device=device)
        if idx.shape[0] != local.shape[0]:
            raise ValueError("one shard index per root")
        rec = torch.cat([idx, local], dim=1).contiguous()  # [k][9]
        if dist is None:
            rows = rec.cpu().tolist()
        else:
            world = dist.get_world_size()
            counts = torch.zeros(world, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(counts, torch.tensor([rec.shape[0]], dtype=torch.int64, device=device), group=group)
            most = int(counts.max())
            padded = torch.full((max(most, 1), 9), -1, dtype=torch.int64, device=device)
            padded[: rec.shape[0]] = rec
            out = torch.zeros((world,) + tuple(padded.shape), dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(out.view(-1), padded.view(-1), group=group)
            rows = [r for per_rank in out.cpu().tolist() for r in per_rank if r[0] >= 0]
        rows.sort(key=lambda r: r[0])
        if [r[0] for r in rows] != list(range(len(rows))) or (n_shards is not None and len(rows) != n_shards):
            raise ValueError("shard indices of the ranks are not a partition of 0 .. n-1")
        return [[int(x) for x in r[1:]] for r in rows]
    if dist is None:
        return [[int(x) for x in r] for r in local.cpu().tolist()]
    world = dist.get_world_size()
    out = torch.zeros((world,) + tuple(local.shape), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out.view(-1), local.view(-1), group=group)
    gathered = out.cpu().tolist()  # [rank][k][8]
    k = local.shape[0]
    return [[int(x) for x in gathered[s % world][s // world]] for s in range(k * world)]


def reduce_cumulative_sums(local_sums, device="cpu", comm=None, group=None):
    """local_sums: iterable of extension-field elements (4 canonical lanes each): the cumulative sums of every chip of
    every shard this rank proved.  Returns the machine-wide total (4 lanes, reduced mod p) on every rank; the proof
    set is consistent iff it is zero.  `comm`: behind the C ABI on RCCL (lurkhip_reduce_sums)."""
    if comm is not None:
        return comm.reduce_sums(local_sums)
    import torch

    acc = np.zeros(4, dtype=np.int64)
    for s in local_sums:
        acc += np.asarray(s, dtype=np.int64)
        acc %= P
    t = torch.from_numpy(acc).to(device)
    dist = _dist()
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tuple(int(x) % P for x in t.cpu().tolist())


def proof_cumulative_sums(words) -> list:
    """The chips' cumulative sums (4 canonical lanes each) straight from the flat proof words of lurkhip_shard_prove
    (header 10 words, then 11 words per chip: 7 of metadata + the sum)."""
    n_chips = int(words[1])
    return [words[10 + 11 * i + 7:10 + 11 * i + 11] for i in range(n_chips)]


class RankStep:
    """One rank's share of ONE machine proof over several ranks -- the body of bench.py's timed step and of the multi-process
    tests, so that what is timed is what is verified.

    phase 1 (`LocalProver::commit_shards` [UPSTREAM-RECALL]): trace generation + main commitment of this rank's shards;
    exchange: all-gather of (shard index, main root) records -- every shard's transcript observes the verifying key, pc_start,
              then every shard's root and the public values in SHARD order, whatever the assignment -- (`exchange_roots`);
    phase 2: this rank's shards proved with clones of that transcript, two in flight when the rank has several (`prove_lanes`);
    check:   the extension-field cumulative sums all-reduced as 4 x int64 (`reduce_cumulative_sums`): each rank's own sum is
             non-zero, the total must vanish (/root/reference/src/lair/execute.rs:186-241 shards, lair_chip.rs:104-139).
    `device` is where the two tiny collectives' tensors live: "cuda" over RCCL, "cpu" over gloo or without a process group;
    with `comm` both collectives run inside the library (lurkhip_exchange_roots / lurkhip_reduce_sums on the context's stream)."""

    def __init__(self, machine, vk_root, public_values, prepared_all, shard_indices, num_queries, pow_bits, device="cpu", lane_ctx=None, comm=None,
                 n_shards=None, group=None):
        self.group = group        # torch process group of the two collectives (None: the default group)
        self.n_shards = n_shards  # shards of the whole execution: with it the ranks may hold different numbers of shards (or none)
        self.machine, self.vk_root, self.pv = machine, vk_root, list(public_values)
        self.prepared_all, self.mine = prepared_all, list(shard_indices)
        self.num_queries, self.pow_bits, self.device, self.lane_ctx = num_queries, pow_bits, device, lane_ctx
        self.comm = comm        # lurk_amd.comm.Comm: the collectives behind the C ABI (RCCL); None: torch.distributed / single process
        self.host_ms = {}       # host milliseconds spent in the two collectives, summed over calls
        self.calls = 0
        self.rank_sums = []     # this rank's own sum, per call
        self.grand_sums = []    # the all-reduced total, per call
        self.roots = None       # the gathered roots of the last call, in shard order
        self.last_proofs = None  # this rank's proof words of the last call

    # ---- the four parts of a step; only `exchange` and `check` are collectives
    def commit(self):
        """Phase 1 on this rank: traces + main commitments of its shards (GPU work on the machine's context, no collective).
        Returns the state `prove` continues from."""
        from . import prover

        m, ctx = self.machine, self.machine.ctx
        handles, roots, ch = [], [], None
        for pr in self.prepared_all:
            ctx.span_begin("trace_all")
            traces = m.run_prepared(pr)
            ctx.span_end("trace_all")
            if ch is None:  # the transcript is opened on the host while the trace kernels run
                ch = prover.Challenger(ctx)
                ch.observe(self.vk_root)
                ch.observe([0])
            handle, root = m.commit_shard(traces)
            handles.append(handle)
            roots.append(root)
        return {"handles": handles, "roots": roots, "ch": ch}

    def exchange(self, state):
        """Collective: every shard's main root to every rank, in shard order."""
        import time

        t = time.perf_counter()
        self.roots = exchange_roots(state["roots"], device=self.device, shard_indices=self.mine, comm=self.comm, n_shards=self.n_shards, group=self.group)
        self.host_ms["exchange_roots"] = self.host_ms.get("exchange_roots", 0.0) + (time.perf_counter() - t) * 1e3
        return self.roots

    def prove(self, state, gathered):
        """Phase 2 on this rank (GPU work on the machine's context and its second lane, no collective)."""
        from . import prover

        m, ch = self.machine, state["ch"]
        for r in gathered:
            ch.observe(r)
            ch.observe(self.pv)
        proofs = prover.prove_lanes(m, state["handles"], ch, self.pv, self.num_queries, self.pow_bits, parse=False, lane_ctx=self.lane_ctx)
        for handle in state["handles"]:
            m.free_shard(handle)
        return proofs

    def check(self, proofs):
        """Collective: the cumulative sums of all ranks' proofs, all-reduced; records this rank's own sum and the total."""
        import time

        cs = [c for words in proofs for c in proof_cumulative_sums(words)]
        mine_sum = np.zeros(4, dtype=np.int64)
        for c in cs:
            mine_sum = (mine_sum + np.asarray(c, dtype=np.int64)) % P
        self.rank_sums.append(tuple(int(x) for x in mine_sum))
        t = time.perf_counter()
        self.grand_sums.append(reduce_cumulative_sums(cs, device=self.device, comm=self.comm, group=self.group))
        self.host_ms["reduce_sums"] = self.host_ms.get("reduce_sums", 0.0) + (time.perf_counter() - t) * 1e3
        self.calls += 1
        self.last_proofs = proofs

    def __call__(self, parse=False):
        from . import prover

        state = self.commit()
        gathered = self.exchange(state)
        proofs = self.prove(state, gathered)
        self.check(proofs)
        return [prover.parse_proof(w) for w in proofs] if parse else proofs


def run_pipelined(steps, n_steps: int, on_proofs=None):
    """n_steps machine proofs on this rank with phase 1 of proof j + 1 under phase 2 of proof j (round 3).

    `steps` = two RankStep objects over two Machines of the same toplevel (own contexts, own prepared inputs): proof j runs on
    steps[j % 2].  While proof j is in phase 2 -- latency chains: tree tails, FRI layers, transcript round trips, and with the
    reference's sharding a light second shard whose lane idles early -- the throughput-bound traces and main commitments of proof
    j + 1 run on the other machine's context.  Every collective is issued by the calling thread, in the same order on every rank
    (exchange j, check j, exchange j + 1, ...): the worker thread only proves.  `on_proofs(j, proofs)` sees each proof's words."""
    from concurrent.futures import ThreadPoolExecutor

    assert len(steps) == 2 and n_steps >= 1
    pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="lurkhip-phase2")
    try:
        pending = steps[0].commit()
        for j in range(n_steps):
            cur = steps[j % 2]
            gathered = cur.exchange(pending)
            fut = pool.submit(cur.prove, pending, gathered)
            try:
                nxt = steps[(j + 1) % 2].commit() if j + 1 < n_steps else None
            finally:
                proofs = fut.result()
            cur.check(proofs)
            if on_proofs is not None:
                on_proofs(j, proofs)
            pending = nxt
    finally:
        pool.shutdown(wait=True)


def run_in_flight(steps, n_steps: int, on_proofs=None, stagger_s: float = 0.0):
    """n_steps machine proofs on this rank, K = len(steps) of them IN FLIGHT (round 5): thread t runs proofs t, t + K, t + 2K, ...
    from start to end (commit -> exchange -> prove -> check) on steps[t] -- its own Machine and contexts and, what makes this legal,
    its own communicator (RCCL `Comm` or torch process group): within one communicator the collectives are issued by one thread, in
    proof order, the same on every rank; across communicators nothing is ordered and nothing needs to be.  While one proof waits in
    a collective or in a latency chain, the other's kernels have the device: a rank never idles at a collective (the
    one-proof-at-a-time rank schedule lost 7 % per GPU to exactly that).  Thread t starts t * stagger_s
    late so that the proofs run out of phase.  `on_proofs(j, proofs)` may be called from any of the threads."""
    import threading
    import time

    k = len(steps)
    assert k >= 1 and n_steps >= 1
    errors = []

    def lane(t):
        try:
            if t and stagger_s:
                time.sleep(t * stagger_s)
            for j in range(t, n_steps, k):
                proofs = steps[t]()
                if on_proofs is not None:
                    on_proofs(j, proofs)
        except BaseException as e:  # surfaced on the calling thread
            errors.append(e)

    threads = [threading.Thread(target=lane, args=(t,), name=f"lurkhip-proof-lane{t}") for t in range(1, k)]
    for th in threads:
        th.start()
    lane(0)
    for th in threads:
        th.join()
    if errors:
        raise errors[0]


def run_committed_ahead(steps, n_steps: int, on_proofs=None):
    """n_steps machine proofs on this rank with phase 1 running AHEAD on its own thread (round 3, second attempt at filling a
    rank's device: `run_pipelined` joins the two phases once per proof, so phase 2 runs alone whenever phase 1 of the next proof
    is done first).

    `steps` = K >= 2 RankStep objects over K Machines of the same toplevel (own contexts, own prepared inputs); proof j runs on
    steps[j % K].  A committer thread produces phase 1 (traces + main commitments, no collective) of proofs 0, 1, 2, ... as fast
    as machines come free -- at most K - 1 ahead of the proof being finished --, the calling thread does exchange -> phase 2 ->
    check for each proof in order: every collective is issued by the calling thread, in the same order on every rank."""
    import queue
    import threading

    k = len(steps)
    assert k >= 2 and n_steps >= 1
    committed: "queue.Queue" = queue.Queue()
    free = threading.Semaphore(k)   # machines whose previous proof is finished

    def committer():
        try:
            for j in range(n_steps):
                free.acquire()
                committed.put(steps[j % k].commit())
        except BaseException as e:  # surfaced on the calling thread
            committed.put(e)

    th = threading.Thread(target=committer, name="lurkhip-phase1")
    th.start()
    try:
        for j in range(n_steps):
            state = committed.get()
            if isinstance(state, BaseException):
                raise state
            cur = steps[j % k]
            gathered = cur.exchange(state)
            proofs = cur.prove(state, gathered)
            cur.check(proofs)
            free.release()
            if on_proofs is not None:
                on_proofs(j, proofs)
    finally:
        for _ in range(n_steps):  # a failure on this side: let the committer run out
            free.release()
        th.join()


def scatter_prepared(machine, all_shards, assignment, device="cpu", src: int = 0):
    """ONE rank executed the program (`src`: it holds the QueryRecord, hence `all_shards`); every rank proves the shards it is dealt.
    `src` prepares the other ranks' shards one at a time (Machine.prepare_shard: the chips' kernel inputs on its device), exports
    them (Machine.export_prepared: bytes) and sends them to their owner; returns {shard index: entries} for this rank's shards on
    every other rank ({} on `src`, which prepares its own shards from the record as before).  Every rank calls this with the same
    `assignment`; the ranks other than `src` pass all_shards = None.  Broadcasts on `device` ("cuda": RCCL, "cpu": gloo) that only
    the owner keeps, a small object broadcast per shard for the blob sizes; a rank other than `src` never sees the QueryRecord and
    holds one blob at a time beyond its own shards -- its host seconds and resident set do not grow with the number of ranks."""
    import numpy as np
    import torch

    d = _dist()
    rank, world = d.get_rank(), d.get_world_size()
    mine = {}
    for r in range(world):
        if r == src:
            continue
        for i in assignment[r]:
            entries = None
            head = [None]
            if rank == src:
                prep = machine.prepare_shard(all_shards[i])
                entries = machine.export_prepared(prep)
                for *_, p in prep:
                    if p is not None:
                        p.close()
                head = [[(mi, None if b is None else int(b.nbytes)) for mi, b in entries]]
            d.broadcast_object_list(head, src=src)
            # the blobs go out as broadcasts that only the owner keeps: the one collective family the timed region uses as well
            # (first contact with a real multi-GPU box should not meet a second kind of communicator before the first proof)
            got = []
            for k, (mi, nbytes) in enumerate(head[0]):
                if nbytes is None:
                    got.append((mi, None))
                    continue
                if rank == src:
                    t = torch.from_numpy(entries[k][1])
                    buf = t.to(device) if device != "cpu" else t
                else:
                    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
                d.broadcast(buf, src=src)
                if rank == r:
                    got.append((mi, np.ascontiguousarray(buf.cpu().numpy())))
                del buf
            if rank == r:
                mine[i] = got
    return mine


def gather_proofs(words_list, shard_indices, dst: int = 0):
    """Collects the flat proof words of every rank's shards on rank `dst`, ordered by shard index (None elsewhere): the set a
    verifier receives.  Not part of a timed step -- proofs are megabytes; `gather_object` over the process group's default
    backend (object collectives go through host memory on either backend)."""
    dist = _dist()
    rec = [(int(i), np.ascontiguousarray(w, dtype=np.uint32)) for i, w in zip(shard_indices, words_list)]
    if dist is None:
        got = [rec]
    else:
        got = [None] * dist.get_world_size() if dist.get_rank() == dst else None
        dist.gather_object(rec, got, dst=dst)
        if got is None:
            return None
    flat = sorted((r for per_rank in got for r in per_rank), key=lambda r: r[0])
    if [i for i, _ in flat] != list(range(len(flat))):
        raise ValueError("gathered shard indices are not a partition of 0 .. n-1")
    return [w for _, w in flat]
