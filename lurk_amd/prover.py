"""Host-side mirror of the reference's machine / prover objects over the lurkhip C ABI.

`Machine` plays `StarkMachine<BabyBearPoseidon2, LairChip>` as the reference builds it
(`new_machine`, /root/reference/src/core/stark_machine.rs:16-30; chip vector = Entrypoint + Func chips + 6 Mem chips
+ Bytes, /root/reference/src/lair/lair_chip.rs:196-211), `setup` / `prove` are `machine.setup` / `machine.prove::<P>`
(/root/reference/benches/fib.rs:114-124).  Everything numeric happens behind the C ABI; this file only wires handles
and parses the flat proof.

Proof word layout (canonical values), written by lurk_amd/csrc/prover.hip:
  header[10]: magic "LPRF", n_chips, log_blowup, num_queries, pow_bits, n_public, n_fri_layers, log_max_height,
              n_preprocessed, n_quotient_chunks
  per chip (prover order = height-sorted): machine index, log_n, width, prep width, permutation width (base columns),
              quotient degree, prep index + 1, cumulative_sum[4]
  public values[n_public]
  roots: main[8], permutation[8], quotient[8]
  opened values, round by round (preprocessed if any, main, permutation, quotient), matrix by matrix, point by point
              (zeta, then zeta * w_N; quotient chunks: zeta only), one extension element (4 words) per base column
  fri: commit-phase roots[n_layers][8], final_poly[4], pow_witness, query indices[num_queries]
  per round: record_words, then num_queries records [rows of every matrix | log_max(round) sibling digests]
  per fri layer: record_words, then num_queries records [8 words = the opened pair | sibling digests]
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field as dfield

import numpy as np

from . import _native as N
from . import field
from .air import ChipAir
from .context import Context, _addr, as_u32
from .lair import MEM_TABLE_SIZES, BytesChip, FuncChip, MemChip, QueryRecord, Shard, ShardingConfig, Toplevel, entrypoint_trace

PROOF_MAGIC = 0x4652504C
LOG_BLOWUP = 1          # sphinx BabyBearPoseidon2 [UPSTREAM-RECALL]
NUM_QUERIES = 100       # env FRI_QUERIES upstream
POW_BITS = 16


class Challenger:
    def __init__(self, ctx: Context, handle=None):
        self.ctx = ctx
        if handle is None:
            handle = C.c_void_p()
            ctx.check(N.lib.lurkhip_challenger_new(ctx.handle, C.byref(handle)))
        self.handle = handle

    def clone(self) -> "Challenger":
        h = C.c_void_p()
        N.check(N.lib.lurkhip_challenger_clone(self.handle, C.byref(h)))
        return Challenger(self.ctx, h)

    def observe(self, values):
        v = as_u32(values).reshape(-1)
        N.check(N.lib.lurkhip_challenger_observe(self.handle, _addr(v), len(v)))

    def sample(self, n: int = 1):
        out = np.zeros(n, dtype=np.uint32)
        N.check(N.lib.lurkhip_challenger_sample(self.handle, _addr(out), n))
        return [int(x) for x in out]

    def sample_ext(self):
        return tuple(self.sample(4))

    def sample_bits(self, bits: int) -> int:
        out = C.c_uint32()
        N.check(N.lib.lurkhip_challenger_sample_bits(self.handle, bits, C.byref(out)))
        return int(out.value)

    def __del__(self):
        if getattr(self, "handle", None) and N is not None:
            N.lib.lurkhip_challenger_free(self.handle)
            self.handle = None


@dataclass
class ChipProof:
    machine_index: int
    log_n: int
    width: int
    prep_width: int
    perm_width: int  # base columns
    quotient_degree: int
    prep_index: int  # -1: none
    cumulative_sum: tuple
    opened: dict = dfield(default_factory=dict)  # "prep" / "main" / "perm": (local, next) lists of EF tuples; "quotient": [chunk][4]


@dataclass
class ShardProof:
    log_blowup: int
    num_queries: int
    pow_bits: int
    log_max_height: int
    chips: list
    public_values: list
    main_root: list
    perm_root: list
    quot_root: list
    fri_roots: list
    final_poly: tuple
    pow_witness: int
    query_indices: list
    round_openings: list  # per round: (record_words, [records])
    layer_openings: list
    n_preprocessed: int
    words: np.ndarray = None


def parse_proof(words: np.ndarray) -> ShardProof:
    w = [int(x) for x in words]
    pos = [0]

    def take(n):
        out = w[pos[0]:pos[0] + n]
        if len(out) != n:
            raise ValueError("truncated proof")
        pos[0] += n
        return out

    def take_ef(n):
        flat = take(4 * n)
        return [tuple(flat[4 * i:4 * i + 4]) for i in range(n)]

    magic, n_chips, log_blowup, nq, pow_bits, n_public, n_layers, log_max, n_prep, n_chunks = take(10)
    if magic != PROOF_MAGIC:
        raise ValueError("not a lurkhip proof (bad magic)")
    chips = []
    for _ in range(n_chips):
        mi, log_n, width, pw, permw, qd, pidx = take(7)
        chips.append(ChipProof(mi, log_n, width, pw, permw, qd, pidx - 1, tuple(take(4))))
    public = take(n_public)
    main_root, perm_root, quot_root = take(8), take(8), take(8)
    if n_prep:
        # preprocessed round: matrices in key order; map back to chips through prep_index
        by_idx = {c.prep_index: c for c in chips if c.prep_index >= 0}
        for m in range(n_prep):
            c = by_idx[m]
            c.opened["prep"] = (take_ef(c.prep_width), take_ef(c.prep_width))
    for c in chips:
        c.opened["main"] = (take_ef(c.width), take_ef(c.width))
    for c in chips:
        c.opened["perm"] = (take_ef(c.perm_width), take_ef(c.perm_width))
    for c in chips:
        c.opened["quotient"] = [take_ef(4) for _ in range(c.quotient_degree)]
    if sum(c.quotient_degree for c in chips) != n_chunks:
        raise ValueError("quotient chunk count mismatch")
    fri_roots = [take(8) for _ in range(n_layers)]
    final_poly = tuple(take(4))
    pow_witness = take(1)[0]
    indices = take(nq)
    n_rounds = 3 + (1 if n_prep else 0)
    rounds = []
    for _ in range(n_rounds):
        rw = take(1)[0]
        rounds.append((rw, [take(rw) for _ in range(nq)]))
    layers = []
    for _ in range(n_layers):
        rw = take(1)[0]
        layers.append((rw, [take(rw) for _ in range(nq)]))
    if pos[0] != len(w):
        raise ValueError("trailing words in proof")
    return ShardProof(log_blowup, nq, pow_bits, log_max, chips, public, main_root, perm_root, quot_root, fri_roots, final_poly,
                      pow_witness, indices, rounds, layers, n_prep, words)


def alloc_trace_buffers(shapes, device):
    """Device tensors for the traces of one shard, shapes = [(height, width)]: the matrices of one height are column ranges of ONE
    buffer whose row pitch is a multiple of 32 words where the library says that pays (lurkhip_trace_group_layout: the coset
    LDE's first pass then reads whole 128-byte lines), dense tensors of their own otherwise.  Every trace kernel and the shard
    commitment take the pitch (tensor.stride(0))."""
    import torch

    n = len(shapes)
    if n == 0:
        return []
    lh = np.array([h.bit_length() - 1 for h, _ in shapes], dtype=np.uint32)
    ws = np.array([w for _, w in shapes], dtype=np.uint32)
    pitch, col = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
    grp, ng = np.zeros(n, dtype=np.int32), np.zeros(1, dtype=np.int32)
    N.check(N.lib.lurkhip_trace_group_layout(n, _addr(lh), _addr(ws), _addr(pitch), _addr(col), _addr(grp), _addr(ng)))
    bufs = {}
    out = []
    for i, (h, w) in enumerate(shapes):
        g = int(grp[i])
        if g not in bufs:
            bufs[g] = torch.empty((h, int(pitch[i])), dtype=torch.int32, device=device)  # every word a reader touches is written by a trace kernel
        out.append(bufs[g][:, int(col[i]):int(col[i]) + w])
    return out


class _ShardProver:
    """commit / prove / free of one shard's chip list through the C ABI; shared by the Lair `Machine` and the generic
    `StarkMachine`.  Subclasses provide `self.ctx`, `self.pk`, `self.chips` and `_prep_index(machine_index)`."""

    _lane_pool = None  # the second prove lane's worker thread (prove_lanes), started on first use

    def _prep_index(self, machine_index: int) -> int:
        return -1

    def close(self):
        if self.pk:
            N.lib.lurkhip_pk_free(self.ctx.handle, self.pk)
            self.pk = None
        pool = self._lane_pool
        if pool is not None:
            pool.shutdown(wait=True)
            self._lane_pool = None
        if self._lane_ctx is not None:
            self._lane_ctx.close()
            self._lane_ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def commit_shard(self, traces):
        n = len(traces)
        airs = (C.c_void_p * n)(*[a.handle.value for _, a, _, _ in traces])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, _, _, t in traces])
        lh = np.array([lg for _, _, lg, _ in traces], dtype=np.uint32)
        # a trace may be a column range of an aligned group buffer (alloc_trace_buffers): its rows are stride(0) words apart
        pitches = np.array([t.stride(0) for _, _, _, t in traces], dtype=np.uint32)
        prep_idx = np.array([self._prep_index(mi) for mi, _, _, _ in traces], dtype=np.int32)
        h = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        self.ctx.check(N.lib.lurkhip_shard_commit_pitched(self.ctx.handle, n, C.cast(airs, C.c_void_p), _addr(lh), C.cast(ptrs, C.c_void_p), _addr(pitches),
                                                          _addr(prep_idx), LOG_BLOWUP, C.byref(h), _addr(root)))
        self._included = getattr(self, "_included", {})
        self._included[h.value] = [mi for mi, _, _, _ in traces]
        return h, [int(x) for x in root]

    def prove_shard(self, shard_handle, challenger: Challenger, public_values, num_queries=NUM_QUERIES, pow_bits=POW_BITS, parse=True, ctx=None):
        """`ctx`: prove on that context's stream (same device and protocol profile; the shard must be fully committed, i.e. the
        committing context synchronised) -- a second lane for the latency chains of a multi-shard proof (prove_lanes)."""
        pv = as_u32(public_values)
        p = C.c_void_p()
        ctx = ctx or self.ctx
        ctx.check(N.lib.lurkhip_shard_prove(ctx.handle, self.pk, shard_handle, challenger.handle, _addr(pv), len(pv), num_queries, pow_bits,
                                            C.byref(p)))
        n = int(N.lib.lurkhip_proof_words(p))
        words = np.zeros(n, dtype=np.uint32)
        N.check(N.lib.lurkhip_proof_read(p, _addr(words), n))
        N.lib.lurkhip_proof_free(p)
        # the C ABI numbers chips by their position in the shard's chip list: map back to the machine's chip vector (in the
        # words too, so that they stand on their own: lurk_amd/proofs.py names chips by machine index)
        included = self._included[shard_handle.value]
        for i in range(int(words[1])):
            words[10 + 11 * i] = included[int(words[10 + 11 * i])]
        return parse_proof(words) if parse else words

    def free_shard(self, shard_handle):
        self._included.pop(shard_handle.value, None)
        N.lib.lurkhip_shard_free(self.ctx.handle, shard_handle)


class StarkMachine(_ShardProver):
    """A machine over an explicit chip list without preprocessed traces (sphinx `StarkMachine::new(config, chips, n)`), for
    chips that live outside a Lair toplevel -- e.g. the narrow Poseidon2 chip."""

    def __init__(self, ctx: Context, airs):
        self.ctx = ctx
        self.chips = [("air", None, a) for a in airs]
        self.pk = None

    def setup(self):
        h = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        self.ctx.check(N.lib.lurkhip_setup(self.ctx.handle, 0, None, None, None, LOG_BLOWUP, C.byref(h), _addr(root)))
        self.pk = h
        self.vk_root = [int(x) for x in root]
        return self.vk_root

    def prove(self, traces, public_values=(), num_queries=NUM_QUERIES, pow_bits=POW_BITS):
        """traces[i]: device matrix (Montgomery, power-of-two height) of chip i, or None when the chip is not included.
        One shard; same transcript as Machine.prove."""
        if self.pk is None:
            self.setup()
        pv = np.array(list(public_values), dtype=np.uint32)
        included = [(mi, self.chips[mi][2], t.shape[0].bit_length() - 1, t) for mi, t in enumerate(traces) if t is not None]
        ch = Challenger(self.ctx)
        ch.observe(self.vk_root)
        ch.observe([0])
        handle, root = self.commit_shard(included)
        ch.observe(root)
        if len(pv):
            ch.observe(pv)
        try:
            return self.prove_shard(handle, ch, pv, num_queries, pow_bits)
        finally:
            self.free_shard(handle)

    def verify(self, proof, profile=None):
        """The host verifier (csrc/verify.cpp) on this machine's one-shard proof."""
        if self.pk is None:
            self.setup()
        return verify_machine_proof([air for _, _, air in self.chips], self.vk_root, [], [], [proof], profile)


class Machine(_ShardProver):
    """Chip vector of one Lair toplevel with `entry` as its entrypoint (lair_chip.rs:196-211)."""

    def __init__(self, ctx: Context, toplevel: Toplevel, entry: str, num_public_values: int):
        self.ctx, self.toplevel = ctx, toplevel
        self.entry_idx = toplevel.func_index(entry)
        self.num_public_values = num_public_values
        self.chips = [("entrypoint", None, ChipAir.for_entrypoint(self.entry_idx, num_public_values))]
        for i in range(toplevel.num_funcs()):
            self.chips.append(("func", i, ChipAir.for_func(toplevel, i)))
        for ml in MEM_TABLE_SIZES:
            self.chips.append(("mem", ml, ChipAir.for_mem(ml)))
        self.chips.append(("bytes", None, ChipAir.for_bytes()))
        self.compiled_traces = []  # names of the function chips whose trace generator runs compiled (compile_airs)
        self.side_stream = True     # run_prepared: the short chips' trace kernels on the context's side streams
        self._lane_ctx = None       # second proving context of multi-shard proofs (prove)
        self.pk = None
        self._prep = None

    # machine.setup(&LairMachineProgram): only the byte chip has a preprocessed trace
    def setup(self):
        import torch

        prep = BytesChip(self.ctx).generate_preprocessed_trace(repr=N.REPR_MONTY)
        self._prep = torch.from_numpy(prep.view(np.int32)).cuda(self.ctx.device)
        ptrs = (C.c_void_p * 1)(self._prep.data_ptr())
        lh = np.array([16], dtype=np.uint32)
        ws = np.array([6], dtype=np.uint32)
        h = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        self.ctx.check(N.lib.lurkhip_setup(self.ctx.handle, 1, C.cast(ptrs, C.c_void_p), _addr(lh), _addr(ws), LOG_BLOWUP, C.byref(h), _addr(root)))
        self.pk = h
        self.vk_root = [int(x) for x in root]
        return self.vk_root

    def _prep_index(self, machine_index: int) -> int:
        return 0 if self.chips[machine_index][0] == "bytes" else -1

    def verify(self, proofs, profile=None):
        """machine.verify(&vk, &proof, &mut challenger) (benches/fib.rs:105-133) on the host: raises VerificationError unless
        every shard proof checks out and the cumulative sums cancel.  `profile`: the ProtocolProfile the proofs were made
        under when it is not the default (ProtocolProfile.of(ctx))."""
        if self.pk is None:
            self.setup()
        return verify_machine_proof([air for _, _, air in self.chips], self.vk_root, [16], [6], proofs, profile)

    def shard_traces(self, shard: Shard):
        """[(machine index, air, log_height, device trace (Montgomery))] of the chips included in the shard
        (LairChip::included, lair_chip.rs:124-139)."""
        import torch

        out = []
        for mi, (kind, arg, air) in enumerate(self.chips):
            if kind == "entrypoint":
                if shard.index != 0:
                    continue
                t = torch.from_numpy(field.to_monty(entrypoint_trace(shard.queries)).view(np.int32)).cuda(self.ctx.device)
            elif kind == "func":
                chip = FuncChip(self.ctx, arg, self.toplevel)
                n, h, w = chip.trace_shape(shard)
                if n == 0:
                    continue
                # torch.empty, not zeros: a fill would be enqueued on torch's stream, which is not ordered with the
                # context's (non-blocking) stream, and could land after the trace kernel; the kernel writes every word
                t = torch.empty((h, w), dtype=torch.int32, device=f"cuda:{self.ctx.device}")
                chip.generate_trace_dev(shard, t, repr=N.REPR_MONTY)
            elif kind == "mem":
                if shard.index != 0:
                    continue
                t = torch.from_numpy(MemChip(self.ctx, arg).generate_trace(shard, repr=N.REPR_MONTY).view(np.int32)).cuda(self.ctx.device)
            else:
                t = torch.from_numpy(BytesChip(self.ctx).generate_trace(shard, repr=N.REPR_MONTY).view(np.int32)).cuda(self.ctx.device)
            out.append((mi, air, t.shape[0].bit_length() - 1, t))
        self.ctx.sync()
        return out

    def shard_cost(self, shard: Shard) -> int:
        """Main-trace cells (padded height x width) of the shard's function chips: the estimate of its proving work that
        `shards.assign_shards_balanced` deals by."""
        cells = 0
        for kind, arg, _ in self.chips:
            if kind == "func":
                n, h, w = FuncChip(self.ctx, arg, self.toplevel).trace_shape(shard)
                if n:
                    cells += h * w
        return cells

    def prepare_shard(self, shard: Shard, input_ctx=None, n_threads: int = 0):
        """Uploads every included chip's trace inputs once; returns [(machine index, air, log_height, out tensor,
        prepared inputs or None)] -- `run_prepared` then regenerates all traces on the device without touching the host.
        The FuncChips' row streams are flattened together on host threads (lurkhip_func_trace_prepare_many).
        `input_ctx`: flatten and upload on that context (its own stream and pool) instead of the machine's -- a streaming
        prover stages the next shard there while this context proves the current one; the call returns once the uploads
        have completed, and the inputs must be closed only after this context has finished with them."""
        import torch

        from .lair import PreparedFuncTrace

        ictx = input_ctx or self.ctx
        func_chips = [FuncChip(ictx, arg, self.toplevel) for kind, arg, _ in self.chips if kind == "func"]
        func_inputs = iter(PreparedFuncTrace.prepare_many(func_chips, shard, n_threads))
        out = []
        for mi, (kind, arg, air) in enumerate(self.chips):
            if kind == "entrypoint":
                if shard.index != 0:
                    continue
                t = torch.from_numpy(field.to_monty(entrypoint_trace(shard.queries)).view(np.int32)).cuda(self.ctx.device)
                out.append((mi, air, 0, t, None))
                continue
            if kind == "func":
                p = next(func_inputs)
                if p is None:
                    continue
            elif kind == "mem":
                if shard.index != 0:
                    continue
                p = PreparedFuncTrace(MemChip(ictx, arg), shard)
            else:
                p = PreparedFuncTrace(BytesChip(ictx), shard)
            out.append((mi, air, p.height.bit_length() - 1, None, p))
        todo = [k for k, e in enumerate(out) if e[4] is not None]
        for k, t in zip(todo, alloc_trace_buffers([(out[k][4].height, out[k][4].width) for k in todo], f"cuda:{self.ctx.device}")):
            out[k] = out[k][:3] + (t,) + out[k][4:]
        if input_ctx is None:
            ictx.sync()
            torch.cuda.synchronize()
            return out
        # staged on another context: nobody waits here -- the uploads are followed by an event the proving context's stream
        # waits for (run_prepared), so the staging thread is already flattening the next shard while these copies run
        out = PreparedShard(out)
        out.event = ictx.record_event()
        return out

    # ---- a prepared shard as bytes: rank 0 executes the program, every rank proves the shards it is dealt (shards.scatter_prepared)
    def export_prepared(self, prepared):
        """[(machine index, blob or None)] of `prepare_shard`'s result: each chip's kernel inputs as bytes
        (PreparedFuncTrace.export); None for the entrypoint chip, whose one row is the public values."""
        return [(mi, p.export() if p is not None else None) for mi, _, _, _, p in prepared]

    def import_prepared(self, entries, public_values):
        """`prepare_shard`'s result rebuilt from `export_prepared`'s entries on this machine's context -- on a process that never
        ran the program: chip AIRs come from the (compiled) toplevel, kernel inputs from the blobs, the entrypoint row from the
        public values."""
        import torch

        from .lair import PreparedFuncTrace

        out = []
        for mi, blob in entries:
            air = self.chips[mi][2]
            if blob is None:
                assert self.chips[mi][0] == "entrypoint"
                t = torch.from_numpy(field.to_monty(np.array([list(public_values)], dtype=np.uint32)).view(np.int32)).cuda(self.ctx.device)
                out.append((mi, air, 0, t, None))
            else:
                p = PreparedFuncTrace.from_blob(self.ctx, blob)
                assert p.width == air.width, (air.name, p.width, air.width)
                out.append((mi, air, p.height.bit_length() - 1, None, p))
        todo = [k for k, e in enumerate(out) if e[4] is not None]
        for k, t in zip(todo, alloc_trace_buffers([(out[k][4].height, out[k][4].width) for k in todo], f"cuda:{self.ctx.device}")):
            out[k] = out[k][:3] + (t,) + out[k][4:]
        self.ctx.sync()
        torch.cuda.synchronize()
        return out

    def compile_airs(self, prepared, min_log_rows: int | None = None, min_instrs: int | None = None):
        """Compile (hiprtc, or the on-disk code-object cache) the AIR programs of the chips whose traces in `prepared` have at
        least 2^min_log_rows rows or whose constraint program has at least min_instrs instructions (the Poseidon2 chips: on
        the interpreter their quotient is a millisecond-long dependent chain whatever their height): their permutation traces
        and quotients then run straight-line device code.  A chip that fails to compile keeps the interpreter; returns the
        names of the compiled chips."""
        from . import jit_warm

        lo = jit_warm.COMPILE_MIN_LOG_ROWS if min_log_rows is None else min_log_rows
        ni = jit_warm.COMPILE_MIN_INSTRS if min_instrs is None else min_instrs
        done = []
        self.compile_failures = getattr(self, "compile_failures", [])  # [(what, error)]: a caller that counts on compiled kernels looks here
        for mi, chip_air, lg, _, _ in prepared:
            if lg >= lo or chip_air.constraint_instrs >= ni:
                try:
                    chip_air.compile(self.ctx)
                    done.append(chip_air.name)
                except Exception as e:  # hiprtc unavailable / compilation error: the interpreter stays in place
                    self.compile_failures.append((f"AIR of {chip_air.name}", str(e)[:300]))
            # the same for the trace generator of a tall function chip: its micro-program as a straight-line row kernel
            kind, arg, _ = self.chips[mi]
            if kind == "func" and lg >= lo:
                try:
                    FuncChip(self.ctx, arg, self.toplevel).compile_trace()
                    self.compiled_traces.append(chip_air.name)
                except Exception as e:
                    self.compile_failures.append((f"trace generator of {chip_air.name}", str(e)[:300]))
        return done

    def run_prepared(self, prepared):
        ev = getattr(prepared, "event", None)
        if ev:
            self.ctx.wait_event(ev)  # inputs uploaded on a staging context: its copies first
        # One call for the shard's chips: the short ones (hash chips: one Poseidon2 witness per lane, a few waves in all; ingress /
        # egress / lurk_main; memory tables) are latency-bound launches with nothing to fill the device -- the library deals them to
        # the context's side streams, forked behind everything queued so far and joined before the commitment, where they run
        # under one another and under the tall chips' kernels (lurkhip_func_trace_run_many; round 3 ran them one after another on
        # one side stream: 0.95 ms of a 2^12-row proof).
        todo = [(t, p) for _, _, lg, t, p in prepared if p is not None]
        if self.side_stream and len(todo) > 1:
            ps = (C.c_void_p * len(todo))(*[p.handle for _, p in todo])
            outs = (C.c_void_p * len(todo))(*[_addr(t) for t, _ in todo])
            pitches = np.array([t.stride(0) for t, _ in todo], dtype=np.uint32)
            self.ctx.check(N.lib.lurkhip_func_trace_run_many_pitched(self.ctx.handle, len(todo), ps, outs, _addr(pitches), N.REPR_MONTY))
        else:
            for t, p in todo:
                p.run(t, repr=N.REPR_MONTY, ctx=self.ctx)
        return [(mi, air, lg, t) for mi, air, lg, t, _ in prepared]

    def prove(self, queries: QueryRecord, config: ShardingConfig | None = None, num_queries=NUM_QUERIES, pow_bits=POW_BITS, resident_shards: int = 1,
              parse=True, lanes: int = 2):
        """machine.prove: commit every shard's main traces, observe (preprocessed root, pc_start = 0, then per shard
        the main root and the public values), prove every shard with a clone of that transcript
        [UPSTREAM-RECALL: sphinx LocalProver::prove_shards].

        Like sphinx, phase 1 keeps only the main roots: a shard's traces and main commitment (about 15 GB for a 2^22-row
        shard) are dropped once its root is known and regenerated in phase 2 (trace generation + main commit are under a
        third of a shard's proving time), so HBM holds `resident_shards` shards at a time instead of all of them; the last
        `resident_shards` shards of phase 1 stay resident.  Phase 2 keeps `lanes` (2) shards in flight on two contexts of the GPU
        (prove_lanes), so up to max(resident_shards, lanes) shards are resident then.  The proofs do not depend on that schedule."""
        if self.pk is None:
            self.setup()
        full = Shard.new(queries)
        shards = full.shard(config) if config is not None else [full]
        pv = queries.expect_public_values()
        ch = Challenger(self.ctx)
        ch.observe(self.vk_root)
        ch.observe([0])
        keep_from = max(0, len(shards) - max(1, resident_shards))
        committed, roots = {}, []
        for i, sh in enumerate(shards):
            traces = self.shard_traces(sh)
            handle, root = self.commit_shard(traces)
            roots.append(root)
            if i >= keep_from:
                committed[i] = (handle, traces)
            else:
                self.free_shard(handle)
                del traces
            ch.observe(root)
            ch.observe(pv)
        # phase 2, `lanes` shards at a time: each batch is brought back (regenerated unless it stayed resident) and proved on two
        # contexts concurrently (prove_lanes) -- a multi-shard proof keeps two shards in flight by default
        lanes = max(1, min(lanes, len(shards)))
        lane_ctx = None
        if lanes > 1:
            if self._lane_ctx is None:
                self._lane_ctx = lane_context(self)  # a stream measured to run beside this context's (Context(beside=...))
            lane_ctx = lane_context(self, self._lane_ctx)
        proofs = []
        for at in range(0, len(shards), lanes):
            batch = []
            for i in range(at, min(at + lanes, len(shards))):
                if i in committed:
                    handle, traces = committed.pop(i)
                else:
                    traces = self.shard_traces(shards[i])
                    handle, root = self.commit_shard(traces)
                    if root != roots[i]:
                        raise RuntimeError(f"shard {i}: regenerated main commitment differs from phase 1")
                batch.append((handle, traces))
            proofs += prove_lanes(self, [h for h, _ in batch], ch, pv, num_queries, pow_bits, parse=parse, lane_ctx=lane_ctx)
            for handle, _ in batch:
                self.free_shard(handle)
            del batch
        return proofs


# stream priority of a second prove lane (0 = equal queues, > 0 lower; measured on the two-shard bench step: no difference
# between 0, 1 and -1, so the queues stay equal; LURKHIP_LANE_PRIORITY for A/B measurements)
LANE_PRIORITY = int(os.environ.get("LURKHIP_LANE_PRIORITY", "0"))


def lane_context(machine, ctx=None):
    """A second context for `prove_lanes` on the machine's device with the machine context's protocol profile."""
    from .profile import ProtocolProfile

    # (its stream is measured to run beside the machine context's: Context(beside=...))
    ctx = ctx or (Context(machine.ctx.device, priority=LANE_PRIORITY) if LANE_PRIORITY else Context(beside=machine.ctx))
    ProtocolProfile.of(machine.ctx).install(ctx)
    return ctx


LANE_OFFSET_MAX_S = 0.25  # cap of prove_lanes' lane_offset_s


def prove_lanes(machine, handles, transcript: Challenger, pv, num_queries, pow_bits, parse=True, lane_ctx=None, lane_offset_s=0.0):
    """Phase 2 of a multi-shard proof on two lanes: the committed shards `handles` are proved alternately on the machine's
    context and on `lane_ctx` (its own stream, one host thread each), every shard with a clone of `transcript` -- while one
    shard sits in a latency chain (FRI layers, tree tails, transcript round trips) the other's big kernels fill the device
    (+14 % shards per second on the fib-mix shard).  The proofs are the sequential ones, in shard order.  One lane when
    `lane_ctx` is None or there is a single shard."""
    if lane_ctx is None or len(handles) < 2:
        return [machine.prove_shard(h, transcript.clone(), pv, num_queries, pow_bits, parse=parse) for h in handles]
    machine.ctx.sync()  # the commitments the second lane reads were made on this context's stream
    proofs, errors = [None] * len(handles), []

    # Out of phase (lane_offset_s > 0, the caller's choice; default none): two lanes that start together run the same stages at the
    # same time; with the second lane half a SEQUENTIAL proof behind, one lane's commitments (hashing) run under the other's openings
    # and FRI (bench.py --stagger-ms: 40.0 against 41.4 ms per proof, the delay included).  Round 3 derived the offset from the last
    # proof time seen -- measured with two proofs in flight, i.e. about twice the sequential time -- on a field written from the
    # worker thread; now the library never sleeps unless asked to, and never longer than LANE_OFFSET_MAX_S.
    import time

    offset_s = min(max(float(lane_offset_s), 0.0), LANE_OFFSET_MAX_S) if len(handles) >= 4 else 0.0

    def lane(j, ctx):
        try:
            if j == 1 and offset_s:
                time.sleep(offset_s)
            for i in range(j, len(handles), 2):
                proofs[i] = machine.prove_shard(handles[i], transcript.clone(), pv, num_queries, pow_bits, parse=parse, ctx=ctx)
            (ctx or machine.ctx).sync()
        except BaseException as e:  # surfaced after the join
            errors.append(e)

    # lane 0 on the calling thread, lane 1 on ONE long-lived worker of the machine: a host thread that has made HIP calls is not
    # cheap to start and end (two fresh threads per call cost a bench step 4.5 or 9 ms every other time)
    pool = machine._lane_pool
    if pool is None:
        from concurrent.futures import ThreadPoolExecutor

        pool = machine._lane_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="lurkhip-lane")
    other = pool.submit(lane, 1, lane_ctx)
    lane(0, None)
    other.result()
    if errors:
        raise errors[0]
    return proofs


class PreparedShard(list):
    """prepare_shard's result when the inputs were uploaded on a staging context: the list plus the event behind its uploads."""
    event = None

    def close(self):
        for *_, p in self:
            if p is not None:
                p.close()
        if self.event:
            Context.destroy_event(self.event)
            self.event = None


def prove_streamed(machine: "Machine", queries: QueryRecord, config: ShardingConfig, num_queries=NUM_QUERIES, pow_bits=POW_BITS, input_ctx=None,
                   n_threads: int = 0, parse=True, prepared=None, stats=None, lanes: int = 2):
    """`Machine.prove` fed by a host pipeline: a staging thread flattens shard k + 1 (host threads, page-locked staging) and
    uploads it on `input_ctx` -- its own stream, so the copy runs under the kernels of this context -- while the machine's
    context generates the traces of shard k and commits them.  Every shard's inputs, traces and main commitment stay resident
    (a 2^20-row fib shard holds 0.7 GB of inputs and 4 GB of main LDEs: a dozen shards fit the 288 GB), so phase 2 proves the
    shards without regenerating anything.  The proofs are `Machine.prove`'s, in shard order.
    `prepared`: inputs staged beforehand ([prepare_shard result per shard]) -- the resident-input reference the streamed run is
    measured against.  `stats` (dict) receives the host seconds spent staging.  `lanes` = 2: phase 2 proves the shards on two
    contexts concurrently (prove_lanes)."""
    import queue
    import threading
    import time

    if machine.pk is None:
        machine.setup()
    shards = Shard.new(queries).shard(config)
    pv = queries.expect_public_values()
    ready: "queue.Queue" = queue.Queue(maxsize=2)
    staged_s = [0.0]
    # Phase 1 takes the shards LIGHTEST FIRST: nothing overlaps the staging of the first shard, and the reference's sharding
    # makes the first shard the heaviest (every chip) and the last ones the lightest (the tallest chip only).  The main
    # commitments do not depend on one another; the transcript observes the roots in shard order afterwards.
    order = sorted(range(len(shards)), key=lambda i: machine.shard_cost(shards[i])) if prepared is None else list(range(len(shards)))

    def stage():
        try:
            for i in order:
                t0 = time.perf_counter()
                item = machine.prepare_shard(shards[i], input_ctx=input_ctx, n_threads=n_threads)
                staged_s[0] += time.perf_counter() - t0
                ready.put(item)
        except BaseException as e:  # surfaced on the proving thread
            ready.put(e)

    th, own_ctx = None, None
    if prepared is None:
        if input_ctx is None:  # the staging thread needs a stream and a pool of its own
            input_ctx = own_ctx = Context(beside=machine.ctx)  # (it is the second proving lane of phase 2 too)
        th = threading.Thread(target=stage)
        th.start()
    ch = Challenger(machine.ctx)
    ch.observe(machine.vk_root)
    ch.observe([0])
    committed, inputs, roots = [None] * len(shards), [], [None] * len(shards)
    try:
        for i in order:
            item = prepared[i] if prepared is not None else ready.get()
            if isinstance(item, BaseException):
                raise item
            inputs.append(item)
            traces = machine.run_prepared(item)
            handle, roots[i] = machine.commit_shard(traces)
            committed[i] = (handle, traces)
        for root in roots:
            ch.observe(root)
            ch.observe(pv)
        # phase 2 on two lanes: the staging context is idle by now and serves as the second one
        lane_ctx = None
        if lanes > 1 and len(committed) > 1:
            if input_ctx is None:
                input_ctx = own_ctx = Context(beside=machine.ctx)
            if th is not None:
                th.join()
            lane_ctx = lane_context(machine, input_ctx)
        proofs = prove_lanes(machine, [h for h, _ in committed], ch, pv, num_queries, pow_bits, parse=parse, lane_ctx=lane_ctx)
    finally:
        if th is not None:
            while th.is_alive():  # a failure on this side: drain so the staging thread can finish
                try:
                    ready.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
        machine.ctx.sync()
        for c in committed:
            if c is not None:
                machine.free_shard(c[0])
        if prepared is None:
            for item in inputs:
                item.close() if isinstance(item, PreparedShard) else [p.close() for *_, p in item if p is not None]
        if own_ctx is not None:
            own_ctx.close()
    if stats is not None:
        stats["staging_s"] = staged_s[0]
    return proofs


class VerificationError(Exception):
    """lurkhip_machine_verify rejected the proof (the reference's `verify` returning Err)."""


def verify_machine_proof(airs, vk_root, prep_log_heights, prep_widths, proofs, profile=None):
    """StarkMachine::verify on the host (csrc/verify.cpp; no device is used): `airs` = the ChipAir of every machine index,
    `proofs` = the shard proofs in shard order (ShardProof objects or their flat words), `profile` = a ProtocolProfile or None
    for the default preset.  Returns True; raises VerificationError with the verifier's reason otherwise."""
    words = [np.ascontiguousarray(p.words if hasattr(p, "words") else p, dtype=np.uint32) for p in proofs]
    air_ptrs = (C.c_void_p * len(airs))(*[a.handle for a in airs])
    proof_ptrs = (C.c_void_p * len(words))(*[w.ctypes.data for w in words])
    n_words = np.array([w.size for w in words], dtype=np.uint64)
    vk = np.array(vk_root, dtype=np.uint32)
    lh, ws = np.array(prep_log_heights, dtype=np.uint32), np.array(prep_widths, dtype=np.uint32)
    err = C.create_string_buffer(512)
    st = N.lib.lurkhip_machine_verify(C.byref(profile) if profile is not None else None, C.cast(air_ptrs, C.c_void_p), len(airs), _addr(vk),
                                      _addr(lh) if lh.size else None, _addr(ws) if ws.size else None, lh.size,
                                      C.cast(proof_ptrs, C.c_void_p), n_words.ctypes.data, len(words), err, len(err))
    if st != N.OK:
        raise VerificationError(f"lurkhip status {st}: {err.value.decode('utf-8', 'replace')}")
    return True


def grand_sum(proofs):
    """Sum of the cumulative sums of every chip of every shard proof (extension field, canonical lanes): zero for a
    consistent machine proof (the verifier's global LogUp check)."""
    acc = [0, 0, 0, 0]
    for p in proofs:
        for c in p.chips:
            for k in range(4):
                acc[k] = (acc[k] + int(c.cumulative_sum[k])) % field.P
    return tuple(acc)


def shard_sum(proof):
    return grand_sum([proof])


def prove_pipelined(machines, queries: QueryRecord, config: ShardingConfig | None = None, num_queries=NUM_QUERIES, pow_bits=POW_BITS):
    """`Machine.prove` with the shards dealt round-robin to several machines of the same toplevel, each on its own context
    (= its own HIP stream) of one GPU, and one host thread per machine: while one shard sits in a latency chain (FRI layers,
    tree tails, transcript round trips) the other's big kernels keep the device busy (bench.py: +18 % shards per second with
    two).  The proofs are the ones `Machine.prove` returns, in shard order: the transcript prefix (every shard's main root)
    is assembled before any shard is proved, exactly as there."""
    import threading

    for m in machines:
        if m.pk is None:
            m.setup()
    full = Shard.new(queries)
    shards = full.shard(config) if config is not None else [full]
    pv = queries.expect_public_values()
    k = len(machines)
    committed = [None] * len(shards)
    errors = []

    def run(target, *args):
        def wrapped():
            try:
                target(*args)
            except BaseException as e:  # surfaced after the join
                errors.append(e)
        return threading.Thread(target=wrapped)

    def commit_lane(j):
        m = machines[j]
        for i in range(j, len(shards), k):
            traces = m.shard_traces(shards[i])
            committed[i] = m.commit_shard(traces) + (traces,)  # the shard handle points into the trace buffers: keep them

    ths = [run(commit_lane, j) for j in range(k)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        raise errors[0]
    proofs = [None] * len(shards)

    def prove_lane(j):
        m = machines[j]
        for i in range(j, len(shards), k):
            handle = committed[i][0]
            c = Challenger(m.ctx)
            c.observe(m.vk_root)
            c.observe([0])
            for _, root, _ in committed:
                c.observe(root)
                c.observe(pv)
            proofs[i] = m.prove_shard(handle, c, pv, num_queries, pow_bits)
            m.free_shard(handle)
            committed[i] = (None, committed[i][1], None)

    ths = [run(prove_lane, j) for j in range(k)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        raise errors[0]
    return proofs
