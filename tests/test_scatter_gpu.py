"""Prepared shards as bytes (round 5): the process that executed the program hands a shard's kernel inputs to the process that
proves it (`Shard::shard`, /root/reference/src/lair/execute.rs:186-216, cuts ONE QueryRecord: only the executing process holds it).
`lurkhip_func_trace_export` / `_import` through `Machine.export_prepared` / `import_prepared`: a machine built on a SECOND context
from nothing but the toplevel, the public values and the bytes produces the same traces, the same main root and the same proof
words as the machine that holds the record; malformed bytes are refused."""
import ctypes as C

import numpy as np
import pytest

import lurk_amd
from lurk_amd import _native as N
from lurk_amd import lair, prover
from lurk_amd.programs import lurk_mix as lm

pytestmark = pytest.mark.gpu


def _prove(ctx, m, traces, pv):
    ch = prover.Challenger(ctx)
    ch.observe(m.vk_root)
    ch.observe([0])
    handle, root = m.commit_shard(traces)
    ch.observe(root)
    ch.observe(pv)
    try:
        return root, m.prove_shard(handle, ch, pv, num_queries=6, pow_bits=4, parse=False)
    finally:
        m.free_shard(handle)


def test_imported_shards_prove_like_the_executing_machine(ctx):
    mix = lm.fib_mix(1 << 11)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    m.setup()
    ctx2 = lurk_amd.Context(0)
    try:
        # the "other rank": its own toplevel object (compiled from the same text), no record
        top2 = lair.Toplevel(mix.source, lurk_chips=True)
        m2 = prover.Machine(ctx2, top2, mix.entry, len(pv))
        assert m2.setup() == m.vk_root
        for sh in lair.Shard.new(q).shard(lair.ShardingConfig(1 << 10)):
            prep = m.prepare_shard(sh)
            entries = m.export_prepared(prep)
            assert [mi for mi, _ in entries] == [mi for mi, *_ in prep]
            assert (entries[0][1] is None) == (sh.index == 0)  # the entrypoint chip travels as the public values
            prep2 = m2.import_prepared(entries, pv)
            a, b = m.run_prepared(prep), m2.run_prepared(prep2)
            ctx.sync()
            ctx2.sync()
            for (_, air, lg, ta), (_, _, lg2, tb) in zip(a, b):
                assert lg == lg2 and np.array_equal(ta.cpu().numpy(), tb.cpu().numpy()), air.name
            root_a, words_a = _prove(ctx, m, a, pv)
            root_b, words_b = _prove(ctx2, m2, b, pv)
            assert root_a == root_b and np.array_equal(words_a, words_b)
    finally:
        ctx2.close()


def test_import_refuses_malformed_blobs(ctx):
    mix = lm.fib_mix(300)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    chip = lair.FuncChip(ctx, top.func_index("eval"), top)
    blob = lair.PreparedFuncTrace(chip, lair.Shard.new(q)).export()
    good = lair.PreparedFuncTrace.from_blob(ctx, blob)
    assert (good.n_real, good.width) == (300, 78)
    good.close()

    def refused(b):
        h = C.c_void_p()
        return N.lib.lurkhip_func_trace_import(ctx.handle, b.ctypes.data, b.nbytes, C.byref(h)) == N.ERR_INVALID_ARG and not h.value

    bad = blob.copy()
    bad[0] ^= 1                                   # magic
    assert refused(bad)
    assert refused(blob[:40].copy())              # shorter than the header
    w = blob.copy().view(np.uint32)
    w[6] += 1                                     # width no longer the program's
    assert refused(w.view(np.uint8))
    w = blob.copy().view(np.uint32)
    w[4] = w[5] + 1                               # more real rows than the height
    assert refused(w.view(np.uint8))
    w = blob.copy().view(np.uint32)
    w[12] = 0xFFFFFFF0                            # an offset past the block
    assert refused(w.view(np.uint8))
    # round 6: a truncated blob, a flipped word of the block, sections that overlap
    assert refused(blob[:len(blob) - 64].copy())  # truncated: the header's length says so
    bad = blob.copy()
    bad[len(bad) // 2] ^= 0x40                    # one bit of the block: the checksum
    assert refused(bad)
    w = blob.copy().view(np.uint32)
    w[14], w[15] = w[12], w[13]                   # the outputs' section starts where the arguments' does
    assert refused(w.view(np.uint8))
    w = blob.copy().view(np.uint32)
    w[9] += 1 << 20                               # more stream words than the block holds
    assert refused(w.view(np.uint8))
