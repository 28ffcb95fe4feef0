"""Host pipeline (the callers' side of trace generation): the row streams of a shard flattened on host
threads into page-locked staging (lurkhip_func_trace_prepare_many) and a multi-shard proof fed by a staging thread on a
second context (prover.prove_streamed).  Reference: FuncChip::generate_trace's per-row parallelism,
/root/reference/src/lair/trace.rs:86-132; LocalProver::prove_shards [UPSTREAM-RECALL]."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import lair, prover
from lurk_amd import _native as N
from lurk_amd.programs import lurk_mix as lm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with lurk_amd.Context(0) as c:
        yield c


def _traces(ctx, prepared):
    import torch

    outs = []
    for _, _, _, t, p in prepared:
        if p is not None:
            p.run(t, repr=N.REPR_MONTY, ctx=ctx)
    ctx.sync()
    torch.cuda.synchronize()
    for _, _, _, t, _ in prepared:
        outs.append(t.cpu().numpy().copy())
    return outs


@pytest.mark.parametrize("n_threads", [1, 3, 0])
def test_prepare_many_equals_one_by_one(ctx, n_threads):
    """Every function of a sharded fib-mix execution: the batched, multi-threaded flattening gives the traces of the
    one-function-at-a-time path, shard by shard (ranges of 2^14 rows, so 2^16 eval rows exercise several ranges per function)."""
    import torch

    mix = lm.fib_mix(1 << 16)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute(top.func_index(mix.entry), mix.main_args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    for sh in lair.Shard.new(q).shard(lair.ShardingConfig(1 << 15)):
        prepared = m.prepare_shard(sh, n_threads=n_threads)
        got = _traces(ctx, prepared)
        want = []
        for kind, arg, _ in m.chips:
            if kind != "func":
                continue
            chip = lair.FuncChip(ctx, arg, top)
            n, h, w = chip.trace_shape(sh)
            if n == 0:
                continue
            t = torch.empty((h, w), dtype=torch.int32, device="cuda")
            p = lair.PreparedFuncTrace(chip, sh)
            p.run(t, repr=N.REPR_MONTY)
            ctx.sync()
            want.append(t.cpu().numpy().copy())
            p.close()
        func_got = [g for g, (mi, *_) in zip(got, prepared) if m.chips[mi][0] == "func"]
        assert len(func_got) == len(want)
        for a, b in zip(func_got, want):
            assert np.array_equal(a, b)
    m.close()


def test_streamed_proof_equals_machine_prove(ctx):
    """prove_streamed (staging thread on a second context, everything resident for phase 2) returns Machine.prove's proofs."""
    mix = lm.fib_mix(1 << 10)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute(top.func_index(mix.entry), mix.main_args, q)
    pv = q.expect_public_values()
    cfg = lair.ShardingConfig(1 << 8)
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    m.setup()
    want = m.prove(q, cfg, num_queries=4, pow_bits=2, lanes=1)  # one shard at a time, regenerated in phase 2
    two = m.prove(q, cfg, num_queries=4, pow_bits=2)            # the default: two shards in flight in phase 2 (prove_lanes)
    assert len(two) == len(want)
    for a, b in zip(two, want):
        assert np.array_equal(a.words, b.words)
    with lurk_amd.Context(0) as ctx2:
        stats = {}
        got = prover.prove_streamed(m, q, cfg, num_queries=4, pow_bits=2, input_ctx=ctx2, stats=stats)
        assert stats["staging_s"] > 0
    # staged beforehand on the machine's own context: the resident-input reference
    prepared = [m.prepare_shard(sh) for sh in lair.Shard.new(q).shard(cfg)]
    again = prover.prove_streamed(m, q, cfg, num_queries=4, pow_bits=2, prepared=prepared)
    m.close()
    assert len(got) == len(want) == len(again) >= 4
    for a, b, c in zip(got, want, again):
        assert np.array_equal(a.words, b.words)
        assert np.array_equal(c.words, b.words)
    assert prover.grand_sum(got) == (0, 0, 0, 0)


def test_second_lane_on_a_lower_priority_context(ctx):
    """lurkhip_ctx_create_with_priority: the second prove lane on a context whose streams have a lower priority, called twice
    (the lane's worker thread is the machine's, started once and reused): the proofs are the one-lane ones."""
    mix = lm.fib_mix(1 << 10)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute(top.func_index(mix.entry), mix.main_args, q)
    pv = q.expect_public_values()
    cfg = lair.ShardingConfig(1 << 8)
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    m.setup()
    want = m.prove(q, cfg, num_queries=4, pow_bits=2, lanes=1)
    with lurk_amd.Context(0, priority=1) as low, lurk_amd.Context(0, priority=-1) as high:
        for lane_ctx in (low, high, low):
            got = prover.prove_streamed(m, q, cfg, num_queries=4, pow_bits=2, input_ctx=lane_ctx)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert np.array_equal(a.words, b.words)
        assert m._lane_pool is not None  # one worker for all three calls
    m.close()
    assert m._lane_pool is None
