"""Row pitches (round 5): the prover's own traces as column ranges of aligned group buffers.

`RowMajorMatrix::new(values, width)` (/root/reference/src/lair/trace.rs:133) is dense; the pitched entry points
(`lurkhip_trace_group_layout`, `lurkhip_func_trace_run[_many]_pitched`, `lurkhip_shard_commit_pitched`) exist so that the coset
LDE's first pass reads whole 128-byte lines.  A layout is not allowed to change a single word: the same shard through the dense
route (`Machine.shard_traces`: one contiguous matrix per chip) and through the pitched route (`prepare_shard` / `run_prepared`)
must give the same traces, the same main root and the same proof, on the interpreter kernels and on the compiled ones."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import lair, prover
from lurk_amd.programs import lurk_mix as lm

pytestmark = pytest.mark.gpu


def _machine(ctx, mix):
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    m.setup()
    return m, q, pv


def _prove(ctx, m, traces, pv):
    ch = prover.Challenger(ctx)
    ch.observe(m.vk_root)
    ch.observe([0])
    handle, root = m.commit_shard(traces)
    ch.observe(root)
    ch.observe(pv)
    try:
        words = m.prove_shard(handle, ch, pv, num_queries=6, pow_bits=4, parse=False)
    finally:
        m.free_shard(handle)
    return root, words


@pytest.mark.parametrize("compiled", [False, True], ids=["interpreter", "compiled"])
def test_pitched_traces_commit_and_proof_equal_dense(ctx, compiled, monkeypatch):
    # (the aligned layout is off by default -- measured: no gain for the LDE's first pass, a loss for the writers, DESIGN.md 3.3 --
    # and read from the environment on every call)
    monkeypatch.setenv("LURKHIP_SRC_PADDED", "1")
    mix = lm.fib_mix(1 << 11)
    m, q, pv = _machine(ctx, mix)
    shard = lair.Shard.new(q)
    prepared = m.prepare_shard(shard)
    if compiled:
        assert m.compile_airs(prepared, 0), "nothing compiled"
    pitched = m.run_prepared(prepared)
    ctx.sync()
    dense = m.shard_traces(shard)
    assert [mi for mi, *_ in pitched] == [mi for mi, *_ in dense]
    # the layout is in force: some trace is a column range of a wider buffer, and every padded pitch is a whole number of lines
    strides = [(t.stride(0), t.shape[1]) for _, _, _, t in pitched]
    assert any(s > w for s, w in strides), strides
    assert all(s == w or s % 32 == 0 for s, w in strides), strides
    for (_, air, _, a), (_, _, _, b) in zip(pitched, dense):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()), air.name
    root_p, words_p = _prove(ctx, m, pitched, pv)
    root_d, words_d = _prove(ctx, m, dense, pv)
    assert root_p == root_d
    assert np.array_equal(words_p, words_d)
    assert m.verify([prover.parse_proof(words_p)])


def test_single_pitched_run_matches_dense(ctx):
    """lurkhip_func_trace_run_pitched on one chip at an arbitrary pitch (not a multiple of anything), staged and unstaged kernels:
    the words of the matrix equal the dense run's and the words between rows stay untouched."""
    import torch

    mix = lm.fib_mix(300)
    m, q, pv = _machine(ctx, mix)
    shard = lair.Shard.new(q)
    checked = 0
    for kind, arg, air in m.chips:
        if kind != "func":
            continue
        chip = lair.FuncChip(ctx, arg, m.toplevel)
        n, h, w = chip.trace_shape(shard)
        if n == 0:
            continue
        p = lair.PreparedFuncTrace(chip, shard)
        dense = torch.empty((h, w), dtype=torch.int32, device="cuda")
        p.run(dense, repr=lurk_amd._native.REPR_MONTY)
        pitch = w + 5
        buf = torch.full((h, pitch), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        p.run(buf[:, 2:2 + w], repr=lurk_amd._native.REPR_MONTY)
        ctx.sync()
        got = buf.cpu().numpy()
        assert np.array_equal(got[:, 2:2 + w], dense.cpu().numpy()), air.name
        assert (got[:, :2] == -7).all() and (got[:, 2 + w:] == -7).all(), air.name
        p.close()
        checked += 1
    assert checked >= 10
