"""The boundary's concurrency contract: entry points "may be called concurrently from rayon threads" -- the
reference calls `Chipset::{execute_simple, populate_witness}` from rayon row workers (/root/reference/src/lair/trace.rs:388-406),
`generate_trace` once per chip from rayon threads (/root/reference/src/lair/trace.rs:86-132, lair_chip.rs:96-120) and commits from
the prover (/root/reference/src/lair/chipset.rs:9-47 is the trait those calls go through).

Two regimes: (a) sixteen host threads, each with its OWN context, hammer lurkhip_poseidon2_hash8 / _wide_witness /
lurkhip_generate_trace_func / lurkhip_commit on ragged sizes for a few seconds, every result checked against the oracle (hashes,
roots) or the reference's literal matrices (traces); (b) the same calls from four threads on ONE context: every entry point takes
the context's lock for its duration (ctx.h: lurkhip_ctx::api_mu), so they serialise -- the results must be exactly as correct."""
import random
import threading
import time

import numpy as np
import pytest

import lurk_amd
from lair_helpers import load_cases
from lurk_amd import commit as cm
from lurk_amd import lair, synth
from lurk_amd.poseidon import PoseidonChipset

pytestmark = pytest.mark.gpu


def build_items(oracle):
    """(name, fn(ctx)) work items with their expected results computed up front, on this thread."""
    items = []
    for width in (24, 32, 40):
        for n in (1, 63, 65, 300, 1000):
            x = synth.field_elements((n, width), seed=4000 + 13 * width + n)
            want = oracle.p2_hash8(width, x)

            def hash_item(ctx, width=width, x=x, want=want):
                assert np.array_equal(PoseidonChipset(ctx, width).hash_batch(x), want)

            items.append((f"hash8[{width}x{n}]", hash_item))
    for width in (24, 32):
        for n in (1, 65, 257):
            x = synth.field_elements((n, width), seed=5000 + 13 * width + n)
            want = oracle.p2_wide_witness(width, x)

            def wit_item(ctx, width=width, x=x, want=want):
                assert np.array_equal(PoseidonChipset(ctx, width).witness_batch(x), want)

            items.append((f"wide_witness[{width}x{n}]", wit_item))
    for case in load_cases()[:5]:
        top = lair.Toplevel(case["source"], lurk_chips=case["lurk_chips"])
        q = lair.QueryRecord(top)
        for name, args in case["calls"]:
            top.execute_by_name(name, args, q)
        want = case["trace"]

        def trace_item(ctx, top=top, q=q, func=case["func"], want=want):
            got = lair.FuncChip.from_name(ctx, func, top).generate_trace(lair.Shard.new(q))
            assert got.flatten().tolist() == want

        items.append((f"generate_trace[{case['name']}]", trace_item))
    for k, shapes in enumerate([[(6, 5), (9, 13), (4, 3)], [(11, 7)], [(8, 78), (8, 33), (3, 2)]]):
        mats = [synth.field_elements((1 << lg, w), seed=6000 + 10 * k + i) for i, (lg, w) in enumerate(shapes)]
        root, _ = oracle.merkle_commit([oracle.lde(m, 1) for m in mats])

        def commit_item(ctx, mats=mats, root=root):
            c = cm.commit(ctx, mats, log_blowup=1)
            try:
                assert np.array_equal(c.root, root)
            finally:
                c.close()

        items.append((f"commit[{shapes}]", commit_item))
    return items


def hammer(ctxs, items, seconds):
    """One thread per entry of `ctxs` (entries may repeat: a shared context); each loops over the items in its own order."""
    deadline = time.monotonic() + seconds
    errors, counts = [], [0] * len(ctxs)

    def worker(k, ctx):
        rng = random.Random(900 + k)
        order = list(range(len(items)))
        try:
            while time.monotonic() < deadline and not errors:
                rng.shuffle(order)
                for i in order:
                    items[i][1](ctx)
                    counts[k] += 1
                    if time.monotonic() >= deadline or errors:
                        break
        except BaseException as e:  # noqa: BLE001 - reported by the test thread
            errors.append((k, e))

    ths = [threading.Thread(target=worker, args=(k, c)) for k, c in enumerate(ctxs)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return errors, counts


def test_sixteen_threads_with_their_own_contexts(oracle):
    items = build_items(oracle)
    ctxs = [lurk_amd.Context(0) for _ in range(16)]
    try:
        errors, counts = hammer(ctxs, items, 4.0)
    finally:
        for c in ctxs:
            c.close()
    assert not errors, errors[:2]
    assert min(counts) >= 1 and sum(counts) >= 5 * len(items)


def test_four_threads_on_one_shared_context(oracle):
    items = build_items(oracle)
    ctx = lurk_amd.Context(0)
    try:
        errors, counts = hammer([ctx] * 4, items, 3.0)
        assert not errors, errors[:2]
        assert min(counts) >= 1
        # ... and the context is intact afterwards
        for _, fn in items:
            fn(ctx)
    finally:
        ctx.close()


def test_an_entry_point_leaves_the_callers_device_alone(ctx):
    """a call used to leave the context's device current on the calling thread; it is restored on return."""
    import torch

    if torch.cuda.device_count() < 1:
        pytest.skip("no device")
    before = torch.cuda.current_device()
    PoseidonChipset(ctx, 24).hash_batch(synth.field_elements((3, 24), seed=1))
    assert torch.cuda.current_device() == before
