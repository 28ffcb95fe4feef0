#!/usr/bin/env python3
"""GPU box only: determinism soak.  The same execution is proved over and over -- sequentially, with compiled chips, and with
two shards in flight on two contexts -- and every proof must be word-for-word the first one (the transcript is deterministic):
a race between streams, a stale cached table or a buffer handed out while still in use would show up as a differing proof.
usage: python tools/soak.py [rounds=30] [fib argument=60]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import lurk_amd  # noqa: E402
from lair_helpers import load_cases  # noqa: E402
from lurk_amd import lair, prover  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
arg = int(sys.argv[2]) if len(sys.argv) > 2 else 60
src = load_cases()[0]["source"]
top = lair.Toplevel(src)
q = lair.QueryRecord(top)
top.execute_by_name("fib", [arg], q)
pv = q.expect_public_values()
cfg = lair.ShardingConfig(16)
t0 = time.time()
with lurk_amd.Context(0) as ctx, lurk_amd.Context(0) as ctx2:
    m1 = prover.Machine(ctx, top, "fib", len(pv))
    m2 = prover.Machine(ctx2, top, "fib", len(pv))
    m1.setup()
    m2.setup()
    want = [p.words.copy() for p in m1.prove(q, cfg, num_queries=8, pow_bits=4)]
    print(f"{len(want)} shards per round, {sum(len(w) for w in want)} proof words")
    bad = 0
    for r in range(rounds):
        for name, got in (("sequential", m1.prove(q, cfg, num_queries=8, pow_bits=4)),
                          ("second context", m2.prove(q, cfg, num_queries=8, pow_bits=4)),
                          ("two in flight", prover.prove_pipelined([m1, m2], q, cfg, num_queries=8, pow_bits=4))):
            for i, (a, b) in enumerate(zip(got, want)):
                if not np.array_equal(a.words, b):
                    bad += 1
                    print(f"round {r}: {name}: shard {i} differs")
    m1.close()
    m2.close()
print(f"{rounds} rounds, {bad} differing proofs, {time.time() - t0:.1f} s")
bad_total = bad

# ---- the bench's step (device-resident inputs, compiled trace / AIR kernels, short chips' traces on a side stream, one sponge launch
# per tree) and the streamed multi-shard prover (staging context, two prove lanes), on a small fib-mix machine
from lurk_amd.programs import lurk_mix as lm  # noqa: E402

mix_rounds = max(10, rounds * 3)
mix = lm.fib_mix(1 << 15)  # tall chips (>= 2^13 rows) on the main stream, short ones on the side lanes: both in every proof
top = lair.Toplevel(mix.source, lurk_chips=True)
q = lair.QueryRecord(top)
top.execute(top.func_index(mix.entry), mix.main_args, q)
pv = q.expect_public_values()
t0 = time.time()
with lurk_amd.Context(0) as ctx, lurk_amd.Context(0) as ctx_in:
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    vk = m.setup()
    shard = lair.Shard.new(q)
    prepared = m.prepare_shard(shard)
    m.compile_airs(prepared, min_log_rows=12)

    def step():
        traces = m.run_prepared(prepared)
        handle, root = m.commit_shard(traces)
        ch = prover.Challenger(ctx)
        ch.observe(vk)
        ch.observe([0])
        ch.observe(root)
        ch.observe(pv)
        w = m.prove_shard(handle, ch, pv, num_queries=8, pow_bits=4, parse=False)
        m.free_shard(handle)
        return w

    want = step()
    bad = 0
    for r in range(mix_rounds):
        if not np.array_equal(step(), want):
            bad += 1
            print(f"fib-mix round {r}: proof differs")
    cfg = lair.ShardingConfig(1 << 13)
    ref = [p.words.copy() for p in m.prove(q, cfg, num_queries=8, pow_bits=4, lanes=1)]
    for r in range(max(5, rounds // 3)):
        for name, got in (("prove, two lanes", m.prove(q, cfg, num_queries=8, pow_bits=4)),
                          ("streamed", prover.prove_streamed(m, q, cfg, num_queries=8, pow_bits=4, input_ctx=ctx_in))):
            for i, (a, b) in enumerate(zip(got, ref)):
                if not np.array_equal(a.words, b):
                    bad += 1
                    print(f"fib-mix sharded round {r} {name}: shard {i} differs")
    m.close()
print(f"fib-mix: {mix_rounds} steps + {max(5, rounds // 3)} sharded rounds x 2 variants x {len(ref)} shards, {bad} differing, {time.time() - t0:.1f} s")
sys.exit(1 if bad_total + bad else 0)
