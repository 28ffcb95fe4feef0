import sys, time
sys.path.insert(0, ".")
import lurk_amd
from lurk_amd import lair, prover
from lurk_amd.programs import lurk_mix as lm
mix = lm.fib_mix(1 << 18)
top = lair.Toplevel(mix.source, lurk_chips=True)
q = lair.QueryRecord(top)
top.execute_by_name(mix.entry, mix.main_args, q)
pv = q.expect_public_values()
with lurk_amd.Context(0) as ctx:
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    m.setup()
    for log_shard in (12, 14):
        t0 = time.perf_counter()
        proofs = m.prove(q, lair.ShardingConfig(1 << log_shard), num_queries=20, pow_bits=8)
        t1 = time.perf_counter()
        ok = m.verify(proofs)
        t2 = time.perf_counter()
        print(f"2^18 eval rows in shards of 2^{log_shard}: {len(proofs)} proofs in {t1 - t0:.2f} s, verified {ok} in {t2 - t1:.2f} s, grand sum {prover.grand_sum(proofs)}, pool {ctx.pool_stats()}", flush=True)
    m.close()
