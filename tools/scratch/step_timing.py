import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import lurk_amd
from lurk_amd import lair, prover, shards
from lurk_amd.programs import lurk_mix as lm
ctx = lurk_amd.Context(0)
n = 1 << 20
mix = lm.fib_mix(n)
top = lair.Toplevel(mix.source, lurk_chips=True)
q = lair.QueryRecord(top); top.execute(top.func_index(mix.entry), mix.main_args, q)
pv = q.expect_public_values()
m = prover.Machine(ctx, top, mix.entry, len(pv)); vk = m.setup()
sh = lair.Shard.new(q).shard(lair.ShardingConfig(n))
prep = m.prepare_shard(sh[0]); m.compile_airs(prep)
T = {}
def tick(name, t0):
    ctx.sync(); T[name] = T.get(name, 0) + time.perf_counter() - t0
def step():
    t=time.perf_counter(); traces = m.run_prepared(prep); tick("run_prepared", t)
    t=time.perf_counter(); handle, root = m.commit_shard(traces); tick("commit_shard", t)
    t=time.perf_counter()
    ch = prover.Challenger(ctx); ch.observe(vk); ch.observe([0])
    for r in shards.exchange_roots([root], device="cpu"):
        ch.observe(r); ch.observe(pv)
    tick("transcript", t)
    t=time.perf_counter(); words = m.prove_shard(handle, ch, pv, num_queries=100, pow_bits=16, parse=False); tick("prove_shard", t)
    t=time.perf_counter(); m.free_shard(handle); tick("free_shard", t)
    t=time.perf_counter()
    n_chips = int(words[1]); cs = [words[10 + 11 * i + 7:10 + 11 * i + 11] for i in range(n_chips)]
    mine = np.zeros(4, dtype=np.int64)
    for c in cs: mine = (mine + np.asarray(c, dtype=np.int64)) % 2013265921
    g = shards.reduce_cumulative_sums(cs, device="cpu"); tick("sums", t)
    return words
step(); T.clear()
t0=time.perf_counter()
for _ in range(5): step()
tot=(time.perf_counter()-t0)/5*1e3
print("total ms/step (with syncs)", tot)
for k,v in T.items(): print("  %-14s %.3f ms"%(k, v/5*1e3))
# host-only cost of prove_shard's python wrapper: time with GPU idle is included above; separately time the numpy conversion
