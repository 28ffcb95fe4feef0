"""Times the host interpreter (lurkhip_execute) on the bench machine: python tools/scratch/interp_time.py [log_rows] [workload]"""
import sys, time
sys.path.insert(0, ".")
from lurk_amd import lair
from lurk_amd.programs import lurk_mix as lm

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
wl = sys.argv[2] if len(sys.argv) > 2 else "fib-mix"
mix = lm.fib_mix(1 << lg) if wl == "fib-mix" else lm.lurk_mix(1 << lg)
top = lair.Toplevel(mix.source, lurk_chips=True)
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    q = lair.QueryRecord(top)
    t0 = time.perf_counter()
    top.execute_by_name(mix.entry, list(mix.main_args), q)
    dt = time.perf_counter() - t0
    print(f"{wl} 2^{lg} eval rows: execute {dt:.3f} s = {dt / (1 << lg) * 1e9:.0f} ns per eval row", flush=True)
    del q
