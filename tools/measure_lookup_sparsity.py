#!/usr/bin/env python3
"""How sparse are the lookups of a real Lurk evaluation?  (Build container only: reads /root/reference at run time through
tools/lurk_reference.py, stores numbers only.)

Every branch of a Lair function carries its own `require`s (air/builder.rs:75-104: one receive and one send each), and a row
takes ONE branch: of the 156 interactions of eval_builtin_expr a row of `(fib N)` has a handful with a non-zero multiplicity.
sphinx batches interactions two to a permutation column (log_quotient_degree = 1 for every Lurk chip); a column whose two
interactions are never real in a shard is identically zero.  This script counts, on the ORACLE's traces of a real `(fib N)` (the
reference's functions, oracle/lair.py generate_trace + oracle/air.py), per chip: interactions, interactions that are real on at
least one row, permutation columns, columns with no real interaction -- and the same on the fib-mix stand-in the bench proves
(lurk_amd/programs/lurk_mix.py), so that what the device skips on the stand-in can be held against what it would skip on the
real machine.

    python tools/measure_lookup_sparsity.py [N] [--write]     -> prints the table; --write merges it into tests/golden/fib_shape.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def chip_liveness(air, rows, prep=None, public=()):
    """[live?] per interaction in sphinx order (sends, then receives) over all rows of one chip."""
    from oracle import air as oa

    live = None
    h = len(rows)
    for r in range(h):
        n = (r + 1) % h
        b = oa.Builder(rows[r], rows[n], prep[r] if prep is not None else (), prep[n] if prep is not None else (), public,
                       (1 if r == 0 else 0, 1 if r == h - 1 else 0, 0 if r == h - 1 else 1))
        air.eval(b)
        m = [int(x != 0) for x, _ in b.sends + b.receives]
        live = m if live is None else [a | c for a, c in zip(live, m)]
    return live or []


def summarise(live, batch=2, sel_live=None):
    cols = [(live[i:i + batch]) for i in range(0, len(live), batch)]
    out = {"interactions": len(live), "live_interactions": sum(live), "columns": len(cols), "dead_columns": sum(1 for c in cols if not any(c))}
    if sel_live is not None:
        out["selectors"], out["live_selectors"] = len(sel_live), sum(sel_live)
    return out


def oracle_machine(otop, q, witness, entry, pv):
    from oracle import air as oa
    from oracle import lair as ol

    out = {}
    for g in otop.funcs:
        if not q.func[g["index"]]:
            continue
        rows, _ = ol.generate_trace(otop, g["name"], q, witness=witness)
        if rows:
            nsel = otop.layout(g)["sel"]
            sel_live = [int(any(r[len(r) - nsel + k] for r in rows)) for k in range(nsel)]
            out[g["name"]] = (summarise(chip_liveness(oa.FuncAir(otop, g["name"]), rows, public=pv), sel_live=sel_live), len(rows))
    return out


def real_fib(n):
    import lurk_reference as lr
    import measure_lurk_shape as ms
    from lurk_amd import zstore as zs
    from oracle import binding
    from oracle import lair as ol
    from test_lair_gpu import oracle_chip_callbacks

    binding.build()
    real = ms.RealLurk()
    otop = ol.Toplevel(real.source, chips=ol.lurk_chips())
    poseidon, witness = oracle_chip_callbacks(binding)
    z = zs.ZStore(real.hasher)
    zp = ms.intern_syntax(z, lr.read_lurk(lr.fib_program(n)))
    q = ol.QueryRecord(otop)
    i4 = otop.index["hash4"]
    for pre, dig in z.hashes.items():
        if len(pre) == 32:
            q.inv[i4][tuple(dig)] = tuple(pre)
    args = [0] * 24
    args[0] = zp.tag
    args[8:16] = zp.digest
    ol.execute(otop, "lurk_main", args, q, poseidon=poseidon)
    return oracle_machine(otop, q, witness, "lurk_main", q.public_values)


def real_mastermind():
    """The same table for BASELINE config 5's program (demo/mastermind.lurk folded into one expression, as
    tests/test_real_evaluator.py evaluates it): the oracle's interpreter recurses deeply on it, hence the thread with a big stack."""
    import threading

    out = {}

    def run():
        import lurk_reference as lr
        import measure_lurk_shape as ms
        from lurk_amd import zstore as zs
        from oracle import binding
        from oracle import lair as ol
        from test_lair_gpu import oracle_chip_callbacks

        binding.build()
        real = ms.RealLurk()
        otop = ol.Toplevel(real.source, chips=ol.lurk_chips())
        poseidon, witness = oracle_chip_callbacks(binding)
        z = zs.ZStore(real.hasher)
        zp = ms.intern_syntax(z, lr.read_lurk(lr.fold_repl_script(lr.demo_script("mastermind.lurk"))))
        q = ol.QueryRecord(otop)
        for name, ln in (("hash3", 24), ("hash4", 32), ("hash5", 40)):
            i = otop.index[name]
            for pre, dig in z.hashes.items():
                if len(pre) == ln:
                    q.inv[i][tuple(dig)] = tuple(pre)
        args = [0] * 24
        args[0] = zp.tag
        args[8:16] = zp.digest
        res = ol.execute(otop, "lurk_main", args, q, poseidon=poseidon)
        t_digest = [int(x) for x in real.resolver.digest[("lurk", "t")]]
        assert list(res) == [lr.enums()["Tag"]["Sym"]] + [0] * 7 + t_digest, "the script's assertions do not hold on the oracle's interpreter"
        out.update(oracle_machine(otop, q, witness, "lurk_main", q.public_values))

    sys.setrecursionlimit(1000000)
    threading.stack_size(512 * 1024 * 1024)
    th = threading.Thread(target=run)
    th.start()
    th.join()
    threading.stack_size(0)
    assert out, "the mastermind run failed"
    return out


def stand_in(eval_rows, workload="fib"):
    from lurk_amd.programs import lurk_mix as lm
    from oracle import binding
    from oracle import lair as ol
    from test_lair_gpu import oracle_chip_callbacks

    binding.build()
    mix = lm.fib_mix(eval_rows) if workload == "fib" else lm.lurk_mix(eval_rows)
    otop = ol.Toplevel(mix.source, chips=ol.lurk_chips())
    poseidon, witness = oracle_chip_callbacks(binding)
    q = ol.QueryRecord(otop)
    ol.execute(otop, mix.entry, list(mix.main_args), q, poseidon=poseidon)
    return oracle_machine(otop, q, witness, mix.entry, q.public_values)


def weighted(table, heights=None):
    """fraction of permutation-trace cells (padded height x columns) that lie in dead columns, over the growing chips"""
    def pad(r):
        p = 4
        while p < r:
            p *= 2
        return p
    tot = dead = 0
    for name, (s, rows) in table.items():
        h = pad(heights[name] if heights else rows)
        tot += h * (s["columns"] + 1)  # + the running-sum column, never dead
        dead += h * s["dead_columns"]
    return dead / tot if tot else 0.0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 24
    real = real_fib(n)
    # the stand-in at the number of eval rows the real run has
    mix = stand_in(real["eval"][1])
    print("%-26s %6s | %5s %5s %5s %5s | %5s %5s %5s %5s" % ("chip", "rows", "inter", "live", "cols", "dead", "inter", "live", "cols", "dead"))
    for name in real:
        r, rows = real[name]
        m = mix.get(name, ({"interactions": 0, "live_interactions": 0, "columns": 0, "dead_columns": 0}, 0))[0]
        print("%-26s %6d | %5d %5d %5d %5d | %5d %5d %5d %5d" % (name, rows, r["interactions"], r["live_interactions"], r["columns"], r["dead_columns"],
                                                                m["interactions"], m["live_interactions"], m["columns"], m["dead_columns"]))
    # weight by the heights of a 2^20-row fib shard (the per-level ratios of fib_shape.json)
    with open(os.path.join(ROOT, "tests", "golden", "fib_shape.json")) as f:
        shape = json.load(f)
    per_level = shape["fib_per_level"]
    levels = (1 << 20) / per_level["eval"]
    heights = {c: int(per_level[c] * levels) for c in real if c in per_level}
    grow = {c: v for c, v in real.items() if c in heights}
    grow_mix = {c: v for c, v in mix.items() if c in heights}
    fr, fm = weighted(grow, heights), weighted(grow_mix, heights)
    print("dead share of the permutation-trace cells of the growing chips at 2^20 eval rows: real %.3f, fib-mix %.3f" % (fr, fm))
    mm = real_mastermind() if "--mastermind" in sys.argv or "--write" in sys.argv else None
    if mm:
        mmix = stand_in(mm["eval"][1] // 8, "lurk")
        print("mastermind (real | lurk-mix):")
        for name in mm:
            r, rows = mm[name]
            m = mmix.get(name, ({"interactions": 0, "live_interactions": 0, "columns": 0, "dead_columns": 0}, 0))[0]
            print("%-26s %6d | %5d %5d %5d %5d | %5d %5d %5d %5d" % (name, rows, r["interactions"], r["live_interactions"], r["columns"], r["dead_columns"],
                                                                    m["interactions"], m["live_interactions"], m["columns"], m["dead_columns"]))
    if "--write" in sys.argv:
        shape["lookup_sparsity_mastermind"] = {
            "_about": "the same per-chip table for demo/mastermind.lurk (BASELINE config 5) on the oracle's traces: what lurk-mix is dialled to",
            "real": {c: s for c, (s, _) in mm.items()}}
        shape["lookup_sparsity"] = {
            "_about": "tools/measure_lookup_sparsity.py on (fib %d), the reference's functions on the oracle's traces: per chip, interactions / interactions real on some row / permutation columns (batches of two) / columns with no real interaction / return selectors / selectors some row takes; the stand-ins are dialled to live_interactions and live_selectors (lurk_amd/programs/lurk_mix.py)" % n,
            "fib_n": n,
            "real": {c: s for c, (s, _) in real.items()},
            "dead_cell_share_at_2^20": round(fr, 4),
        }
        with open(os.path.join(ROOT, "tests", "golden", "fib_shape.json"), "w") as f:
            json.dump(shape, f, indent=1, sort_keys=False)
            f.write("\n")


if __name__ == "__main__":
    main()
