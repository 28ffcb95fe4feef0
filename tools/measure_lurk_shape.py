#!/usr/bin/env python3
"""Measure the shape of the REAL Lurk machine in the build container and write tests/golden/fib_shape.json (numbers only).

What runs: the reference's own 39 Lair functions, read from /root/reference at run time by tools/lurk_reference.py (never
stored), compiled by the PRODUCT's host compiler (lurk_amd/csrc/lair/compile.cpp) with the native chips, and the benchmark's
program (/root/reference/benches/fib.rs:36-44) interned through lurk_amd.zstore (hashing on the product's host interpreter)
and executed by the product's interpreter (lurk_amd/csrc/lair/execute.cpp) exactly as benches/fib.rs:46-67 sets it up
(hash4 inverse queries injected, 24-lane input `[tag, 0*7, digest, 0*8]`).

What is checked on the way (the script fails when one does not hold):
  * all 39 literals of `test_widths` (/root/reference/src/core/eval_direct.rs:2025-2063) on the real function bodies;
  * `(fib N)` evaluates to U64 fib(N) mod 2^64 (tag lane + 8 little-endian bytes) for every N measured.

  * demo/mastermind.lurk (BASELINE config 5), its REPL commands folded into one expression, evaluates to `t`: all 41 assertions
    of the script hold under the evaluator.

What is written: per chip the layout and the AIR's size (columns, selectors, interactions, constraints, the product's
program lengths) and per N the rows of every chip, memory-table sizes and byte records -- the numbers SURVEY.md appendix C
estimated by hand.  `lurk_amd/programs/lurk_mix.py` dials its stand-in machine from this file.

CPU only; no GPU, no oracle.  Skips (exit 0 with a message) when /root/reference is absent.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import lurk_reference as lr  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "fib_shape.json")
HASH_LEAVES = (
    "invertible fn hash3(preimg: [24]): [8] {\n    let img: [8] = extern_call(hasher3, preimg);\n    return img\n}\n"
    "invertible fn hash4(preimg: [32]): [8] {\n    let img: [8] = extern_call(hasher4, preimg);\n    return img\n}\n"
    "invertible fn hash5(preimg: [40]): [8] {\n    let img: [8] = extern_call(hasher5, preimg);\n    return img\n}\n"
)


class HostHasher:
    """Poseidon2 hash3/4/5 on the product's host interpreter (the native chips behind `extern_call`)."""

    def __init__(self):
        from lurk_amd import lair

        self.top = lair.Toplevel(HASH_LEAVES, lurk_chips=True)
        self.q = lair.QueryRecord(self.top)

    def hash(self, pre):
        return self.top.execute_by_name({24: "hash3", 32: "hash4", 40: "hash5"}[len(pre)], pre, self.q)


def intern_syntax(z, syn):
    """lurk_amd.zstore syntax tuples -> ZPtr through the sequential store (zstore.rs:513-549)."""
    k = syn[0]
    if k == "u64":
        return z.u64(syn[1])
    if k == "num":
        return z.num(syn[1])
    if k == "char":
        return z.char(syn[1])
    if k == "bignum":
        return z.big_num(syn[1])
    if k == "comm":
        return z.comm(syn[1])
    if k == "str":
        return z.intern_string(syn[1])
    if k == "sym":
        return z.intern_symbol(list(syn[1]), builtin=syn[2] == "builtin", keyword=syn[2] == "keyword")
    if k == "list":
        return z.intern_list([intern_syntax(z, x) for x in syn[1]])
    if k == "improper":
        return z.intern_list([intern_syntax(z, x) for x in syn[1]], tail=intern_syntax(z, syn[2]))
    if k == "quote":
        return z.intern_list([z.builtin_sym("quote"), intern_syntax(z, syn[1])])
    raise ValueError(k)


class RealLurk:
    """The native Lurk toplevel built from the reference's sources (eval_direct.rs:81-118), on the product's host side."""

    def __init__(self):
        from lurk_amd import lair
        from lurk_amd import zstore as zs

        self.lair, self.zs = lair, zs
        self.hasher = HostHasher()
        z = zs.ZStore(self.hasher)

        def digest(kind, name):
            return (z.intern_symbol([zs.LURK_PACKAGE, name]) if kind == "lurk" else z.builtin_sym(name)).digest

        self.resolver = lr.Resolver(digest)
        self.funcs = self.resolver.functions()
        self.names = list(self.funcs)
        self.source = "\n".join(self.funcs.values())
        self.top = lair.Toplevel(self.source, lurk_chips=True)

    def widths(self):
        return {n: self.top.func_info(self.top.func_index(n))["layout"].total() for n in self.names}

    def chip_shapes(self):
        from lurk_amd.air import ChipAir

        out = {}
        for n in self.names:
            i = self.top.func_index(n)
            info = self.top.func_info(i)
            lay = info["layout"]
            a = ChipAir.for_func(self.top, i)
            sizes = a.interaction_sizes()
            hist = {}
            for s in sizes:
                hist[str(s)] = hist.get(str(s), 0) + 1
            out[n] = {
                "width": lay.total(), "input": lay.input, "output": lay.output, "aux": lay.aux, "sel": lay.sel,
                "partial": int(info["partial"]), "invertible": int(info["invertible"]),
                "sends": a.num_sends, "receives": a.num_receives, "constraints": a.num_constraints,
                "max_constraint_degree": a.max_constraint_degree, "log_quotient_degree": a.log_quotient_degree,
                "permutation_width": a.permutation_width, "interaction_tuple_words": sum(sizes),
                "interaction_size_histogram": dict(sorted(hist.items(), key=lambda kv: int(kv[0]))),
                # lengths of the product's register programs for this chip (air_program.h): what its quotient / permutation-row
                # kernels execute per row
                "constraint_instrs": a.constraint_instrs, "interaction_instrs": a.interaction_instrs,
            }
        return out

    def run(self, text: str):
        """Execute one Lurk expression in the empty environment: (output lanes, QueryRecord, seconds)."""
        z = self.zs.ZStore(self.hasher)
        zp = intern_syntax(z, lr.read_lurk(text))
        q = self.lair.QueryRecord(self.top)
        i4 = self.top.func_index("hash4")
        for pre, dig in z.hashes.items():  # benches/fib.rs:56: every hash4 the store knows
            if len(pre) == 32:
                q.inject_inv_query(i4, list(pre), list(dig))
        args = [0] * 24
        args[0] = zp.tag
        args[8:16] = zp.digest
        t = time.time()
        out = self.top.execute_by_name("lurk_main", args, q)
        return out, q, time.time() - t

    def run_zptr(self, z, zp, env=None):
        """`run_tests` of the reference's evaluator corpus (src/core/tests/mod.rs:28-56): the expression `zp` under the environment
        `env` (a ZPtr; None: the empty one), every hash3 / hash4 / hash5 the store `z` knows injected as an inverse query;
        returns (the 16 output lanes, QueryRecord)."""
        q = self.lair.QueryRecord(self.top)
        idx = {24: self.top.func_index("hash3"), 32: self.top.func_index("hash4"), 40: self.top.func_index("hash5")}
        for pre, dig in z.hashes.items():
            q.inject_inv_query(idx[len(pre)], list(pre), list(dig))
        args = zp.flatten() + (list(env.digest) if env is not None else [0] * 8)
        return self.top.execute_by_name("lurk_main", args, q), q

    def record_counts(self, q):
        rows = {n: q.num_func_queries(self.top.func_index(n)) for n in self.names}
        mem = {str(l): q.num_mem_queries(l) for l in self.lair.MEM_TABLE_SIZES}
        return rows, mem, q.num_byte_records()


def fib_mod64(n: int) -> int:
    a, b = 0, 1
    for _ in range(n):
        a, b = b, (a + b) & 0xFFFFFFFFFFFFFFFF
    return a


def measure(ns):
    real = RealLurk()
    want = lr.test_widths()
    got = real.widths()
    bad = {n: (got[n], want[n]) for n in want if got.get(n) != want[n]}
    if bad:
        raise SystemExit(f"width mismatch on the real function bodies (got, reference literal): {bad}")
    print(f"test_widths: {len(want)}/{len(want)} literals reproduced on the real function bodies (product compiler)")
    chips = real.chip_shapes()
    mem_w = {str(l): 4 + l for l in real.lair.MEM_TABLE_SIZES}
    tag_u64 = lr.enums()["Tag"]["U64"]
    runs = {}
    for n in ns:
        out, q, dt = real.run(lr.fib_program(n))
        v = fib_mod64(n)
        expect = [tag_u64] + [0] * 7 + [(v >> (8 * i)) & 0xFF for i in range(8)]
        if list(out) != expect:
            raise SystemExit(f"(fib {n}) evaluated to {out}, expected {expect}")
        rows, mem, nbytes = real.record_counts(q)
        pv = q.expect_public_values()
        e = rows["eval"]
        main_cols = sum(chips[c]["width"] * r for c, r in rows.items()) + sum(mem_w[l] * r for l, r in mem.items())
        perm_cols = sum(4 * chips[c]["permutation_width"] * r for c, r in rows.items())
        runs[str(n)] = {
            "value_mod_2_64": v, "rows": {c: r for c, r in rows.items() if r}, "mem_rows": {l: r for l, r in mem.items() if r},
            "byte_records": nbytes, "depth": sum(b << (8 * i) for i, b in enumerate(pv[40:44])),
            "main_columns_per_eval_row": round(main_cols / e, 3), "func_permutation_columns_per_eval_row": round(perm_cols / e, 3),
            "host_execute_s": round(dt, 4),
        }
        print(f"(fib {n}) = {v} ok: eval rows {e}, main columns / eval row {main_cols / e:.1f}, {dt:.3f} s")
    # per fib level: slope between the two largest runs (the cost is linear in N: every (expr, env) pair is memoised)
    per_level = {}
    if len(ns) >= 2:
        a, b = sorted(ns)[-2:]
        ra, rb = runs[str(a)], runs[str(b)]
        for c in rb["rows"]:
            d = (rb["rows"][c] - ra["rows"].get(c, 0)) / (b - a)
            if d:
                per_level[c] = round(d, 6)
        for l in rb["mem_rows"]:
            d = (rb["mem_rows"][l] - ra["mem_rows"].get(l, 0)) / (b - a)
            if d:
                per_level["mem" + l] = round(d, 6)
        per_level["byte_records"] = round((rb["byte_records"] - ra["byte_records"]) / (b - a), 6)
    # BASELINE config 5: demo/mastermind.lurk, its REPL commands folded into one expression (lurk_reference.fold_repl_script); every
    # assertion of the script holds iff the fold evaluates to `t`
    out, q, dt = real.run(lr.fold_repl_script(lr.demo_script("mastermind.lurk")))
    t_digest = [int(x) for x in real.resolver.digest[("lurk", "t")]]
    if list(out) != [lr.enums()["Tag"]["Sym"]] + [0] * 7 + t_digest:
        raise SystemExit(f"demo/mastermind.lurk: an assertion of the script failed under the evaluator (result {out})")
    rows, mem, nbytes = real.record_counts(q)
    e = rows["eval"]
    mastermind = {
        "script": "demo/mastermind.lurk: 13 def + 24 defq/transition + 3 defrec folded to let / letrec, 41 assertions folded to `if` (all hold)",
        "rows": {c: r for c, r in rows.items() if r}, "mem_rows": {l: r for l, r in mem.items() if r}, "byte_records": nbytes,
        "main_columns_per_eval_row": round((sum(chips[c]["width"] * r for c, r in rows.items()) + sum(mem_w[l] * r for l, r in mem.items())) / e, 3),
        "func_permutation_columns_per_eval_row": round(sum(4 * chips[c]["permutation_width"] * r for c, r in rows.items()) / e, 3),
    }
    print(f"demo/mastermind.lurk: all assertions hold; eval rows {e}, {sum(1 for r in rows.values() if r)} function chips with rows")
    return {
        "_about": "Measured by tools/measure_lurk_shape.py on the reference's own Lair functions (read from /root/reference at run time, "
                  "compiled and executed by this repo's host code). Numbers only: no program text, no bytecode.",
        "sources": {"functions": "src/core/eval_direct.rs:119-1957, src/core/ingress.rs:99-325, src/core/misc.rs:5-121",
                    "program": "benches/fib.rs:36-44", "widths": "src/core/eval_direct.rs:2025-2063", "mastermind": "demo/mastermind.lurk"},
        "widths_reproduced": f"{len(want)}/{len(want)}",
        "func_order": real.names,
        "chips": chips,
        "fib": runs,
        "fib_per_level": per_level,
        "mastermind": mastermind,
    }


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--n", type=int, nargs="*", default=[30, 1000, 10000, 75000])
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--check", action="store_true", help="compare with the committed file instead of writing it")
    a = ap.parse_args()
    if not lr.available():
        print("measure_lurk_shape: /root/reference is absent: nothing to measure (skipped)")
        return 0
    shape = measure(a.n)
    if a.check:
        with open(a.out) as f:
            old = json.load(f)
        for r in list(shape["fib"].values()) + list(old["fib"].values()):
            r.pop("host_execute_s", None)
        if old != shape:
            raise SystemExit("tests/golden/fib_shape.json differs from a fresh measurement")
        print("fib_shape.json matches a fresh measurement")
        return 0
    with open(a.out, "w") as f:
        json.dump(shape, f, indent=1, sort_keys=False)
        f.write("\n")
    print("wrote", os.path.relpath(a.out, ROOT))
    return 0


if __name__ == "__main__":
    sys.exit(main())
