"""Shape-matched stand-ins for the Lurk evaluator's Lair machine: the `fib-mix` and `lurk-mix` workloads.

The reference's evaluator (/root/reference/src/core/eval_direct.rs: 39 Lair functions) is program text and cannot be
shipped here.  What the proving hot path sees of it is a *machine shape*, and since round 5 that shape is MEASURED: in the
build container tools/measure_lurk_shape.py runs the reference's own functions through this repo's compiler and
interpreter and writes tests/golden/fib_shape.json -- per chip: columns, selectors, lookups (sends / receives), constraints,
lookup-tuple words; per `(fib N)` run: rows of every chip, memory-table sizes.  This module generates Lair functions with
the reference's names, signatures and flags whose compiled chips reproduce those numbers:

  main-trace width, selector count, sends and receives   -- exactly (asserted when a machine is built);
  constraints and lookup-tuple words                     -- as closely as the construction allows (reported by
                                                            `Mix.fit`, held within a few per cent by tests/test_mix_programs.py);
  chip heights, memory-table heights                     -- the measured ratios of a fib run, per eval row.

How a stand-in is built.  Every non-leaf function is a *walker*: `F(j, phase, ...)` calls `F(j - 1, phase', ...)` until its
counter is zero, so starting it at n - 1 yields exactly n queries = n trace rows.  Like the real functions it is a `match`
with one return per selector: the first R branches are *live* -- branch k makes the recursive call with phase' = k + 1 mod R,
so consecutive rows take different branches -- the others are never taken by a run (the error and rare-form branches of
the real evaluator) and only exist in the AIR.  Auxiliary columns are shared between branches the way the layout rules
share them (func_chip.rs:146-175), lookups are not (every branch's `require` is its own interaction): the generator
places the measured number of lookups (stores of constant memory cells: 4 columns each, re-used cells add no memory
rows; byte-pair range checks where columns are short) over the branches within the measured column budget, pads one
branch with products to the exact width, then adds equality assertions to the never-taken branches up to the measured
constraint count and sizes the cells for the measured tuple words.  Leaf functions (hashers, u64 gadgets) are the thin
wrappers of /root/reference/src/core/misc.rs; a walker calls the leaves it is paired with on step-dependent arguments, the
u64 gadgets only in branch 0 (every R-th row), which is what makes their chips 1/R as tall.

`lurk_main` (partial, 24 inputs, 16 outputs: the 44-lane public-value layout of /root/reference/src/core/stark_machine.rs:16-17)
starts the first walker; each walker's bottom frame starts the next.

`fib_mix(eval_rows)`: the 17 function chips a real `(fib N)` touches, at the measured per-level ratios (per 10 eval rows:
5 eval_builtin_expr, 4 eval_binop_num, 4 apply, 2 env_lookup, 1 eval_begin, 1 each u64_add / u64_sub / u64_lessthan, 2 rows
of the 5-wide and 2 of the 8-wide memory table) and the N-independent chips (ingress 41 rows, hash4, preallocate_symbols, the
letrec helpers, egress) at their measured sizes.  `lurk_mix(eval_rows)`: all 39 functions + the 6 memory tables + byte table +
entrypoint (BASELINE config 5, `demo/mastermind.lurk`: irregular widths 9 ... 815), every chip at its measured shape, heights
in the ratios the real evaluator gives them on the mastermind script (env_lookup 0.88, eval_builtin_expr 0.45, apply 0.35 per
eval row, ...; the 13 functions the script never calls at a token height).
"""
from __future__ import annotations

import functools
import json
import os
import re
from dataclasses import dataclass, field as dfield

# name -> (partial, invertible, input sizes, output size, width); /root/reference/src/core/eval_direct.rs:121-1957,2025-2063,
# /root/reference/src/core/ingress.rs:101,144,241, /root/reference/src/core/misc.rs:7-121
LURK_FUNCS = {
    "lurk_main": (True, False, (8, 8, 8), 16, 97),
    "preallocate_symbols": (False, False, (), 0, 188),
    "eval_coroutine_expr": (False, False, (1, 1, 1, 1), 2, 10),  # the native toplevel's stub (eval_direct.rs:142), not the coroutine one
    "eval": (True, False, (1, 1, 1), 2, 78),
    "eval_builtin_expr": (True, False, (1, 1, 1, 1), 2, 148),
    "eval_bind_builtin": (True, False, (1, 1, 1), 2, 110),
    "eval_env_builtin": (True, False, (1, 1, 1), 2, 81),
    "eval_apply_builtin": (True, False, (1, 1, 1, 1, 1), 2, 79),
    "eval_opening_unop": (True, False, (1, 1, 1, 1), 2, 97),
    "eval_hide": (True, False, (1, 1, 1), 2, 115),
    "eval_unop": (True, False, (1, 1, 1, 1), 2, 78),
    "eval_binop_num": (True, False, (1, 1, 1, 1, 1, 1), 2, 107),
    "eval_binop_misc": (True, False, (1, 1, 1, 1, 1, 1), 2, 70),
    "eval_begin": (True, False, (1, 1, 1), 2, 68),
    "eval_list": (True, False, (1, 1, 1), 2, 72),
    "eval_let": (True, False, (1, 1, 1, 1, 1), 2, 94),
    "eval_letrec": (True, False, (1, 1, 1, 1, 1), 2, 66),
    "extend_env_with_mutuals": (False, False, (1, 1, 1, 1), 2, 54),
    "eval_letrec_bindings": (True, False, (1, 1), 2, 66),
    "coerce_if_sym": (False, False, (1,), 1, 9),
    "open_comm": (False, False, (1,), 2, 50),
    "equal": (True, False, (1, 1, 1, 1), 2, 86),
    "equal_inner": (False, False, (1, 1, 1, 1), 1, 58),
    "car_cdr": (True, False, (1, 1, 1), 4, 61),
    "apply": (True, False, (1, 1, 1, 1, 1), 2, 114),
    "env_lookup": (False, False, (9, 1), 2, 52),
    "ingress": (False, False, (8, 8), 2, 104),
    "egress": (False, False, (1, 1), 9, 81),
    "hash3": (False, True, (24,), 8, 493),
    "hash4": (False, True, (32,), 8, 655),
    "hash5": (False, True, (40,), 8, 815),
    "u64_add": (False, False, (1, 1), 1, 53),
    "u64_sub": (False, False, (1, 1), 1, 53),
    "u64_mul": (False, False, (1, 1), 1, 85),
    "u64_divrem": (False, False, (1, 1), 2, 166),
    "u64_lessthan": (False, False, (1, 1), 1, 44),
    "u64_iszero": (False, False, (1,), 1, 26),
    "digest_equal": (False, False, (1, 1), 1, 38),
    "big_num_lessthan": (False, False, (1, 1), 1, 78),
}
LURK_FUNC_ORDER = list(LURK_FUNCS)  # the order of `test_widths`

# Leaf functions: thin wrappers around the native chips, shaped by the layout rules of SURVEY.md appendix B
# (u64_add = 1 + 2 + 1 + aux[2 + (8 + 3) + (8 + 3) + (8 + 4 * 3) + 4] + 1 = 53, ...).
LEAVES = {
    "hash3": "invertible fn hash3(preimg: [24]): [8] {\n    let img: [8] = extern_call(hasher3, preimg);\n    return img\n}\n",
    "hash4": "invertible fn hash4(preimg: [32]): [8] {\n    let img: [8] = extern_call(hasher4, preimg);\n    return img\n}\n",
    "hash5": "invertible fn hash5(preimg: [40]): [8] {\n    let img: [8] = extern_call(hasher5, preimg);\n    return img\n}\n",
    "u64_add": "fn u64_add(a, b): [1] {\n    let x: [8] = load(a);\n    let y: [8] = load(b);\n    let z: [8] = extern_call(u64_add, x, y);\n    let p = store(z);\n    return p\n}\n",
    "u64_sub": "fn u64_sub(a, b): [1] {\n    let x: [8] = load(a);\n    let y: [8] = load(b);\n    let z: [8] = extern_call(u64_sub, x, y);\n    let p = store(z);\n    return p\n}\n",
    "u64_mul": "fn u64_mul(a, b): [1] {\n    let x: [8] = load(a);\n    let y: [8] = load(b);\n    let z: [8] = extern_call(u64_mul, x, y);\n    let p = store(z);\n    return p\n}\n",
    "u64_divrem": "fn u64_divrem(a, b): [2] {\n    let x: [8] = load(a);\n    let y: [8] = load(b);\n    let (q: [8], r: [8]) = extern_call(u64_divrem, x, y);\n    let pq = store(q);\n    let pr = store(r);\n    return (pq, pr)\n}\n",
    "u64_lessthan": "fn u64_lessthan(a, b): [1] {\n    let x: [8] = load(a);\n    let y: [8] = load(b);\n    let lt = extern_call(u64_lessthan, x, y);\n    return lt\n}\n",
    "u64_iszero": "fn u64_iszero(a): [1] {\n    let x: [8] = load(a);\n    let z = extern_call(u64_iszero, x);\n    return z\n}\n",
    "digest_equal": "fn digest_equal(a, b): [1] {\n    let x: [8] = load(a);\n    let y: [8] = load(b);\n    let d = sub(x, y);\n    let one = 1;\n    if !d {\n        return one\n    }\n    let zero = 0;\n    return zero\n}\n",
    "big_num_lessthan": "fn big_num_lessthan(a, b): [1] {\n    let x: [8] = load(a);\n    let y: [8] = load(b);\n    let lt = extern_call(big_num_lessthan, x, y);\n    return lt\n}\n",
}

SHAPE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "fib_shape.json")
MEM_LENS = (2, 3, 4, 5, 6, 8)  # /root/reference/src/lair/execute.rs:243-244

# functions that are not walkers: fixed text (LEAVES, eval_coroutine_expr) or a shaped body without recursion
SHAPED_LEAVES = ("preallocate_symbols", "coerce_if_sym", "open_comm", "equal_inner")
FIXED = dict(LEAVES)
FIXED["eval_coroutine_expr"] = "fn eval_coroutine_expr(a0, a1, a2, a3): [2] {\n    let zero = 0;\n    return (zero, zero)\n}\n"

# walker -> leaf calls made on every step, before the match.  `@j` is the step counter (distinct per row), `@arr` the array
# parameter the counter is lane 0 of.
LEAF_CALLS = {
    # the hashers: hash4 under ingress (the real ingress reads conses and strings in through it: 1261 hash4 rows for 1378 ingress rows
    # on mastermind), hash3 / hash5 under egress (a handful of rows: commitments opened, closures hashed out)
    "hash4": ("ingress", ["let pz4 = [0; 31];", "let pre4: [32] = (@j, pz4);", "let h4: [8] = call(hash4, pre4);"]),
    "hash3": ("egress", ["let pz3 = [0; 23];", "let pre3: [24] = (@j, pz3);", "let h3: [8] = call(hash3, pre3);"]),
    "hash5": ("egress", ["let pz5 = [0; 39];", "let pre5: [40] = (@j, pz5);", "let h5: [8] = call(hash5, pre5);"]),
    "coerce_if_sym": ("eval_unop", ["let cs = call(coerce_if_sym, @j);"]),
    "open_comm": ("eval_opening_unop", ["let (oc0, oc1) = call(open_comm, @j);"]),
    "equal_inner": ("equal", ["let ei = call(equal_inner, @j, a1, a2, a3);"]),
    "eval_coroutine_expr": ("eval_builtin_expr", ["let (co0, co1) = call(eval_coroutine_expr, @j, a1, a2, a3);"]),
}
# the u64 gadgets are called from branch 0 of their owner (every R-th row) on the accumulator pointer `@acc` the owner threads
# through its recursion (U64_OWNER_PARAM): u64_add advances it, so every call is a new query and stores a new 8-lane cell
U64_OWNER_PARAM = {"eval_binop_num": "a4", "apply": "a3"}
U64_CALLS = {
    "u64_add": "let acc2 = call(u64_add, @acc, c8);",
    "u64_sub": "let us = call(u64_sub, @acc, c8s);",
    "u64_mul": "let um = call(u64_mul, @acc, @acc);",
    "u64_divrem": "let (uq, ur) = call(u64_divrem, @acc, c8);",
    "u64_lessthan": "let ul = call(u64_lessthan, @acc, c8);",
    "u64_iszero": "let uz = call(u64_iszero, @acc);",
    "digest_equal": "let de = call(digest_equal, @acc, c8);",
    "big_num_lessthan": "let bl = call(big_num_lessthan, @acc, c8);",
}
C8 = ["let b1 = 3;", "let b2 = 1;", "let c8 = store(b1, b2, zero, zero, zero, zero, zero, zero);"]  # the u64 constant 259 (one shared cell)
# u64_sub takes another constant: the accumulator runs over the multiples of 259, so acc - 130 is a cell nobody else stores
# (with 259 the difference would be the previous accumulator and the store would find its cell)
C8S = ["let b3 = 130;", "let c8s = store(b3, zero, zero, zero, zero, zero, zero, zero);"]
# walkers that allocate new memory cells: walker -> (cell length, every k-th row).  fib: an env frame (5 lanes, ingress.rs:216-219) per
# env_lookup row = the measured 2 rows of the 5-wide table per fib level; mastermind: 0.47 frames and 0.19 cons cells per eval row
FIB_FRESH = {"env_lookup": (5, 1)}
LURK_FRESH = {"eval_builtin_expr": (5, 1), "apply": (4, 2)}  # (env_lookup's 4 lookups leave no room for a store in one of two live branches)


@functools.lru_cache(maxsize=1)
def load_shape() -> dict:
    """tests/golden/fib_shape.json: the measured machine (tools/measure_lurk_shape.py)."""
    with open(SHAPE_PATH) as f:
        return json.load(f)


def _sig(name):
    sizes = LURK_FUNCS[name][2]
    return [(f"a{i}", s) for i, s in enumerate(sizes)]


def _counter(name):
    """(counter variable, array parameter it is lane 0 of or None)."""
    for p, s in _sig(name):
        if s == 1 and p != U64_OWNER_PARAM.get(name):
            return p, None
    arr, _ = _sig(name)[-1]
    return "j", arr


def _phase(name):
    """(phase variable, array parameter it is a lane of or None): the second scalar parameter that is neither the counter nor the
    u64 accumulator; without one, lane 1 of the counter's array, else lane 0 of the first array parameter."""
    j, arr = _counter(name)
    for p, s in _sig(name):
        if s == 1 and p != j and p != U64_OWNER_PARAM.get(name):
            return p, None
    if arr is not None:
        return "ph", arr
    for p, s in _sig(name):
        if s > 1:
            return "ph", p
    raise AssertionError(f"{name}: no parameter to carry the phase")


def _lanes(name, p, s, first, second):
    """Names of the lanes of array parameter `p` in a destructuring / rebuilding pattern: the counter (when `p` is its array) and the
    phase (when `p` carries it) come first."""
    j, arr = _counter(name)
    ph, pharr = _phase(name)
    head = ([first] if p == arr else []) + ([second] if p == pharr else [])
    return head + [f"{p}_{k}" for k in range(len(head), s)]


def _head(name):
    partial, invertible, _, out, _ = LURK_FUNCS[name]
    sig = ", ".join(p if s == 1 else f"{p}: [{s}]" for p, s in _sig(name))
    return ("partial " if partial else "") + ("invertible " if invertible else "") + f"fn {name}({sig}): [{out}] {{\n"


def _ret(out, pool):
    if out == 0:
        return "return ()"
    vals = [pool[k % len(pool)] for k in range(out)]
    return f"return ({', '.join(vals)})" if out != 1 else f"return {vals[0]}"


@dataclass
class Branch:
    """Filler of one `match` branch: constant cells stored (their lengths), loads of the first of them, byte-pair range checks,
    products, assertions."""
    live: bool = False
    cells: list = dfield(default_factory=list)
    loads: int = 0
    ranges: int = 0
    muls: int = 0
    asserts: int = 0
    pinned: bool = False   # cell 0's length is part of the width fit (the loads read it): the tuple-word pass leaves it alone


def _filler(b: Branch, tag: int, seed: str, scalars: list, taken: bool = False, fact=None):
    """Lines of one branch's filler; returns (lines, last product or None).  `taken`: a run executes this branch, so its
    assertions must hold: the phase the branch was selected by (`fact` = (variable, constant)) and pointers of cells stored
    twice (cell 2i + 1 then repeats cell 2i); a never-taken branch asserts equalities between whatever is in scope."""
    L = []
    ptrs = []
    twins = taken and b.asserts > 0
    lens = list(b.cells)
    for c, ln in enumerate(lens):
        src = c - 1 if twins and c % 2 == 1 else c
        ln = lens[src]
        names = [f"k{c}_{i}" for i in range(ln)]
        L += [f"let {nm} = {100 + 16 * tag + 10 * src + i};" for i, nm in enumerate(names)]
        L.append(f"let fp{c} = store({', '.join(names)});")
        ptrs.append(f"fp{c}")
    for i in range(b.loads):
        L.append(f"let ({', '.join(f'l{i}_{k}' for k in range(lens[0]))}) = load(fp0);")
    for _ in range(b.ranges):
        L.append("range_u8!(zero, one);")
    prev = None
    for m in range(b.muls):
        L.append(f"let m{m} = mul({prev or seed}, {seed});")
        prev = f"m{m}"
    if b.asserts:
        if taken:
            pairs = []
            if fact:
                L.append(f"let cfact = {fact[1]};")
                pairs.append((fact[0], "cfact"))
            pairs += [(ptrs[i], ptrs[i + 1]) for i in range(0, len(ptrs) - 1, 2)]
        else:
            pool = list(dict.fromkeys(scalars + ptrs + ([prev] if prev else [])))
            pairs = [(u, v) for i, u in enumerate(pool) for v in pool[i + 1:]] or [(pool[0], "one")]
        for a in range(b.asserts if pairs else 0):
            u, v = pairs[a % len(pairs)]
            L.append(f"assert_eq!({u}, {v});")
    return L, prev


@dataclass
class Spec:
    """What the generator dials per function (found by `_fit`)."""
    branches: list           # [Branch]; walkers: R live ones first
    live: int = 1            # R
    next_walker: str | None = None


def emit_walker(name, spec: Spec, shared_pre, phase_pre, base):
    _, _, _, out, _ = LURK_FUNCS[name]
    sig = _sig(name)
    j, arr = _counter(name)
    ph, pharr = _phase(name)
    acc = U64_OWNER_PARAM.get(name)
    R = spec.live
    sub = lambda t: t.replace("@j", j).replace("@acc", acc or "zero").replace("@arr", arr or "zero")
    L = ["let zero = 0;", "let one = 1;"]
    for p, s in sig:
        if p == arr or p == pharr:
            L.append(f"let ({', '.join(_lanes(name, p, s, 'j', 'ph'))}) = {p};")
    L.append(f"if !{j} {{")
    L += ["    " + b for b in base]
    L.append("    " + _ret(out, ["zero"]))
    L.append("}")
    L.append(f"let jn = sub({j}, one);")
    L += [sub(x) for x in shared_pre]
    scalars = [p for p, s in sig if s == 1] + ["jn"]
    L.append(f"match {ph} {{")
    for k, b in enumerate(spec.branches):
        B = [f"{k} => {{"]
        rets = None
        if b.live:
            B.append(f"    let np = {(k + 1) % R};")
            pre = [sub(x) for x in phase_pre.get(k, [])]
            B += ["    " + x for x in pre]
            args = []
            for p, s in sig:
                if p == j:
                    args.append("jn")
                elif p == arr or p == pharr:
                    B.append(f"    let n{p}: [{s}] = ({', '.join(_lanes(name, p, s, 'jn', 'np'))});")
                    args.append(f"n{p}")
                elif p == ph:
                    args.append("np")
                elif p == acc and any("acc2" in x for x in pre):
                    args.append("acc2")
                else:
                    args.append(p)
            rets = [f"r{i}" for i in range(out)]
            B.append(f"    let {'(' + ', '.join(rets) + ')' if out != 1 else rets[0]} = call({name}, {', '.join(args)});")
        fl, prev = _filler(b, k, j, scalars, taken=b.live, fact=(ph, k))
        B += ["    " + x for x in fl]
        pool = (rets or []) + ([prev] if prev else []) + ([j] if b.live else scalars)
        B.append("    " + _ret(out, pool))
        B.append("}")
        L += ["    " + x for x in B]
    L.append("}")
    return _head(name) + "".join("    " + x + "\n" for x in L) + "}\n"


def emit_leaf(name, spec: Spec):
    """A function without recursion: one branch = straight-line; several = a `match` on the first parameter whose default is the
    branch every call takes (callers pass step counters)."""
    _, _, _, out, _ = LURK_FUNCS[name]
    sig = _sig(name)
    scalars = [p for p, s in sig if s == 1]
    seed = scalars[0] if scalars else None
    L = ["let zero = 0;", "let one = 1;"]
    bs = spec.branches
    if len(bs) == 1:
        fl, prev = _filler(bs[0], 0, seed or "one", scalars or ["one"], taken=True)
        L += fl
        L.append(_ret(out, ([prev] if prev else []) + ([seed] if seed else []) + ["one"]))
    else:
        L.append(f"match {seed} {{")
        for k, b in enumerate(bs[1:]):
            fl, prev = _filler(b, k + 1, seed, scalars)
            blk = [f"{P_MINUS(k)} => {{"] + ["    " + x for x in fl] + ["    " + _ret(out, scalars + ["one"]), "}"]
            L += ["    " + x for x in blk]
        L.append("};")
        fl, prev = _filler(bs[0], 0, seed, scalars, taken=True)
        L += fl
        L.append(_ret(out, ([prev] if prev else []) + [seed, "one"]))
    return _head(name) + "".join("    " + x + "\n" for x in L) + "}\n"


def P_MINUS(k):
    """Case constants no step counter reaches: p - 1 - k."""
    return 2013265920 - k


def emit_main(first_args_lines, first_call, has_prealloc, spec: Spec):
    L = ["let zero = 0;", "let one = 1;"]
    if has_prealloc:
        L.append("let () = call(preallocate_symbols, );")
    L += first_args_lines
    L.append(first_call)
    L.append("let (t0, t1, t2, t3, t4, t5, t6, t7) = a0;")
    fl, prev = _filler(spec.branches[0], 0, "t0", ["t0", "t1", "t2"], taken=True)
    L += fl
    pool = ["r0", "r1"] + ([prev] if prev else []) + ["t0", "t1", "one"]
    L.append(_ret(16, pool))
    return _head("lurk_main") + "".join("    " + x + "\n" for x in L) + "}\n"


def _start_call(name, count, tag):
    """Lines that start walker `name` with `count` steps from inside another function: constants for its arguments (phase 0), the
    shared u64 cell for its accumulator."""
    L = [f"let n_{tag} = {count};"]
    args = []
    j, arr = _counter(name)
    acc = U64_OWNER_PARAM.get(name)
    for p, s in _sig(name):
        if p == j:
            args.append(f"n_{tag}")
        elif p == arr:
            L.append(f"let s_{tag}: [{s}] = ({', '.join([f'n_{tag}'] + ['zero'] * (s - 1))});")
            args.append(f"s_{tag}")
        elif p == acc:
            L += [x.replace("b1", f"b1_{tag}").replace("b2", f"b2_{tag}").replace("c8", f"c8_{tag}") for x in C8]
            args.append(f"c8_{tag}")
        elif s == 1:
            args.append("zero")
        else:
            L.append(f"let z_{tag}_{p}: [{s}] = ({', '.join(['zero'] * s)});")
            args.append(f"z_{tag}_{p}")
    out = LURK_FUNCS[name][3]
    rets = [f"r{k}" for k in range(out)]
    L.append(f"let {'(' + ', '.join(rets) + ')' if out != 1 else rets[0]} = call({name}, {', '.join(args)});")
    return L


@dataclass
class Mix:
    name: str
    source: str
    entry: str
    eval_rows: int
    rows: dict          # function name -> number of queries (= trace rows) the run produces
    main_args: list
    fit: dict = dfield(default_factory=dict)   # function name -> {"got": {...}, "target": {...}} for the shaped functions


def _measure(src_all, fname):
    from .. import lair
    from ..air import ChipAir

    top = lair.Toplevel(src_all, lurk_chips=True)
    i = top.func_index(fname)
    lay = top.func_info(i)["layout"]
    a = ChipAir.for_func(top, i)
    return {"width": lay.total(), "aux": lay.aux, "sel": lay.sel, "sends": a.num_sends, "receives": a.num_receives, "constraints": a.num_constraints,
            "interaction_tuple_words": sum(a.interaction_sizes())}


def _stub(f):
    _, _, _, out, _ = LURK_FUNCS[f]
    return _head(f) + "    let zero = 0;\n    " + _ret(out, ["zero"]) + "\n}\n"


def _fit_once(fname, target, emit, n_branches, live, others_src, fixed_live=False, can_mul=True, pad_first=True):
    """Finds the Spec under which `emit(spec)` compiles to the target's width, selectors and lookups exactly, and as close to
    its constraints and tuple words as the filler allows.  `n_branches` match branches, the first `live` of them live."""
    def build(spec):
        return _measure(others_src + emit(spec), fname)

    def fresh(live_n):
        return Spec([Branch(live=k < live_n) for k in range(n_branches)], live=max(live_n, 1))

    # 1. lookups: with no filler the structure alone has I0 <= I of them (fewer live branches when it has not)
    while True:
        spec = fresh(live)
        got = build(spec)
        assert got["sel"] == target["sel"], (fname, got["sel"], target["sel"])
        need = target["sends"] - got["sends"]
        if need >= 0 or live <= 1 or fixed_live:
            break
        live -= 1
    if need < 0:
        raise ValueError(f"{fname}: the stand-in's structure alone has {got['sends']} sends, the real function {target['sends']}")
    # 2. columns: every branch may grow to the real function's aux budget; the structure's own maximum is got['aux']
    #    (the widest branch is a live one: the recursive call, in branch 0 also the u64 calls)
    room_total = target["aux"] - got["aux"]
    if room_total < 0:
        raise ValueError(f"{fname}: the stand-in's structure alone needs {got['aux']} aux columns, the real function has {target['aux']}")
    # room of a branch = room_total + (columns the widest structural branch has and this one has not); measured per kind of
    # branch by giving it one product more than everybody else could take
    def room_of(k):
        lo, hi = 0, target["aux"] + 1
        # largest number of products branch k can take without widening the function
        while lo < hi:
            mid = (lo + hi + 1) // 2
            s = fresh(live)
            s.branches[k].muls = mid
            if build(s)["aux"] <= target["aux"]:
                lo = mid
            else:
                hi = mid - 1
        return lo

    kinds = {}
    rooms = []
    for k in range(n_branches):
        kind = (spec.branches[k].live, k == 0)
        if kind not in kinds:
            kinds[kind] = room_of(k) if can_mul else room_total  # (a function without parameters has nothing to multiply)
        rooms.append(kinds[kind])
    # 3. exact width.  One branch has to be exactly as wide as the real function: the roomiest one is filled with one stored cell and
    #    loads of it (a load of an L-lane cell is one lookup like a store but L + 3 columns instead of 4 and no constraint more: the
    #    way the real functions spend their columns), what remains with products (one column and one constraint each); the other
    #    lookups are stores spread over the other branches, byte-pair range checks (3 columns) where 4 do not fit
    # (round 5) `dead_quota`: how many of the function's lookups sit in branches a run never takes -- measured on the real function
    # (tests/golden/fib_shape.json "lookup_sparsity": sends - live sends): the filler's lookups go to never-taken branches up to
    # that number and to live ones beyond it, so that the stand-in's permutation trace has the real one's share of identically-zero
    # columns (the prover leaves those out of the LDE); None: spread by room alone
    dead_quota = target.get("dead_lookups")
    is_dead = [not b.live for b in spec.branches]

    def place(pad_k, n_loads, l0):
        left = list(rooms)
        for b in spec.branches:
            b.cells, b.loads, b.ranges, b.muls, b.pinned = [], 0, 0, 0, False
        todo = need
        placed_dead = 0
        if n_loads >= 0 and need >= 1 + n_loads and left[pad_k] >= 4 + n_loads * (l0 + 3):
            bk = spec.branches[pad_k]
            bk.cells, bk.loads, bk.pinned = [l0], n_loads, n_loads > 0
            left[pad_k] -= 4 + n_loads * (l0 + 3)
            todo -= 1 + n_loads
            placed_dead += (1 + n_loads) if is_dead[pad_k] else 0
        order = [i for i in range(n_branches) if i != pad_k]
        for _ in range(todo):
            cand = order
            if dead_quota is not None:
                want_dead = placed_dead < dead_quota
                cand = [i for i in order if is_dead[i] == want_dead and left[i] >= 3] or order
            k = max(cand, key=lambda i: (left[i], -i)) if cand else pad_k
            if left[k] < 3 and left[pad_k] >= 3:
                k = pad_k
            placed_dead += 1 if is_dead[k] else 0
            if left[k] >= 4:
                spec.branches[k].cells.append(4)
                left[k] -= 4
            elif left[k] >= 3:
                spec.branches[k].ranges += 1
                left[k] -= 3
            else:
                return None
        spec.branches[pad_k].muls = left[pad_k]
        return left[pad_k]

    def place_spread():
        """The other way round: stores spread evenly first, then the widest branch reads its first cell back instead of storing
        its further ones."""
        left = list(rooms)
        for b in spec.branches:
            b.cells, b.loads, b.ranges, b.muls, b.pinned = [], 0, 0, 0, False
        placed_dead = 0
        for _ in range(need):
            cand = list(range(n_branches))
            if dead_quota is not None:
                want_dead = placed_dead < dead_quota
                cand = [i for i in cand if is_dead[i] == want_dead and left[i] >= 3] or cand
            k = max(cand, key=lambda i: (left[i], -i))
            placed_dead += 1 if is_dead[k] else 0
            if left[k] >= 4:
                spec.branches[k].cells.append(4)
                left[k] -= 4
            elif left[k] >= 3:
                spec.branches[k].ranges += 1
                left[k] -= 3
            else:
                return None
        k = min(range(n_branches), key=lambda i: (left[i], i))
        bk = spec.branches[k]
        best = (0, 0, 4)
        for l0 in MEM_LENS:
            for n in range(0, len(bk.cells)):
                if best[0] < n * (l0 - 1) <= left[k]:
                    best = (n * (l0 - 1), n, l0)
        absorbed, n_loads, l0 = best
        if n_loads:
            bk.cells = [l0] + bk.cells[1:len(bk.cells) - n_loads]
            bk.loads, bk.pinned = n_loads, True
        bk.muls = left[k] - absorbed
        return bk.muls

    if pad_first:
        pad_k = max(range(n_branches), key=lambda i: (rooms[i], -i))
        options = sorted(((rooms[pad_k] - 4 - n * (l0 + 3), -n, l0) for l0 in MEM_LENS for n in range(0, need) if rooms[pad_k] >= 4 + n * (l0 + 3)),
                         key=lambda o: (o[0], o[1]))
        for _, neg_n, l0 in options + [(0, 1, 4)]:  # last resort: no cell of its own in the padded branch, products only
            if place(pad_k, -neg_n, l0) is not None:
                break
        else:
            raise ValueError(f"{fname}: {target['sends']} lookups do not fit {n_branches} branches of {target['aux']} aux columns")
    elif place_spread() is None:
        raise ValueError(f"{fname}: {target['sends']} lookups do not fit {n_branches} branches of {target['aux']} aux columns")
    got = build(spec)
    assert (got["width"], got["sends"], got["receives"], got["sel"]) == (target["width"], target["sends"], target["receives"], target["sel"]), (fname, got, target)
    # 4. tuple words: size the stored cells (any length costs the same 4 columns)
    cells = [(bi, ci) for bi, b in enumerate(spec.branches) for ci in range(len(b.cells)) if not (b.pinned and ci == 0)]
    if cells:
        gap = target["interaction_tuple_words"] - got["interaction_tuple_words"]
        per_lane = 2  # one lane more in a stored cell = one word more in its receive and one in its send
        want = 4 * len(cells) + gap // per_lane
        lens = [2] * len(cells)
        budget = max(0, want - 2 * len(cells))
        for i in range(len(cells)):
            take = min(6, budget)
            take = take if take != 5 else 4  # no 7-wide table
            lens[i] = 2 + take
            budget -= take
        for (bi, ci), ln in zip(cells, lens):
            spec.branches[bi].cells[ci] = ln
    # 5. constraints: assertions in the never-taken branches (live ones when there is no other)
    got = build(spec)
    gap = target["constraints"] - got["constraints"]
    if gap > 0:
        dead = [i for i, b in enumerate(spec.branches) if not b.live] or list(range(n_branches))
        for a in range(gap):
            spec.branches[dead[a % len(dead)]].asserts += 1
    got = build(spec)
    assert (got["width"], got["sends"], got["receives"], got["sel"]) == (target["width"], target["sends"], target["receives"], target["sel"]), (fname, got, target)
    return spec, got


def _fit(fname, target, emit, n_branches, live, others_src, fixed_live=False, can_mul=True):
    """`_fit_once` for every admissible number of live branches (fewer live branches = fewer recursive calls = more of the
    lookups free to be loads, which fill columns without constraints) and both placements; the closest constraint count wins,
    then tuple words."""
    best = None
    err = None
    for lv in ([live] if fixed_live or live <= 1 else range(live, 0, -1)):
        for pad_first in (True, False):
            try:
                spec, got = _fit_once(fname, target, emit, n_branches, lv, others_src, fixed_live=True, can_mul=can_mul, pad_first=pad_first)
            except ValueError as e:
                err = e
                continue
            score = (abs(got["constraints"] - target["constraints"]), abs(got["interaction_tuple_words"] - target["interaction_tuple_words"]))
            if best is None or score < best[0]:
                best = (score, spec, got)
    if best is None:
        raise err
    return best[1], best[2]


_PLAN_CACHE = {}


# The stand-in's `eval` cannot hold the real function's share of live lookups in its live branches (their columns go to the recursive
# call): it ends up with 6 dead permutation columns where the real one has 2, at the tallest height of the machine.  eval_builtin_expr,
# half as tall, gives the difference back twice over, so that the machine's share of identically-zero permutation cells stays at or
# below the real machine's (tests/test_mix_programs.py holds it there; tools/measure_lookup_sparsity.py prints both tables).
DEAD_COLUMN_ADJUST = {"eval_builtin_expr": -6}
DEAD_COLUMN_ADJUST_MASTERMIND = {"eval_builtin_expr": -9}  # (the same trade on lurk-mix, whose eval_builtin_expr is half as tall as eval too)


def _plan(funcs, walkers, pre, phase_pre, u64_owner, live_wanted, sparsity=False):
    """Specs of every shaped function of a machine (independent of the row counts: they only appear as constants).
    `sparsity`: dial the walkers' live branches and the share of lookups in never-taken branches to the measured ones of a real
    `(fib N)` (fib_shape.json "lookup_sparsity")."""
    key = (sparsity, tuple(funcs), tuple(walkers), u64_owner, tuple(sorted(live_wanted.items())), tuple(sorted((k, tuple(v)) for k, v in pre.items())),
           tuple(sorted((k, tuple(sorted((b, tuple(x)) for b, x in v.items()))) for k, v in phase_pre.items())))
    if key in _PLAN_CACHE:
        return _PLAN_CACHE[key]
    chips = load_shape()["chips"]
    sparse = load_shape().get({"fib": "lookup_sparsity", "mastermind": "lookup_sparsity_mastermind"}.get(sparsity, "lookup_sparsity") if sparsity else "", {}).get("real", {}) if sparsity else {}
    have = set(funcs)
    fixed_src = "".join(FIXED[f] for f in funcs if f in FIXED)
    specs, fits = {}, {}
    for f in funcs:
        if f in FIXED:
            continue
        others = "".join(_stub(g) for g in funcs if g != f and g not in FIXED) + fixed_src
        t = chips[f]
        if f == "lurk_main":
            lines = _start_call(walkers[0], 1, "ev")
            emit = lambda s, lines=lines: emit_main(lines[:-1], lines[-1], "preallocate_symbols" in have, s)
            specs[f], got = _fit(f, t, emit, 1, 0, others)
        elif f in SHAPED_LEAVES:
            specs[f], got = _fit(f, t, lambda s, f=f: emit_leaf(f, s), t["sel"], 0, others, can_mul=bool(LURK_FUNCS[f][2]))
        else:
            i = walkers.index(f)
            nxt = walkers[i + 1] if i + 1 < len(walkers) else None
            base = [re.sub(r"\br(\d+)\b", r"q\1", x) for x in _start_call(nxt, 1, "nx")] if nxt else []
            nb = t["sel"] - 1
            assert nb >= 1, f"{f}: a walker needs two selectors"
            want = live_wanted.get(f)
            live = want or min(4, max(1, nb - 1))
            if f in sparse:
                t = dict(t)
                # (the real function's dead interactions do not all pair up into dead columns -- 58 dead lookups of eval_builtin_expr
                # make 52 dead columns --, the stand-in's branch-by-branch ones do: the quota is the number of dead COLUMNS, which
                # leaves the stand-in with a few live interactions more than the real function has)
                adjust = DEAD_COLUMN_ADJUST if sparsity in (True, "fib") else DEAD_COLUMN_ADJUST_MASTERMIND
                t["dead_lookups"] = max(0, sparse[f]["dead_columns"] + adjust.get(f, 0))
                if want is None:  # the selectors a real run takes, less the bottom frame's return
                    want = live = max(1, min(nb, sparse[f]["live_selectors"] - 1))
            emit = lambda s, f=f, base=base: emit_walker(f, s, pre[f], phase_pre[f], base)
            specs[f], got = _fit(f, t, emit, nb, min(live, nb), others, fixed_live=want is not None)
            specs[f].next_walker = nxt
        fits[f] = {"got": got, "target": {k: chips[f][k] for k in got}}
    _PLAN_CACHE[key] = (specs, fits)
    return specs, fits


def build_mix(name, funcs, counts, u64_owner=None, u64_every=4, fresh=None, sparsity=False):
    """funcs: function names in machine order (a subset of LURK_FUNC_ORDER, `lurk_main` first); counts: walker name -> rows.
    Walkers are chained: `lurk_main` starts the first one, each walker's bottom frame starts the next.  The u64 gadgets are
    called from branch 0 of `u64_owner`, which has `u64_every` live branches."""
    have = set(funcs)
    walkers = [f for f in funcs if f not in FIXED and f not in SHAPED_LEAVES and f != "lurk_main"]
    # chain order: a total function may not call a partial one (toplevel.rs), so the partial walkers come first
    walkers = [w for w in walkers if LURK_FUNCS[w][0]] + [w for w in walkers if not LURK_FUNCS[w][0]]
    for w in walkers:
        assert counts.get(w, 0) >= 1, f"no row count for {w}"
    pre = {w: [] for w in walkers}
    phase_pre = {w: {} for w in walkers}   # walker -> {live branch index: lines run only by the rows that take it}
    rows = {w: counts[w] for w in walkers}
    u64_owner = u64_owner or next((w for w in ("eval_binop_num", "apply") if w in have), None)
    for leafname, (owner, lines) in LEAF_CALLS.items():
        if leafname in have:
            assert owner in have, f"{leafname} needs its caller {owner}"
            pre[owner] += lines
            rows[leafname] = counts[owner] - 1
    u64_ops = [op for op in U64_CALLS if op in have]
    live_wanted = {}
    for w, (ln, every) in (fresh or {}).items():  # one new memory cell per row (every = 1) or per row of phase 0 of `every` live branches
        line = "let fresh = store(" + ", ".join(["@j", "one", "jn"] + ["zero"] * (ln - 3)) + ");"
        if every == 1:
            pre[w].append(line)
        else:
            phase_pre[w].setdefault(0, []).append(line)
            live_wanted[w] = every
    if u64_ops:
        assert u64_owner, "u64 gadgets need eval_binop_num or apply in the machine"
        R = min(u64_every, load_shape()["chips"][u64_owner]["sel"] - 1)
        for i, op in enumerate(u64_ops):  # dealt over the owner's live branches: op i is called by the rows of phase i mod R
            lines = phase_pre[u64_owner].setdefault(i % R, [])
            call = U64_CALLS[op]
            if "c8s" in call and C8S[0] not in lines:
                lines += C8S
            elif "c8)" in call and C8[0] not in lines:
                lines += C8
            lines.append(call)
        assert live_wanted.get(u64_owner, R) == R, "the u64 owner's live branches are already dialled otherwise"
        live_wanted[u64_owner] = R
    specs, fits = _plan(funcs, walkers, pre, phase_pre, u64_owner, live_wanted, sparsity)
    if u64_ops:
        R = specs[u64_owner].live
        steps = counts[u64_owner] - 1  # rows of the owner that are not its bottom frame; phases run 0, 1, .., R - 1, 0, ..
        for i, op in enumerate(u64_ops):
            rows[op] = max(0, -(-(steps - i % R) // R))
    srcs = {}
    for f in funcs:
        if f in FIXED:
            srcs[f] = FIXED[f]
        elif f in SHAPED_LEAVES:
            srcs[f] = emit_leaf(f, specs[f])
        elif f == "lurk_main":
            lines = _start_call(walkers[0], counts[walkers[0]] - 1, "ev")
            srcs[f] = emit_main(lines[:-1], lines[-1], "preallocate_symbols" in have, specs[f])
        else:
            nxt = specs[f].next_walker
            base = [re.sub(r"\br(\d+)\b", r"q\1", x) for x in _start_call(nxt, counts[nxt] - 1, "nx")] if nxt else []
            srcs[f] = emit_walker(f, specs[f], pre[f], phase_pre[f], base)
    if "preallocate_symbols" in have:
        rows["preallocate_symbols"] = 1
    rows["lurk_main"] = 1
    source = "".join(srcs[f] for f in funcs)
    return Mix(name, source, "lurk_main", counts["eval"], rows, [0] * 24, fits)


# the function chips of a real `(fib N)` run (tests/golden/fib_shape.json "fib"), in machine order
FIB_FUNCS = ["lurk_main", "preallocate_symbols", "eval", "eval_builtin_expr", "eval_binop_num", "eval_begin", "eval_letrec",
             "extend_env_with_mutuals", "eval_letrec_bindings", "apply", "env_lookup", "ingress", "egress", "hash4", "u64_add", "u64_sub",
             "u64_lessthan"]


def fib_counts(eval_rows: int) -> dict:
    """Walker rows of a fib run with `eval_rows` eval rows: the measured per-level slopes for the chips that grow with N, the
    measured sizes for those that do not (never more than the eval chip, so that it stays the tallest)."""
    shape = load_shape()
    per = shape["fib_per_level"]
    big = shape["fib"][max(shape["fib"], key=int)]["rows"]
    counts = {}
    for f in FIB_FUNCS:
        if f in FIXED or f in SHAPED_LEAVES or f == "lurk_main":
            continue
        if f in per:
            counts[f] = max(2, int(eval_rows * per[f] / per["eval"]))
        else:
            counts[f] = max(1, min(big[f], eval_rows // 2))
    counts["eval"] = eval_rows
    return counts


def fib_mix(eval_rows: int) -> Mix:
    """The chips of a real `(fib N)` run at the measured shape (see the module docstring): `apply` owns the u64 gadgets and calls
    them in branch 0 of 4 live branches (4 apply rows per fib level, one u64_add / u64_sub / u64_lessthan each; u64_add and u64_sub
    store one 8-lane cell per row), env_lookup stores one 5-lane cell per row."""
    return build_mix("fib-mix", FIB_FUNCS, fib_counts(eval_rows), u64_owner="apply", u64_every=4, fresh=FIB_FRESH, sparsity=True)


# lurk-mix (BASELINE config 5, `demo/mastermind.lurk`): every function of the Lurk toplevel.  Heights: the rows the REAL evaluator
# gives each chip on the folded mastermind script (tests/golden/fib_shape.json "mastermind"), scaled with the eval chip -- except
# the chips that read the program in (ingress / egress and the hashers under them: their rows follow the size of the source
# text, not the length of the run), which keep their measured sizes.  The 13 functions mastermind never calls (the bind / env /
# apply builtins, u64_mul, ...) stay in the machine at a token height, so that all 39 widths (9 ... 815) are proved.
INGRESS_SIDE = ("ingress", "egress", "hash3", "hash4", "hash5")
TOKEN_ROWS = 3


def lurk_fractions() -> dict:
    """walker -> (rows per eval row, or None for a fixed size; fixed size) from the measured mastermind run."""
    m = load_shape()["mastermind"]["rows"]
    e = m["eval"]
    out = {}
    for f in LURK_FUNC_ORDER:
        if f in FIXED or f in SHAPED_LEAVES or f == "lurk_main":
            continue
        if f in INGRESS_SIDE:
            out[f] = (None, m.get(f, TOKEN_ROWS))
        elif f in m:
            out[f] = (m[f] / e, None)
        else:
            out[f] = (None, TOKEN_ROWS)
    return out


def lurk_mix(eval_rows: int) -> Mix:
    counts = {}
    for f, (frac, fixed) in lurk_fractions().items():
        counts[f] = max(2, int(eval_rows * frac)) if frac is not None else max(2, min(fixed, max(2, eval_rows // 2)))
    counts["eval"] = eval_rows
    return build_mix("lurk-mix", list(LURK_FUNC_ORDER), counts, u64_owner="eval_binop_num", u64_every=2, fresh=LURK_FRESH, sparsity="mastermind")
