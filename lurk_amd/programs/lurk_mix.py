"""Width-matched stand-ins for the Lurk evaluator's Lair machine: the `fib-mix` and `lurk-mix` workloads.

The reference's evaluator (/root/reference/src/core/eval_direct.rs: 39 Lair functions) is program text and cannot be
shipped here; what the proving hot path sees of it is a *machine shape*: one chip per function with the widths listed by
`test_widths` (/root/reference/src/core/eval_direct.rs:2025-2063), `partial` functions carrying depth bytes and byte-table
lookups (eval_direct.rs:121,388,445,1184,1779), extern chips (hashers, u64 gadgets) behind thin wrapper functions
(/root/reference/src/core/misc.rs), rows that mix calls, memory loads / stores and field arithmetic, and chip heights in
the ratios a program produces.  This module generates Lair functions with exactly those names, signatures (input / output
sizes, `partial` / `invertible` flags), widths and kinds of rows, and row counts one can dial:

* every non-leaf function is a *walker*: `F(j, ...)` calls `F(j - 1, ...)` until its counter is zero, so `F(n - 1, ...)`
  yields exactly n queries = n trace rows; per step it calls the leaf functions it is paired with (hashers, u64 gadgets)
  on step-dependent arguments, stores one fresh memory cell where the spec says so, and is padded to the reference's
  width with lookup-dense filler (a constant cell stored and loaded back: two lookups per 4 + len + 3 columns, no new
  memory rows -- the way the evaluator keeps re-reading the cells of the expression it evaluates) and, for the last few
  columns, products;
* `lurk_main` (partial, 24 inputs, 16 outputs: the 44-lane public-value layout of
  /root/reference/src/core/stark_machine.rs:16-17) starts every walker once.

`fib_mix(eval_rows)`: the chips a `fib` run touches with the row ratios of SURVEY.md appendix C
(per 13 eval rows: 5 eval_builtin_expr, 4 eval_binop_num, 2 apply, 5 env_lookup, ~1 of each u64 op, hash chips of a few
hundred rows); `lurk_mix(eval_rows)`: all 39 functions + the 6 memory tables + byte table + entrypoint (BASELINE config 5,
`demo/mastermind.lurk`: irregular widths 9 ... 815) with heights from a fixed table.  Row ratios are hand estimates
(appendix C says so too) -- the widths, flags and chip set are the reference's.
"""
from __future__ import annotations

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

import re

FILL_LENS = (3, 4, 5, 8, 2, 6)  # memory-table lengths of the filler cells, round robin (/root/reference/src/lair/execute.rs:243-244)

# functions that are not walkers: fixed text (LEAVES, eval_coroutine_expr) or filler only
FILL_LEAVES = ("preallocate_symbols", "coerce_if_sym", "open_comm", "equal_inner")
FIXED = dict(LEAVES)
FIXED["eval_coroutine_expr"] = "fn eval_coroutine_expr(a0, a1, a2, a3): [2] {\n    let zero = 0;\n    return (zero, zero)\n}\n"

# walker -> leaf calls made on every step, before the recursive call.  `@j` is the step counter (distinct per row), `@acc`
# the u64 accumulator pointer threaded through the recursion (the parameter named in THREAD).
U64_OWNER_PARAM = {"eval_binop_num": "a4", "apply": "a3"}
U64_PHASE_PARAM = {"eval_binop_num": "a5", "apply": "a1"}
LEAF_CALLS = {
    "hash4": ("ingress", ["let pre4: [32] = (a0, @arr, a0, @arr);", "let h4: [8] = call(hash4, pre4);"]),
    "hash3": ("ingress", ["let pre3: [24] = (a0, @arr, a0);", "let h3: [8] = call(hash3, pre3);"]),
    "hash5": ("ingress", ["let pre5: [40] = (a0, @arr, a0, @arr, a0);", "let h5: [8] = call(hash5, pre5);"]),
    "coerce_if_sym": ("eval_unop", ["let cs = call(coerce_if_sym, @j);"]),
    "open_comm": ("eval_opening_unop", ["let (oc0, oc1) = call(open_comm, @j);"]),
    "equal_inner": ("equal", ["let ei = call(equal_inner, @j, a1, a2, a3);"]),
    "eval_coroutine_expr": ("eval_builtin_expr", ["let (co0, co1) = call(eval_coroutine_expr, @j, a1, a2, a3);"]),
}
U64_CALLS = {
    "u64_add": "let acc2 = call(u64_add, @acc, c8);",
    "u64_sub": "let us = call(u64_sub, @acc, c8);",
    "u64_mul": "let um = call(u64_mul, @acc, @acc);",
    "u64_divrem": "let (uq, ur) = call(u64_divrem, @acc, c8);",
    "u64_lessthan": "let ul = call(u64_lessthan, @acc, c8);",
    "u64_iszero": "let uz = call(u64_iszero, @acc);",
    "digest_equal": "let de = call(digest_equal, @acc, c8);",
    "big_num_lessthan": "let bl = call(big_num_lessthan, @acc, c8);",
}
C8 = ["let b1 = 3;", "let b2 = 1;", "let c8 = store(b1, b2, zero, zero, zero, zero, zero, zero);"]  # the u64 constant 259 (one shared cell)
FRESH_STORE = ("env_lookup",)  # walkers that allocate one new memory cell per step


def _sig(name):
    sizes = LURK_FUNCS[name][2]
    return [(f"a{i}", s) for i, s in enumerate(sizes)]


def _counter(name):
    """(counter variable, array parameter it is lane 0 of or None)."""
    for p, s in _sig(name):
        if s == 1 and p != U64_OWNER_PARAM.get(name) and p != U64_PHASE_PARAM.get(name):
            return p, None
    arr, _ = _sig(name)[-1]
    return "j", arr


def _head(name):
    partial, invertible, _, out, _ = LURK_FUNCS[name]
    sig = ", ".join(p if s == 1 else f"{p}: [{s}]" for p, s in _sig(name))
    return ("partial " if partial else "") + ("invertible " if invertible else "") + f"fn {name}({sig}): [{out}] {{\n"


def _filler(cells, muls, start, seed):
    L = []
    for c in range(cells):
        ln = FILL_LENS[(start + c) % len(FILL_LENS)]
        names = [f"k{c}_{i}" for i in range(ln)]
        L += [f"let {nm} = {100 + 10 * c + i};" for i, nm in enumerate(names)]
        L.append(f"let fp{c} = store({', '.join(names)});")
        L.append(f"let ({', '.join('f%d_%d' % (c, i) for i in range(ln))}) = load(fp{c});")
    seed = seed or ("f0_0" if cells else None)
    prev = None
    for m in range(muls):
        if seed is None:
            raise ValueError("product filler needs a variable")
        L.append(f"let m{m} = mul({prev or seed}, {seed});")
        prev = f"m{m}"
    return L, prev


def _ret(out, pool):
    if out == 0:
        return "return ()"
    vals = [pool[k % len(pool)] for k in range(out)]
    return f"return ({', '.join(vals)})" if out != 1 else f"return {vals[0]}"


def emit_walker(name, pre, base, cells, muls, start):
    _, _, _, out, _ = LURK_FUNCS[name]
    sig = _sig(name)
    j, arr = _counter(name)
    acc = U64_OWNER_PARAM.get(name)
    L = ["let zero = 0;", "let one = 1;"]
    if arr:
        asz = dict(sig)[arr]
        L.append(f"let ({', '.join(['j'] + [f'{arr}_{k}' for k in range(1, asz)])}) = {arr};")
    L.append(f"if !{j} {{")
    L += ["    " + b for b in base]
    L.append("    " + _ret(out, ["zero"]))
    L.append("}")
    L.append(f"let jn = sub({j}, one);")
    sub = lambda t: t.replace("@j", j).replace("@acc", acc or "zero").replace("@arr", arr or "zero")
    L += [sub(x) for x in pre]
    args = []
    for p, s in sig:
        if p == j:
            args.append("jn")
        elif p == arr:
            L.append(f"let nxt: [{s}] = ({', '.join(['jn'] + [f'{arr}_{k}' for k in range(1, s)])});")
            args.append("nxt")
        elif p == acc and any("nacc" in x for x in pre):
            args.append("nacc")
        elif p == acc and any("acc2" in x for x in pre):
            args.append("acc2")
        elif p == U64_PHASE_PARAM.get(name) and any("nph" in x for x in pre):
            args.append("nph")
        else:
            args.append(p)
    rets = [f"r{k}" for k in range(out)]
    L.append(f"let {'(' + ', '.join(rets) + ')' if out != 1 else rets[0]} = call({name}, {', '.join(args)});")
    if name in FRESH_STORE:
        L.append(f"let fresh = store({j}, {rets[0]}, one);")
    fl, prev = _filler(cells, muls, start, j)
    L += fl
    L.append(_ret(out, rets + ([prev] if prev else []) + [j]))
    return _head(name) + "".join("    " + x + "\n" for x in L) + "}\n"


def emit_leaf(name, cells, muls, start):
    _, _, _, out, _ = LURK_FUNCS[name]
    sig = _sig(name)
    seed = sig[0][0] if sig and sig[0][1] == 1 else None
    L = ["let zero = 0;", "let one = 1;"]
    fl, prev = _filler(cells, muls, start, seed)
    L += fl
    L.append(_ret(out, ([prev] if prev else []) + ([seed] if seed else []) + ["one"]))
    return _head(name) + "".join("    " + x + "\n" for x in L) + "}\n"


def emit_main(first, first_args_lines, first_call, has_prealloc, cells, muls, start):
    L = ["let zero = 0;", "let one = 1;"]
    if has_prealloc:
        L.append("let () = call(preallocate_symbols, );")
    L += first_args_lines
    L.append(first_call)
    L.append("let (t0, t1, t2, t3, t4, t5, t6, t7) = a0;")
    fl, prev = _filler(cells, muls, start, "t0")
    L += fl
    pool = ["r0", "r1"] + ([prev] if prev else []) + ["t0", "t1", "one"]
    L.append(_ret(16, pool))
    return _head("lurk_main") + "".join("    " + x + "\n" for x in L) + "}\n"


def _fill_cost(n_cells: int, start: int) -> int:
    return sum(4 + FILL_LENS[(start + c) % len(FILL_LENS)] + 3 for c in range(n_cells))


def _start_call(name, count, tag):
    """Lines that start walker `name` with `count` steps from inside another function: constants for its arguments, the shared
    u64 cell for its accumulator."""
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
        elif p == U64_PHASE_PARAM.get(name):
            args.append("one")
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


def build_mix(name, funcs, counts, u64_owner=None, u64_every_other_step=False):
    """funcs: function names in machine order (a subset of LURK_FUNC_ORDER, `lurk_main` first); counts: walker name -> rows.
    Walkers are chained: `lurk_main` starts the first one, each walker's bottom frame starts the next."""
    from .. import lair

    have = set(funcs)
    walkers = [f for f in funcs if f not in FIXED and f not in FILL_LEAVES and f != "lurk_main"]
    # chain order: a total function may not call a partial one (toplevel.rs), so the partial walkers come first
    walkers = [w for w in walkers if LURK_FUNCS[w][0]] + [w for w in walkers if not LURK_FUNCS[w][0]]
    for w in walkers:
        assert counts.get(w, 0) >= 1, f"no row count for {w}"
    pre = {w: [] for w in walkers}
    rows = {w: counts[w] for w in walkers}
    u64_owner = u64_owner or next((w for w in ("eval_binop_num", "apply") if w in have), None)
    for leafname, (owner, lines) in LEAF_CALLS.items():
        if leafname in have:
            assert owner in have, f"{leafname} needs its caller {owner}"
            pre[owner] += lines
            rows[leafname] = counts[owner] - 1
    u64_ops = [op for op in U64_CALLS if op in have]
    if u64_ops:
        assert u64_owner, "u64 gadgets need eval_binop_num or apply in the machine"
        pre[u64_owner] += C8 + [U64_CALLS[op] for op in u64_ops]
        steps = counts[u64_owner] - 1
        if u64_every_other_step:
            # the accumulator pointer advances only when the phase parameter is 1, and the phase flips every step: on the
            # other steps the u64 calls repeat the previous step's queries (memoised: no new rows)
            ph = U64_PHASE_PARAM[u64_owner]
            pre[u64_owner] += ["let dacc = sub(acc2, @acc);", f"let pd = mul({ph}, dacc);", "let nacc = add(@acc, pd);", f"let nph = sub(one, {ph});"]
            steps = (steps + 1) // 2
        for op in u64_ops:
            rows[op] = steps
    # every walker: base case starts the next one
    base = {}
    for i, w in enumerate(walkers):
        base[w] = _start_call(walkers[i + 1], counts[walkers[i + 1]] - 1, "nx") if i + 1 < len(walkers) else []
        # the base case must not shadow r0..: rename the results of the started walker
        base[w] = [re.sub(r"\br(\d+)\b", r"q\1", x) for x in base[w]]
    srcs = {}
    fixed_src = "".join(FIXED[f] for f in funcs if f in FIXED)

    def layout_width(src_all, fname):
        top = lair.Toplevel(src_all, lurk_chips=True)
        return top.func_info(top.func_index(fname))["layout"].total()

    # signatures-only stubs let one function be compiled at a time against the others
    def stub(f):
        _, _, _, out, _ = LURK_FUNCS[f]
        body = "    let zero = 0;\n    " + _ret(out, ["zero"]) + "\n"
        return _head(f) + body + "}\n"

    def fit(fname, emit, start):
        target = LURK_FUNCS[fname][4]
        others = "".join(stub(g) for g in funcs if g != fname and g not in FIXED) + fixed_src
        # the width is the maximum over the function's branches, so the filler of the main branch only starts to count once it
        # overtakes the others: grow it until the layout reports the target
        fill = 0
        for _ in range(8):
            cells = 0
            while _fill_cost(cells + 1, start) <= fill:
                cells += 1
            muls = fill - _fill_cost(cells, start)
            src = emit(cells, muls)
            got = layout_width(others + src, fname)
            if got == target:
                return src
            if got > target:
                raise ValueError(f"{fname}: {got} columns with {fill} filler columns, target {target}")
            fill += target - got
        raise AssertionError(f"{fname}: could not reach width {target}")

    k = 0
    for f in funcs:
        if f in FIXED:
            srcs[f] = FIXED[f]
        elif f in FILL_LEAVES:
            srcs[f] = fit(f, lambda c, m, f=f, k=k: emit_leaf(f, c, m, k), k)
        elif f == "lurk_main":
            first = walkers[0]
            lines = _start_call(first, counts[first] - 1, "ev")
            srcs[f] = fit(f, lambda c, m, k=k: emit_main(first, lines[:-1], lines[-1], "preallocate_symbols" in have, c, m, k), k)
        else:
            srcs[f] = fit(f, lambda c, m, f=f, k=k: emit_walker(f, pre[f], base[f], c, m, k), k)
        k += 1
    if "preallocate_symbols" in have:
        rows["preallocate_symbols"] = 1
    rows["lurk_main"] = 1
    source = "".join(srcs[f] for f in funcs)
    return Mix(name, source, "lurk_main", counts["eval"], rows, [0] * 24)


FIB_FUNCS = ["lurk_main", "eval", "eval_builtin_expr", "eval_binop_num", "apply", "env_lookup", "ingress", "egress", "hash3", "hash4", "hash5",
             "u64_add", "u64_sub", "u64_lessthan"]


def fib_mix(eval_rows: int) -> Mix:
    """The chips of a `fib` run (SURVEY.md 8a T1 widths) at appendix C's row ratios: per 13 eval rows 5 eval_builtin_expr,
    4 eval_binop_num, 2 apply, 5 env_lookup (one fresh memory cell per row), 1 each of u64_add / u64_sub / u64_lessthan (`apply`
    owns them and advances its u64 accumulator every other step; each u64_add / u64_sub row stores one 8-lane cell); ingress /
    egress / hash chips: a few hundred rows whatever the run length."""
    e = eval_rows
    small = max(2, min(256, e // 16))
    counts = {"eval": e, "eval_builtin_expr": max(2, 5 * e // 13), "eval_binop_num": max(2, 4 * e // 13), "apply": max(3, 2 * e // 13),
              "env_lookup": max(2, 5 * e // 13), "ingress": small, "egress": max(2, small // 4)}
    return build_mix("fib-mix", FIB_FUNCS, counts, u64_owner="apply", u64_every_other_step=True)


# lurk-mix: every function of the Lurk toplevel; heights as fractions of the eval chip (hand-set: eval / apply / env_lookup /
# builtin dispatch largest, one-off forms small, hash chips capped at 2^12 as SURVEY.md 8d config 5 prescribes), made ragged so
# that padded heights differ from row counts
LURK_FRACTIONS = {
    "eval": 1.0, "eval_builtin_expr": 0.42, "eval_bind_builtin": 0.03, "eval_env_builtin": 0.02, "eval_apply_builtin": 0.03,
    "eval_opening_unop": 0.012, "eval_hide": 0.012, "eval_unop": 0.11, "eval_binop_num": 0.21, "eval_binop_misc": 0.17,
    "eval_begin": 0.06, "eval_list": 0.09, "eval_let": 0.12, "eval_letrec": 0.05, "extend_env_with_mutuals": 0.05,
    "eval_letrec_bindings": 0.05, "equal": 0.04, "car_cdr": 0.1, "apply": 0.45, "env_lookup": 0.48, "ingress": 0.01, "egress": 0.008,
}


def lurk_mix(eval_rows: int) -> Mix:
    counts = {}
    for f, frac in LURK_FRACTIONS.items():
        c = max(2, int(eval_rows * frac))
        if f in ("ingress", "egress"):
            c = min(c, 4096)
        counts[f] = c
    return build_mix("lurk-mix", list(LURK_FUNC_ORDER), counts)
