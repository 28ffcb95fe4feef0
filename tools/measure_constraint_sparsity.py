#!/usr/bin/env python3
"""How many constraints of a Lair function sit in branches a real `(fib N)` run never takes?  (Build container only: reads
/root/reference at run time through tools/lurk_reference.py, stores numbers only.)

The compiled quotient kernels evaluate a group of constraints `sel * x` only where `sel` is non-zero on some point of the wave
(csrc/jit.cpp: emit_constraint_function).  This script takes the generated source of a chip (LURKHIP_JIT_DUMP), reads the groups
and their factors off it, and marks a group dead when its factor is a sum of selector columns that are zero on every row of the
oracle's trace of the run -- for the reference's functions and for the fib-mix stand-in, so that what the device skips on the
stand-in can be held against what it would skip on the real machine.

    python tools/measure_constraint_sparsity.py [N] [--write]
"""
import ctypes as C
import json
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

TALL = ("eval", "eval_builtin_expr", "eval_binop_num", "apply")


def jit_source(top, name):
    from lurk_amd import _native as N
    from lurk_amd.air import ChipAir

    a = ChipAir.for_func(top, top.func_index(name))
    with tempfile.TemporaryDirectory() as d:
        os.environ["LURKHIP_JIT_DUMP"] = os.path.join(d, "k")
        os.environ["LURKHIP_JIT_CACHE"] = ""
        log = C.create_string_buffer(2048)
        try:
            r = N.lib.lurkhip_air_compile_check(a.handle, log, 2048)  # bytes of the code object, or a negative status
            assert r > 0, (name, r, log.value.decode("utf-8", "replace")[:400])
            return open(os.path.join(d, "k.hip")).read()
        finally:
            os.environ.pop("LURKHIP_JIT_DUMP", None)
            os.environ.pop("LURKHIP_JIT_CACHE", None)


def groups_of(src):
    """[(factor text or None, constraints, arithmetic instructions inside the branch)] of the chip's constraint pieces, plus a
    resolver of SSA names to the set of main columns they sum (None: not a plain sum of columns)."""
    defs = {}
    out = []
    for fn in re.findall(r"void quot_cons\d+\(.*?\n}\n", src, flags=re.S):
        cur = None
        for line in fn.splitlines():
            m = re.match(r"\s*const uint32_t (t\d+) = bb::(\w+)\((.+), (.+)\);", line)
            if m:
                defs[m.group(1)] = (m.group(2), m.group(3), m.group(4))
                if cur is not None:
                    cur[2] += 1
                continue
            m = re.match(r"\s*if \(sink\.cond_live\((.+)\)\) {", line)
            if m:
                cur = [m.group(1), 0, 0]
                continue
            if "sink.assert_zero(" in line:
                if cur is not None:
                    cur[1] += 1
                else:
                    out.append((None, 1, 0))
                continue
            if "sink.skip_asserts(" in line:
                out.append(tuple(cur))
                cur = None

    def columns(x):
        m = re.fullmatch(r"s\.main_l\[(\d+)\]", x)
        if m:
            return {int(m.group(1))}
        if x in defs and defs[x][0] == "add":
            a, b = columns(defs[x][1]), columns(defs[x][2])
            return None if a is None or b is None else a | b
        return None

    return out, columns


def chip_numbers(top, name, rows, n_sel):
    w = len(rows[0])
    live_cols = {c for c in range(w - n_sel, w) if any(r[c] for r in rows)}
    groups, columns = groups_of(jit_source(top, name))
    total = sum(g[1] for g in groups)
    dead = instr_dead = 0
    for factor, n, ins in groups:
        if factor is None:
            continue
        cols = columns(factor)
        if cols is not None and all(c >= w - n_sel for c in cols) and not (cols & live_cols):
            dead += n
            instr_dead += ins
    return {"constraints": total, "dead_constraints": dead, "dead_branch_instructions": instr_dead}


def machine_numbers(top, otop, q, witness):
    from oracle import lair as ol

    out = {}
    for g in otop.funcs:
        if g["name"] not in TALL or not q.func[g["index"]]:
            continue
        rows, _ = ol.generate_trace(otop, g["name"], q, witness=witness)
        out[g["name"]] = chip_numbers(top, g["name"], rows, otop.layout(g)["sel"])
    return out


def real(n):
    import lurk_reference as lr
    import measure_lurk_shape as ms
    from lurk_amd import zstore as zs
    from oracle import binding
    from oracle import lair as ol
    from test_lair_gpu import oracle_chip_callbacks

    binding.build()
    rl = ms.RealLurk()
    otop = ol.Toplevel(rl.source, chips=ol.lurk_chips())
    poseidon, witness = oracle_chip_callbacks(binding)
    z = zs.ZStore(rl.hasher)
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
    return machine_numbers(rl.top, otop, q, witness)


def stand_in(eval_rows=256):
    from lurk_amd import lair
    from lurk_amd.programs import lurk_mix as lm
    from oracle import binding
    from oracle import lair as ol
    from test_lair_gpu import oracle_chip_callbacks

    binding.build()
    mix = lm.fib_mix(eval_rows)
    otop = ol.Toplevel(mix.source, chips=ol.lurk_chips())
    poseidon, witness = oracle_chip_callbacks(binding)
    q = ol.QueryRecord(otop)
    ol.execute(otop, mix.entry, list(mix.main_args), q, poseidon=poseidon)
    return machine_numbers(lair.Toplevel(mix.source, lurk_chips=True), otop, q, witness)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    r, m = real(n), stand_in()
    print("%-20s | %5s %5s %6s | %5s %5s %6s" % ("chip", "cons", "dead", "instr", "cons", "dead", "instr"))
    for c in TALL:
        print("%-20s | %5d %5d %6d | %5d %5d %6d" % (c, r[c]["constraints"], r[c]["dead_constraints"], r[c]["dead_branch_instructions"],
                                                     m[c]["constraints"], m[c]["dead_constraints"], m[c]["dead_branch_instructions"]))
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "tests", "golden", "fib_shape.json")
        with open(path) as f:
            shape = json.load(f)
        shape["constraint_sparsity"] = {
            "_about": "tools/measure_constraint_sparsity.py on (fib %d), the reference's functions: per tall chip, constraints / constraints whose selector factor is zero on every row of the run / arithmetic instructions only those constraints need" % n,
            "fib_n": n, "real": r}
        with open(path, "w") as f:
            json.dump(shape, f, indent=1, sort_keys=False)
            f.write("\n")


if __name__ == "__main__":
    main()
