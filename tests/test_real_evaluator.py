"""The reference's OWN Lurk evaluator through this repo's host code (build container only: the functions are read from
/root/reference at run time by tools/lurk_reference.py and exist nowhere in the repo; every test here skips on a box
without the reference).

Pins, all reference-held:
  * the 39 literals of `test_widths` (/root/reference/src/core/eval_direct.rs:2025-2063) on the real function bodies, through
    the product's compiler (lurk_amd/csrc/lair/compile.cpp) and the oracle's (oracle/lair.py) -- T1's strongest in-tree vector;
  * `(fib N)` of /root/reference/benches/fib.rs:36-44 evaluates to fib(N) on the product's interpreter and on the oracle's;
  * the property the reference's evaluator tests check on every case (/root/reference/src/core/tests/mod.rs:28-74 through
    /root/reference/src/air/debug.rs:119-206): on the traces of a real Lurk evaluation every constraint of every chip vanishes
    and the lookups of the whole machine balance -- here with the oracle's trace generator and AIR on the reference's functions;
  * tests/golden/fib_shape.json is what tools/measure_lurk_shape.py measures today.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import lurk_reference as lr  # noqa: E402

pytestmark = pytest.mark.skipif(not lr.available(), reason="/root/reference is not on this box")


@pytest.fixture(scope="module")
def real():
    import measure_lurk_shape as ms

    return ms.RealLurk()


@pytest.fixture(scope="module")
def otop(real):
    from oracle import lair as ol

    return ol.Toplevel(real.source, chips=ol.lurk_chips())


def test_all_39_widths_on_the_real_bodies_product_compiler(real):
    want = lr.test_widths()
    assert real.names == lr.native_func_order() and len(real.names) == 39
    assert real.widths() == want


def test_all_39_widths_on_the_real_bodies_oracle_compiler(real, otop):
    want = lr.test_widths()
    got = {f["name"]: sum(otop.layout(f).values()) for f in otop.funcs}
    assert got == want
    # ... and the two compilers agree on every part of every layout, not only on the sum
    for f in otop.funcs:
        lay = real.top.func_info(real.top.func_index(f["name"]))["layout"]
        assert otop.layout(f) == {"nonce": lay.nonce, "input": lay.input, "output": lay.output, "aux": lay.aux, "sel": lay.sel}, f["name"]


def test_table_of_the_stand_in_machine_is_the_reference_s(real):
    """lurk_amd/programs/lurk_mix.py's LURK_FUNCS (signatures, flags, widths) against the real functions."""
    from lurk_amd.programs import lurk_mix as lm

    assert set(lm.LURK_FUNCS) == set(real.names)
    for n in real.names:
        info = real.top.func_info(real.top.func_index(n))
        partial, invertible, in_sizes, out, width = lm.LURK_FUNCS[n]
        assert (partial, invertible, sum(in_sizes), out, width) == (info["partial"], info["invertible"], info["input_size"], info["output_size"],
                                                                    info["layout"].total()), n


def _oracle_run(real, otop, oracle, text):
    import measure_lurk_shape as ms
    from lurk_amd import zstore as zs
    from oracle import lair as ol
    from test_lair_gpu import oracle_chip_callbacks

    poseidon, witness = oracle_chip_callbacks(oracle)
    z = zs.ZStore(real.hasher)
    zp = ms.intern_syntax(z, lr.read_lurk(text))
    q = ol.QueryRecord(otop)
    i4 = otop.index["hash4"]
    for pre, dig in z.hashes.items():
        if len(pre) == 32:
            q.inv[i4][tuple(dig)] = tuple(pre)
    args = [0] * 24
    args[0] = zp.tag
    args[8:16] = zp.digest
    out = ol.execute(otop, "lurk_main", args, q, poseidon=poseidon)
    return out, q, witness


@pytest.mark.parametrize("n", [1, 6])
def test_fib_on_both_interpreters(real, otop, oracle, n):
    text = lr.fib_program(n)
    out, q, _ = real.run(text)
    v = [0, 1, 1, 2, 3, 5, 8][n]
    assert list(out) == [lr.enums()["Tag"]["U64"]] + [0] * 7 + [v] + [0] * 7
    oout, oq, _ = _oracle_run(real, otop, oracle, text)
    assert list(oout) == list(out)
    assert list(oq.public_values) == q.expect_public_values()
    rows, mem, nbytes = real.record_counts(q)
    assert {f["name"]: len(oq.func[f["index"]]) for f in otop.funcs} == rows
    from oracle import lair as ol

    assert {str(l): len(oq.mem[i]) for i, l in enumerate(ol.MEM_TABLE_SIZES)} == mem
    assert len(oq.bytes) == nbytes


def test_real_evaluation_vanishes_and_balances(real, otop, oracle):
    """debug_constraints_collecting_queries's property (/root/reference/src/air/debug.rs:119-206) on `(fib 3)` evaluated by the
    reference's functions: 17 Func chips with rows, 3 memory tables, the byte table, the entrypoint."""
    from oracle import air as oa
    from oracle import lair as ol

    out, q, witness = _oracle_run(real, otop, oracle, lr.fib_program(3))
    assert out[8] == 2
    pv = q.public_values
    chips = [(oa.EntrypointAir(otop.index["lurk_main"], len(pv)), [list(pv)], None)]
    with_rows = 0
    for g in otop.funcs:
        if not q.func[g["index"]]:
            continue  # chips without rows are not part of the shard (lair_chip.rs:122-133)
        rows, _ = ol.generate_trace(otop, g["name"], q, witness=witness)
        if rows:
            with_rows += 1
            chips.append((oa.FuncAir(otop, g["name"]), rows, None))
    assert with_rows == 17
    for ml in ol.MEM_TABLE_SIZES:
        chips.append((oa.MemAir(ml), ol.mem_trace(q, ml), None))
    prep = [[i & 0xFF, i >> 8, int((i & 0xFF) < (i >> 8)), (i & 0xFF) & (i >> 8), (i & 0xFF) ^ (i >> 8), (i & 0xFF) | (i >> 8)] for i in range(1 << 16)]
    chips.append((oa.BytesAir(), ol.bytes_trace(q), prep))
    assert oa.debug_check(chips, public=pv) > 0


def test_committed_shape_is_a_fresh_measurement(real):
    with open(os.path.join(ROOT, "tests", "golden", "fib_shape.json")) as f:
        shape = json.load(f)
    assert shape["func_order"] == real.names
    assert shape["chips"] == json.loads(json.dumps(real.chip_shapes()))
    out, q, _ = real.run(lr.fib_program(1000))
    rows, mem, nbytes = real.record_counts(q)
    got = shape["fib"]["1000"]
    assert got["rows"] == {c: r for c, r in rows.items() if r}
    assert got["mem_rows"] == {l: r for l, r in mem.items() if r}
    assert got["byte_records"] == nbytes


def test_mastermind_script_holds_under_the_real_evaluator(real):
    """BASELINE config 5's program: demo/mastermind.lurk with its REPL commands folded into one expression (13 `def`, 3 `defrec`, 24
    chain transitions, 41 assertions).  The fold evaluates to `t` only when every assertion of the script holds -- letrec, closures,
    `hide` / `commit` / `open` (hash3 inverse queries made and used within one run), u64 arithmetic and comparisons, `eq` on
    data: a broad check of the interpreter the row streams come from -- and the rows per chip are the ones the lurk-mix heights
    are dialled from (tests/golden/fib_shape.json "mastermind")."""
    out, q, _ = real.run(lr.fold_repl_script(lr.demo_script("mastermind.lurk")))
    t_digest = [int(x) for x in real.resolver.digest[("lurk", "t")]]
    assert list(out) == [lr.enums()["Tag"]["Sym"]] + [0] * 7 + t_digest
    rows, mem, nbytes = real.record_counts(q)
    with open(os.path.join(ROOT, "tests", "golden", "fib_shape.json")) as f:
        m = json.load(f)["mastermind"]
    assert m["rows"] == {c: r for c, r in rows.items() if r}
    assert m["mem_rows"] == {l: r for l, r in mem.items() if r} and m["byte_records"] == nbytes
    # one assertion made false: the fold must not evaluate to `t` any more
    broken = lr.demo_script("mastermind.lurk").replace("!(assert-eq '(t 3 2) (maybe-remove 1 '(1 2 3)))", "!(assert-eq '(t 3 3) (maybe-remove 1 '(1 2 3)))", 1)
    out2, _, _ = real.run(lr.fold_repl_script(broken))
    assert list(out2)[0] == lr.enums()["Tag"]["Key"]  # :assertion-failed


def test_committed_lookup_sparsity_is_a_fresh_measurement():
    """tests/golden/fib_shape.json "lookup_sparsity" (round 5: which interactions of the real functions are ever real on a `(fib N)`
    run, which permutation columns never are -- what the fib-mix stand-in's branches are dialled to and what the prover's dead-batch
    skip and sparse permutation LDE are worth on a real machine) is what tools/measure_lookup_sparsity.py measures today."""
    import measure_lookup_sparsity as msp

    with open(os.path.join(ROOT, "tests", "golden", "fib_shape.json")) as f:
        sp = json.load(f)["lookup_sparsity"]
    got = msp.real_fib(sp["fib_n"])
    assert {c: s for c, (s, _) in got.items()} == sp["real"]
    # the liveness pattern does not depend on N once every branch of the recursion has been taken
    again = msp.real_fib(sp["fib_n"] + 5)
    assert {c: s for c, (s, _) in again.items()} == sp["real"]


def test_committed_mastermind_lookup_sparsity_is_a_fresh_measurement():
    """... and the same table for demo/mastermind.lurk (`lookup_sparsity_mastermind`: what lurk-mix is dialled to), from a run of the
    script under the reference's functions on the ORACLE's interpreter (its assertions hold there too)."""
    import measure_lookup_sparsity as msp

    with open(os.path.join(ROOT, "tests", "golden", "fib_shape.json")) as f:
        sp = json.load(f)["lookup_sparsity_mastermind"]
    got = msp.real_mastermind()
    assert {c: s for c, (s, _) in got.items()} == sp["real"]


@pytest.mark.parametrize("script", ["mini-mastermind.lurk", "simple.lurk"])
def test_smaller_demo_scripts_hold_under_the_real_evaluator(real, script):
    """The demo scripts whose REPL commands the fold covers (`def`, `defrec`, `assert-eq`; the others drive the CLI: `prove`,
    `verify`, `chain`, protocols): every assertion of the script holds under the reference's functions on the product's interpreter."""
    out, _, _ = real.run(lr.fold_repl_script(lr.demo_script(script)))
    t_digest = [int(x) for x in real.resolver.digest[("lurk", "t")]]
    assert list(out) == [lr.enums()["Tag"]["Sym"]] + [0] * 7 + t_digest


def test_ingress_egress_round_trip_on_the_reference_s_own_samples(real):
    """`test_ingress_egress` (/root/reference/src/core/eval_direct.rs:2067-2123): thirteen pieces of Lurk data, read here from that
    test at run time; `ingress` (digest -> pointers, through the injected hash4 preimages) followed by `egress` gives the tag and
    the digest back -- on the reference's two functions under the product's compiler and interpreter."""
    import re

    import measure_lurk_shape as ms
    from lurk_amd import zstore as zs

    src = lr._read("src/core/eval_direct.rs")
    codes = [bytes(c, "utf-8").decode("unicode_escape") for c in re.findall(r'assert_ingress_egress_correctness\("((?:[^"\\]|\\.)*)"\)', src)]
    assert len(codes) >= 13
    i4 = real.top.func_index("hash4")
    for code in codes:
        z = zs.ZStore(real.hasher)
        if code == "~()":          # the root symbol, the root keyword, a symbol of the root package: reader forms of the full parser
            zp = z.intern_symbol([])
        elif code == "~:()":
            zp = z.intern_symbol([], keyword=True)
        elif code.startswith(".") and len(code) > 1:
            zp = z.intern_symbol([code[1:]])
        elif code == "()":
            zp = z.nil
        else:
            zp = ms.intern_syntax(z, lr.read_lurk(code))
        q = real.lair.QueryRecord(real.top)
        for pre, dig in z.hashes.items():
            if len(pre) == 32:
                q.inject_inv_query(i4, list(pre), list(dig))
        args = [0] * 16
        args[0] = zp.tag
        args[8:] = zp.digest
        ptr = real.top.execute_by_name("ingress", args, q)
        back = real.top.execute_by_name("egress", list(ptr), q)
        assert list(back) == [zp.tag] + list(zp.digest), code
