"""The compiled-toplevel exchange of the C ABI (lurkhip_toplevel_from_bytecode / lurkhip_toplevel_to_bytecode, format "LBC1",
lurk_amd/csrc/lair/bytecode_io.cpp): the flat u32 form of /root/reference/src/lair/bytecode.rs:12-146 a host with its own
compiler (the reference's Toplevel::new) hands over instead of source text.

* the product's compiler (C++) and the oracle's (Python) are independent; their exports must agree word for word;
* export -> import -> export is the identity, and the imported toplevel has the same layouts, the same AIR programs and the
  same executions as the one compiled from source (the GPU side of this is tests/test_bytecode_gpu.py);
* malformed blobs are errors, never crashes (the importer is a trust boundary of the library).
No GPU here."""
import ctypes as C

import numpy as np
import pytest

from lair_helpers import PARTIAL_SRC, U64_SRC, load_cases
from lurk_amd import _native as N
from lurk_amd import lair
from lurk_amd.air import ChipAir
from lurk_amd.programs import lurk_mix as lm
from lurk_amd.programs import synth_eval as se
from oracle import lair as ol


def _programs():
    out = [(c["name"], c["source"], c["lurk_chips"], c["calls"]) for c in load_cases()]
    out.append(("partial", PARTIAL_SRC, False, [("top", [9])]))
    out.append(("u64", U64_SRC, True, [("chain", [1, 2, 3, 4, 5, 6, 7, 8])]))
    out.append(("synth_eval", se.SOURCE, False, [(se.FUNC, se.args_for_rows(50))]))
    # arms of an array match written in DESCENDING key order: the reference numbers the return selectors in source order and
    # then sorts the arms by key (toplevel.rs:557-570, map.rs:16-22), so the stored blocks carry selectors 2, 1, 0
    out.append(("match_many_descending", """
fn t(a: [2]): [1] {
    match a {
        [2, 0] => {
            let x = 20;
            return x
        }
        [1, 0] => {
            let x = 10;
            return x
        }
        [0, 1] => {
            let x = 1;
            return x
        }
    };
    let z = 0;
    return z
}
""", False, [("t", [2, 0]), ("t", [0, 1]), ("t", [1, 0]), ("t", [5, 5])]))
    fm, mm = lm.fib_mix(520), lm.lurk_mix(700)
    out.append(("fib-mix", fm.source, True, [(fm.entry, fm.main_args)]))
    out.append(("lurk-mix", mm.source, True, [(mm.entry, mm.main_args)]))
    return out


PROGRAMS = _programs()


def _air_words(air):
    words = []
    for which in range(5):
        idx = 0
        while True:
            n = N.lib.lurkhip_air_program(air.handle, which, idx, None, 0)
            if n < 0:
                break
            buf = np.zeros(max(n, 1), dtype=np.uint32)
            assert N.lib.lurkhip_air_program(air.handle, which, idx, buf.ctypes.data_as(C.c_void_p), n) == n
            words.append((which, idx, buf[:n].tobytes()))
            idx += 1
            if which < 2:
                break
    return words


@pytest.mark.parametrize("name,source,chips,calls", PROGRAMS, ids=[p[0] for p in PROGRAMS])
def test_two_compilers_agree_and_round_trip(name, source, chips, calls):
    top = lair.Toplevel(source, lurk_chips=chips)
    blob = top.to_bytecode()
    otop = ol.Toplevel(source, chips=ol.lurk_chips() if chips else ())
    assert blob.tolist() == ol.to_bytecode(otop), "the product's compiler and the oracle's disagree"
    imp = lair.Toplevel.from_bytecode(blob)
    assert np.array_equal(imp.to_bytecode(), blob)
    assert imp.num_funcs() == top.num_funcs()
    for i in range(top.num_funcs()):
        assert imp.func_info(i) == top.func_info(i)
        a, b = ChipAir.for_func(top, i), ChipAir.for_func(imp, i)
        assert (a.name, a.width, a.num_constraints, a.num_sends, a.num_receives, a.permutation_width) == (
            b.name, b.width, b.num_constraints, b.num_sends, b.num_receives, b.permutation_width)
        assert _air_words(a) == _air_words(b), f"AIR programs of func {i} differ"
    q, qi = lair.QueryRecord(top), lair.QueryRecord(imp)
    for fname, args in calls:
        assert imp.func_index(fname) == top.func_index(fname)
        assert imp.execute_by_name(fname, args, qi) == top.execute_by_name(fname, args, q)
    for i in range(top.num_funcs()):
        assert qi.num_func_queries(i) == q.num_func_queries(i)
    assert qi.expect_public_values() == q.expect_public_values()


def _import_status(words):
    h = C.c_void_p()
    arr = np.ascontiguousarray(words, dtype=np.uint32)
    st = N.lib.lurkhip_toplevel_from_bytecode(arr.ctypes.data_as(C.c_void_p), arr.size, C.byref(h))
    if st == N.OK:
        N.lib.lurkhip_toplevel_free(h)
    return st


def test_truncated_blobs_are_errors():
    blob = lair.Toplevel(PARTIAL_SRC).to_bytecode()
    assert _import_status(blob) == N.OK
    for n in range(len(blob)):
        assert _import_status(blob[:n] if n else np.zeros(0, dtype=np.uint32)) != N.OK, n
    assert _import_status(np.concatenate([blob, [0]])) != N.OK  # trailing words
    with pytest.raises(lair.LairError, match="magic"):
        lair.Toplevel.from_bytecode(np.concatenate([[7], blob[1:]]))


def test_corrupted_blobs_never_crash():
    """Every single-word corruption either is rejected or yields a toplevel that passed validation (and can be exported)."""
    src = load_cases()[0]["source"]
    blob = lair.Toplevel(src).to_bytecode()
    rng = np.random.default_rng(7)
    rejected = 0
    for pos in range(2, len(blob)):
        for val in (0, 1, 0xFFFFFFFF, int(blob[pos]) + 1, int(rng.integers(0, 1 << 32))):
            if val == int(blob[pos]):
                continue
            bad = blob.copy()
            bad[pos] = val & 0xFFFFFFFF
            h = C.c_void_p()
            st = N.lib.lurkhip_toplevel_from_bytecode(bad.ctypes.data_as(C.c_void_p), bad.size, C.byref(h))
            if st != N.OK:
                rejected += 1
                continue
            assert N.lib.lurkhip_toplevel_to_bytecode(h, None, 0) > 0
            N.lib.lurkhip_toplevel_free(h)
    assert rejected > len(blob)  # most corruptions break a reference, a count or a selector number


def test_semantic_checks():
    blob = lair.Toplevel(PARTIAL_SRC).to_bytecode().tolist()
    otop = ol.Toplevel(PARTIAL_SRC)
    # a stack reference above the stack: `sub(n, one)` of pfib reads slot 1; point it far away
    words = ol.to_bytecode(otop)
    assert words == blob
    f = otop.funcs[0]
    f["body"]["ops"][0] = ("const", 5)  # harmless: still valid
    assert _import_status(ol.to_bytecode(otop)) == N.OK
    f["body"]["ops"].append(("add", 0, 99))
    with pytest.raises(lair.LairError, match="stack reference"):
        lair.Toplevel.from_bytecode(ol.to_bytecode(otop))
    # a chip this library does not know
    otop2 = ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
    otop2.chips[0].name = "no_such_chip"
    with pytest.raises(lair.LairError, match="no native chip"):
        lair.Toplevel.from_bytecode(ol.to_bytecode(otop2))


def test_semantic_checks_of_the_trust_boundary():
    """the importer must refuse what the interpreter / layout pass would later throw on or index with --
    Load / Store lengths without a memory table, preimages of non-invertible callees, non-canonical Choose keys."""
    import copy

    base = ol.Toplevel(PARTIAL_SRC)

    def variant(edit):
        t = copy.deepcopy(base)
        edit(t)
        return ol.to_bytecode(t)

    assert _import_status(variant(lambda t: None)) == N.OK
    # Store of 7 values / Load of 7: there are tables for 2, 3, 4, 5, 6, 8 only (/root/reference/src/lair/execute.rs:243-257)
    with pytest.raises(lair.LairError, match="memory tables"):
        lair.Toplevel.from_bytecode(variant(lambda t: t.funcs[0]["body"]["ops"].insert(0, ("store", [0] * 7))))
    with pytest.raises(lair.LairError, match="memory tables"):
        lair.Toplevel.from_bytecode(variant(lambda t: t.funcs[0]["body"]["ops"].insert(0, ("load", 7, 0))))
    with pytest.raises(lair.LairError, match="memory tables"):
        lair.Toplevel.from_bytecode(variant(lambda t: t.funcs[0]["body"]["ops"].insert(0, ("load", 1, 0))))
    # PreImg of a callee that is not invertible
    def preimg(t):
        g = next(i for i, f in enumerate(t.funcs) if not f["invertible"])
        t.funcs[0]["body"]["ops"].insert(0, ("preimg", g, [0] * t.funcs[g]["output_size"]))
    with pytest.raises(lair.LairError, match="not invertible"):
        lair.Toplevel.from_bytecode(variant(preimg))
    # a Choose key that is not a canonical field element: patch the key word of a one-variable match in an exported blob
    src = "fn f(x): [1] {\n    match x {\n        77777 => {\n            let a = 3;\n            return a\n        }\n    };\n    return x\n}\n"
    blob = lair.Toplevel(src).to_bytecode()
    at = [i for i, w in enumerate(blob.tolist()) if w == 77777]
    assert len(at) >= 1 and _import_status(blob) == N.OK  # (the default branch re-states the key as a Const for its AssertNe)
    messages = []
    for i in at:
        bad = blob.copy()
        bad[i] = 2013265921 + 5
        with pytest.raises(lair.LairError, match="canonical") as e:
            lair.Toplevel.from_bytecode(bad)
        messages.append(str(e.value))
    assert any("match key" in m for m in messages), messages
