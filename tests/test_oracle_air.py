"""The oracle's AIR restatement against the property the reference's own tests check
(/root/reference/src/air/debug.rs:119-206, used by /root/reference/src/core/tests/mod.rs:28-74 and the
lair tests): on traces of a real execution every constraint of every chip vanishes on every row and the
send / receive multisets of the whole machine balance."""
import pytest

from lair_helpers import PARTIAL_SRC, load_cases
from oracle import air as oa
from oracle import lair as ol


def machine(top, q, entry, with_bytes):
    f = top.funcs[top.index[entry]]
    pv = q.public_values
    chips = [(oa.EntrypointAir(f["index"], len(pv)), [list(pv)], None)]
    for g in top.funcs:
        rows, _ = ol.generate_trace(top, g["name"], q)
        if rows:
            chips.append((oa.FuncAir(top, g["name"]), rows, None))
    for ml in ol.MEM_TABLE_SIZES:
        chips.append((oa.MemAir(ml), ol.mem_trace(q, ml), None))
    if with_bytes:
        prep = [[i & 0xFF, i >> 8, int((i & 0xFF) < (i >> 8)), (i & 0xFF) & (i >> 8), (i & 0xFF) ^ (i >> 8), (i & 0xFF) | (i >> 8)] for i in range(1 << 16)]
        chips.append((oa.BytesAir(), ol.bytes_trace(q), prep))
    return chips, pv


@pytest.mark.parametrize("entry,args", [("factorial", [5]), ("fib", [7]), ("even", [6])])
def test_demo_machine_constraints_and_lookups(entry, args):
    top = ol.Toplevel(load_cases()[0]["source"])
    q = ol.QueryRecord(top)
    ol.execute(top, entry, args, q)
    chips, pv = machine(top, q, entry, with_bytes=False)
    assert oa.debug_check(chips, public=pv) > 0


def test_memory_case_constraints_and_lookups():
    case = next(c for c in load_cases() if c["mem"])
    top = ol.Toplevel(case["source"])
    q = ol.QueryRecord(top)
    name, args = case["calls"][0]
    ol.execute(top, name, args, q)
    chips, pv = machine(top, q, name, with_bytes=False)
    assert oa.debug_check(chips, public=pv) > 0


def test_partial_machine_with_byte_lookups():
    top = ol.Toplevel(PARTIAL_SRC)
    q = ol.QueryRecord(top)
    ol.execute(top, "top", [7], q)
    chips, pv = machine(top, q, "top", with_bytes=True)
    assert oa.debug_check(chips, public=pv) > 0


def test_broken_trace_is_rejected():
    top = ol.Toplevel(load_cases()[0]["source"])
    q = ol.QueryRecord(top)
    ol.execute(top, "factorial", [5], q)
    chips, pv = machine(top, q, "factorial", with_bytes=False)
    air, rows, prep = chips[1]
    rows[2][3] = (rows[2][3] + 1) % ol.P
    with pytest.raises(AssertionError):
        oa.debug_check(chips, public=pv)


def test_extern_chip_machine_constraints_and_lookups(oracle):
    """Poseidon2 wide AIR (hash3/4) and the u64 gadgets (add, sub, lessthan, iszero + byte lookups) on real traces:
    the property the reference checks in its own chip tests (e.g. /root/reference/src/core/u64.rs:233-300)."""
    from lair_helpers import U64_SRC
    from test_lair_gpu import oracle_chip_callbacks

    poseidon, witness = oracle_chip_callbacks(oracle)

    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    for entry, args in (("u64_ops", u64(0x0102030405060708) + u64(0x01020304FF060708)), ("chain", [9, 8, 7, 6, 5, 4, 3, 2])):
        top = ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
        q = ol.QueryRecord(top)
        ol.execute(top, entry, args, q, poseidon=poseidon)
        f = top.funcs[top.index[entry]]
        pv = q.public_values
        chips = [(oa.EntrypointAir(f["index"], len(pv)), [list(pv)], None)]
        for g in top.funcs:
            rows, _ = ol.generate_trace(top, g["name"], q, witness=witness)
            if rows:
                chips.append((oa.FuncAir(top, g["name"]), rows, None))
        for ml in ol.MEM_TABLE_SIZES:
            chips.append((oa.MemAir(ml), ol.mem_trace(q, ml), None))
        if q.bytes:
            prep = [[i & 0xFF, i >> 8, int((i & 0xFF) < (i >> 8)), (i & 0xFF) & (i >> 8), (i & 0xFF) ^ (i >> 8), (i & 0xFF) | (i >> 8)] for i in range(1 << 16)]
            chips.append((oa.BytesAir(), ol.bytes_trace(q), prep))
        assert oa.debug_check(chips, public=pv) > 0


def test_mul_divrem_bignum_machine_constraints_and_lookups(oracle):
    """u64 mul / divrem and the big-num comparison: witnesses from the oracle interpreter satisfy the oracle AIR and
    the byte lookups balance (the reference's chip tests do the same, /root/reference/src/core/u64.rs:302-420,
    /root/reference/src/core/big_num.rs:111-181)."""
    from lair_helpers import U64_SRC
    from test_lair_gpu import oracle_chip_callbacks

    poseidon, witness = oracle_chip_callbacks(oracle)

    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    cases = [("u64_more", u64(0xFEDCBA9876543210) + u64(0x1234567)), ("u64_more", u64(77) + u64(77)), ("u64_more", u64(5) + u64(2**63)),
             ("big_lt", [1, 2, 3, 4, 5, 6, 7, 8] + [1, 2, 3, 4, 5, 6, 9, 8]), ("big_lt", [9] * 8 + [9] * 8),
             ("big_lt", [0, 0, 0, 0, 0, 0, 0, 2013265920] + [0, 0, 0, 0, 0, 0, 0, 5])]
    for entry, args in cases:
        top = ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
        q = ol.QueryRecord(top)
        ol.execute(top, entry, args, q, poseidon=poseidon)
        f = top.funcs[top.index[entry]]
        pv = q.public_values
        chips = [(oa.EntrypointAir(f["index"], len(pv)), [list(pv)], None)]
        for g in top.funcs:
            rows, _ = ol.generate_trace(top, g["name"], q, witness=witness)
            if rows:
                chips.append((oa.FuncAir(top, g["name"]), rows, None))
        for ml in ol.MEM_TABLE_SIZES:
            chips.append((oa.MemAir(ml), ol.mem_trace(q, ml), None))
        prep = [[i & 0xFF, i >> 8, int((i & 0xFF) < (i >> 8)), (i & 0xFF) & (i >> 8), (i & 0xFF) ^ (i >> 8), (i & 0xFF) | (i >> 8)] for i in range(1 << 16)]
        chips.append((oa.BytesAir(), ol.bytes_trace(q), prep))
        assert oa.debug_check(chips, public=pv) > 0, (entry, args)
