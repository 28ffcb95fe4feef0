"""The generated fib-mix / lurk-mix machines (lurk_amd/programs/lurk_mix.py): every function has the width the reference's
own `test_widths` expects (/root/reference/src/core/eval_direct.rs:2025-2063) under BOTH layout implementations (the C++
host compiler and the oracle's independent Python one), the walkers produce exactly the dialled row counts, and a small
run satisfies the property the reference checks on its machines (every constraint vanishes on every row, lookups balance:
/root/reference/src/air/debug.rs:119-206)."""
import pytest

from lurk_amd import lair
from lurk_amd.programs import lurk_mix as lm
from oracle import air as oa
from oracle import lair as ol

# literal copy of the reference's expectations (eval_direct.rs:2025-2063), in its order
REF_WIDTHS = [97, 188, 10, 78, 148, 110, 81, 79, 97, 115, 78, 107, 70, 68, 72, 94, 66, 54, 66, 9, 50, 86, 58, 61, 114, 52, 104, 81, 493, 655,
              815, 53, 53, 85, 166, 44, 26, 38, 78]


def test_spec_table_is_the_reference_list():
    assert [lm.LURK_FUNCS[f][4] for f in lm.LURK_FUNC_ORDER] == REF_WIDTHS
    assert len(lm.LURK_FUNC_ORDER) == 39


@pytest.mark.parametrize("mix", [lm.fib_mix(520), lm.lurk_mix(700)], ids=["fib-mix", "lurk-mix"])
def test_widths_and_row_counts(mix):
    top = lair.Toplevel(mix.source, lurk_chips=True)
    otop = ol.Toplevel(mix.source, chips=ol.lurk_chips())
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    assert len(q.expect_public_values()) == 44  # 24 inputs + 16 outputs + 4 depth bytes (stark_machine.rs:16-17)
    for i in range(top.num_funcs()):
        name = otop.funcs[i]["name"]
        want = lm.LURK_FUNCS[name][4]
        assert top.func_info(i)["layout"].total() == want, name
        assert sum(otop.layout(otop.funcs[i]).values()) == want, name
        assert q.num_func_queries(i) == mix.rows[name], name
    assert q.num_func_queries(top.func_index("eval")) == mix.eval_rows
    # eval is the tallest chip, so it decides the number of shards (execute.rs:186-216)
    assert max(q.num_func_queries(i) for i in range(top.num_funcs())) == mix.eval_rows


@pytest.mark.parametrize("mix", [lm.fib_mix(40), lm.lurk_mix(60)], ids=["fib-mix", "lurk-mix"])
def test_machine_satisfies_the_reference_property(mix, oracle):
    from test_lair_gpu import oracle_chip_callbacks

    poseidon, witness = oracle_chip_callbacks(oracle)
    top = ol.Toplevel(mix.source, chips=ol.lurk_chips())
    q = ol.QueryRecord(top)
    ol.execute(top, mix.entry, mix.main_args, q, poseidon=poseidon)
    pv = q.public_values
    chips = [(oa.EntrypointAir(top.index[mix.entry], len(pv)), [list(pv)], None)]
    for g in top.funcs:
        rows, _ = ol.generate_trace(top, g["name"], q, witness=witness)
        if rows:
            chips.append((oa.FuncAir(top, g["name"]), rows, None))
    for ml in ol.MEM_TABLE_SIZES:
        chips.append((oa.MemAir(ml), ol.mem_trace(q, ml), None))
    prep = [[i & 0xFF, i >> 8, int((i & 0xFF) < (i >> 8)), (i & 0xFF) & (i >> 8), (i & 0xFF) ^ (i >> 8), (i & 0xFF) | (i >> 8)] for i in range(1 << 16)]
    chips.append((oa.BytesAir(), ol.bytes_trace(q), prep))
    assert oa.debug_check(chips, public=pv) > 0
