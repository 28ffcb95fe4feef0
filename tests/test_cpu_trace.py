"""CPU port of FuncChip trace generation (oracle/cpu_trace.c + oracle/cpu_trace.py: the `trace_all` stage of bench.py's
cpu_baseline) against the oracle's Python generator (oracle/lair.py: generate_trace, itself pinned by the reference's twelve
literal matrices): word for word, every function, sharded and not; and directly against the reference's literal matrices."""
import numpy as np
import pytest

from lair_helpers import PARTIAL_SRC, U64_SRC, load_cases
from oracle import cpu_trace as ct
from oracle import lair as ol


def u64(v):
    return [(v >> (8 * i)) & 0xFF for i in range(8)]


def poseidon_of(oracle):
    return lambda width, inp: [int(v) for v in oracle.p2_permute(width, np.array(inp, dtype=np.uint32))[0]]


def witness_of(oracle):
    from test_lair_gpu import oracle_chip_callbacks  # the Python witnesses the oracle's generator is given (no GPU involved)

    return oracle_chip_callbacks(oracle)[1]


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_reference_golden_traces(oracle, case):
    otop = ol.Toplevel(case["source"], chips=ol.lurk_chips() if case["lurk_chips"] else ())
    oq = ol.QueryRecord(otop)
    for name, args in case["calls"]:
        ol.execute(otop, name, args, oq, poseidon=poseidon_of(oracle))
    assert ct.generate_trace(otop, case["func"], oq).reshape(-1).tolist() == case["trace"]


def _all_funcs(oracle, src, calls, chips=(), shard_sizes=(1 << 22,)):
    otop = ol.Toplevel(src, chips=chips)
    oq = ol.QueryRecord(otop)
    for name, args in calls:
        ol.execute(otop, name, args, oq, poseidon=poseidon_of(oracle))
    n = 0
    for size in shard_sizes:
        most = max(len(oq.func[f["index"]]) for f in otop.funcs)
        for sidx in range(max(1, -(-most // size))):
            for f in otop.funcs:
                want, width = ol.generate_trace(otop, f["name"], oq, sidx, size, witness=witness_of(oracle))
                got = ct.generate_trace(otop, f["name"], oq, sidx, size)
                assert got.shape == (len(want), width) and got.tolist() == want, (f["name"], sidx, size)
                n += 1
    return n


def test_demo_and_partial_functions_sharded(oracle):
    demo = load_cases()[0]["source"]
    assert _all_funcs(oracle, demo, [["fib", [40]], ["factorial", [11]], ["even", [9]]], shard_sizes=(1 << 22, 4)) > 8
    assert _all_funcs(oracle, PARTIAL_SRC, [["top", [12]], ["top", [3]], ["pfib", [14]]], shard_sizes=(1 << 22, 4)) > 6


def test_extern_chips(oracle):
    calls = [["u64_ops", u64(5) + u64(7)], ["u64_ops", u64(2**64 - 1) + u64(1)], ["u64_ops", u64(3 << 40) + u64(3 << 32)],
             ["chain", [9, 8, 7, 6, 5, 4, 3, 2]], ["hash5", list(range(40))], ["u64_more", u64(0xFEDCBA9876543210) + u64(0x1234567)],
             ["u64_more", u64(77) + u64(77)], ["u64_more", u64(5) + u64(2**63)], ["big_lt", [1, 2, 3, 4, 5, 6, 7, 8] + [1, 2, 3, 4, 5, 6, 9, 8]],
             ["big_lt", [9] * 8 + [9] * 8], ["big_lt", [0, 0, 0, 0, 0, 0, 0, 2013265920] + [0, 0, 0, 0, 0, 0, 0, 5]]]
    _all_funcs(oracle, U64_SRC, calls, chips=ol.lurk_chips())


@pytest.mark.parametrize("workload,rows", [("fib-mix", 24), ("lurk-mix", 40)])
def test_mix_machines(oracle, workload, rows):
    from lurk_amd.programs import lurk_mix as lm

    mix = lm.fib_mix(rows) if workload == "fib-mix" else lm.lurk_mix(rows)
    _all_funcs(oracle, mix.source, [[mix.entry, list(mix.main_args)]], chips=ol.lurk_chips(), shard_sizes=(1 << 22, 16))
