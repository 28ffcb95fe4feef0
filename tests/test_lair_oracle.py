"""Pins the independent Python restatement of Lair (oracle/lair.py) against the reference's literal
golden traces and layout sizes."""
import pytest

from lair_helpers import load_cases
from oracle import lair as ol


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_golden_trace(case):
    top = ol.Toplevel(case["source"])
    q = ol.QueryRecord(top)
    for name, args in case["calls"]:
        ol.execute(top, name, args, q)
    rows, width = ol.generate_trace(top, case["func"], q)
    assert width == case["width"]
    flat = [v for r in rows for v in r]
    assert flat == case["trace"]
    if case["layout"]:
        lay = top.layout(top.funcs[top.index[case["func"]]])
        assert lay == case["layout"]
    if case["mem"]:
        mt = ol.mem_trace(q, case["mem"]["len"])
        assert [v for r in mt for v in r] == case["mem"]["trace"]


def test_oracle_execute_known_answers():
    from lair_helpers import load_cases

    demo = load_cases()[0]["source"]
    top = ol.Toplevel(demo)
    q = ol.QueryRecord(top)
    assert ol.execute(top, "factorial", [5], q) == [120]   # src/lair/execute.rs:808-812
    assert ol.execute(top, "even", [7], q) == [0]           # execute.rs:814-817
    assert ol.execute(top, "odd", [4], q) == [0]            # execute.rs:819-822
