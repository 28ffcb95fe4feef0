"""The literal vectors of the reference's native-chip tests (/root/reference/src/core/u64.rs:243-600: `u64_add_test`, `u64_sub_test`,
`u64_mul_test`, `u64_divrem_test`, `u64_lessthan_test`, `u64_iszero_test`; /root/reference/src/core/big_num.rs:127-181:
`big_num_lessthan_test`) -- build container only: program, arguments and expected
outputs are all read from the reference at run time, nothing is written down here.  Each test's function goes through the product's
compiler and interpreter (native chips behind `extern_call`) and through the oracle's; the oracle's traces of the run then satisfy
its AIR and the byte lookups balance, which is what the reference's tests check after the value
(`debug_chip_constraints_and_queries_with_sharding`)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import lurk_reference as lr  # noqa: E402

pytestmark = pytest.mark.skipif(not lr.available(), reason="/root/reference is not on this box")

TESTS = [("src/core/u64.rs", t) for t in ("u64_add_test", "u64_sub_test", "u64_mul_test", "u64_divrem_test", "u64_lessthan_test", "u64_iszero_test")]
TESTS.append(("src/core/big_num.rs", "big_num_lessthan_test"))  # /root/reference/src/core/big_num.rs:127-181: the same kind of test for the big-num comparison


def _numbers(text):
    """the f(..) literals of an array: plain numbers or products of them (`f(16777216 * 2)`)"""
    out = []
    for x in re.findall(r"f\(([^()]+)\)", text):
        v = 1
        for part in x.split("*"):
            v *= int(part.strip())
        out.append(v)
    return out


def _case(path, test_name):
    """(func! source, function name, [args], [expected outputs]) of one test, read where it lies"""
    src = lr._strip_comments(lr._read(path))
    m = re.search(r"fn %s\(\) \{" % test_name, src)
    assert m, test_name
    depth, i = 0, m.end() - 1
    while True:
        depth += src[i] == "{"
        depth -= src[i] == "}"
        if depth == 0:
            break
        i += 1
    body = src[m.end():i]
    fm = re.search(r"func!\(", body)
    end = lr._balanced(body, fm.end() - 1)
    source = body[fm.end():end - 1].strip()
    name = re.search(r'execute_by_name\("(\w+)"', body).group(1)
    args = _numbers(re.search(r"let args = &\[(.*?)\];", body, re.S).group(1))
    want = _numbers(re.search(r"out\.as_ref\(\),\s*&\[(.*?)\]\s*\)", body, re.S).group(1))
    return source, name, args, want


@pytest.mark.parametrize("path,test_name", TESTS, ids=[t for _, t in TESTS])
def test_u64_chip_vectors(path, test_name, oracle):
    from lurk_amd import lair
    from oracle import air as oa
    from oracle import lair as ol
    from test_lair_gpu import oracle_chip_callbacks

    source, name, args, want = _case(path, test_name)
    assert args and want
    top = lair.Toplevel(source, lurk_chips=True)
    q = lair.QueryRecord(top)
    assert list(top.execute_by_name(name, args, q)) == want
    poseidon, witness = oracle_chip_callbacks(oracle)
    otop = ol.Toplevel(source, chips=ol.lurk_chips())
    oq = ol.QueryRecord(otop)
    assert list(ol.execute(otop, name, args, oq, poseidon=poseidon)) == want
    # ... and the vanish-and-balance check the reference makes on the run
    pv = oq.public_values
    f = otop.funcs[otop.index[name]]
    chips = [(oa.EntrypointAir(f["index"], len(pv)), [list(pv)], None)]
    for g in otop.funcs:
        rows, _ = ol.generate_trace(otop, g["name"], oq, witness=witness)
        if rows:
            chips.append((oa.FuncAir(otop, g["name"]), rows, None))
    for ml in ol.MEM_TABLE_SIZES:
        chips.append((oa.MemAir(ml), ol.mem_trace(oq, ml), None))
    prep = [[i & 0xFF, i >> 8, int((i & 0xFF) < (i >> 8)), (i & 0xFF) & (i >> 8), (i & 0xFF) ^ (i >> 8), (i & 0xFF) | (i >> 8)] for i in range(1 << 16)]
    chips.append((oa.BytesAir(), ol.bytes_trace(oq), prep))
    assert oa.debug_check(chips, public=pv) > 0
