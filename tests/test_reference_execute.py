"""The literal vectors of the reference's interpreter tests that were not transcribed before
(/root/reference/src/lair/execute.rs:837-1020: `lair_div_test`, `lair_shadow_test`, `lair_preimg_test`, `lair_array_test`,
`consistent_clean`, `nonpartial_calls_partial`) -- build container only: the `func!` bodies are read from the reference at run time
(tools/lurk_reference.py) and stored nowhere; what is written down here is data: arguments and expected outputs, with the line they
stand on.  Each program goes through the product's Lair compiler and interpreter and through the oracle's."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import lurk_reference as lr  # noqa: E402

pytestmark = pytest.mark.skipif(not lr.available(), reason="/root/reference is not on this box")

P = 2013265921


def _test_funcs(test_name):
    """the func! bodies of one #[test] function of src/lair/execute.rs, in the order of its Toplevel::new_pure(&[..])"""
    src = lr._strip_comments(lr._read("src/lair/execute.rs"))
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
    funcs = {}
    for fm in re.finditer(r"let (\w+) = func!\(", body):
        end = lr._balanced(body, fm.end() - 1)
        funcs[fm.group(1)] = body[fm.end():end - 1].strip()
    order = re.search(r"new_pure\(&\[([^\]]+)\]\)", body).group(1)
    return [funcs[n.strip()] for n in order.split(",")]


def _both(source):
    """(execute on the product, execute on the oracle), each name, args -> outputs on a fresh record per toplevel"""
    from lurk_amd import lair
    from oracle import lair as ol

    top = lair.Toplevel(source)
    q = lair.QueryRecord(top)
    otop = ol.Toplevel(source)
    oq = ol.QueryRecord(otop)
    return (lambda n, a: list(top.execute_by_name(n, a, q)), top, q), (lambda n, a: list(ol.execute(otop, n, a, oq)), otop, oq)


# (test, function, arguments, expected outputs): src/lair/execute.rs:848-849, 865-867, 896-903, 935-946
VECTORS = [
    ("lair_div_test", "test", [20, 4], [5]),
    ("lair_shadow_test", "test", [10], [80]),
    ("lair_preimg_test", "polynomial", [1, 3, 5, 7, 20], [58061]),
    ("lair_array_test", "test1", [1, 2, 3, 4, 5, 6, 7], [5, 7, 9]),
    ("lair_array_test", "test3", [4, 9, 21, 10], [1, 2, 3, 4]),
]


@pytest.mark.parametrize("test,func,args,want", VECTORS, ids=[f"{t}:{f}" for t, f, _, _ in VECTORS])
def test_execute_vectors(test, func, args, want):
    source = "\n".join(_test_funcs(test))
    (run, _, _), (orun, _, _) = _both(source)
    assert run(func, args) == want
    assert orun(func, args) == want


def test_preimage_of_an_invertible_function_comes_back():
    """execute.rs:871-905: polynomial(1, 3, 5, 7; 20) = 58061, and `inverse`, which asks for the preimage, returns the arguments --
    on the record that saw the forward query."""
    source = "\n".join(_test_funcs("lair_preimg_test"))
    for run, _, _ in _both(source):
        assert run("polynomial", [1, 3, 5, 7, 20]) == [58061]
        assert run("inverse", [58061]) == [1, 3, 5, 7, 20]


def test_injected_inverse_query_survives_clean():
    """execute.rs:950-1003 (`consistent_clean`): the preimage of double(1) = 2 is injected, `half(2)` finds it, and after
    `clean` the record still answers the same (the injected inverse is kept, the queries of the run are not)."""
    from lurk_amd import lair

    source = "\n".join(_test_funcs("consistent_clean"))
    top = lair.Toplevel(source)
    q = lair.QueryRecord(top)
    q.inject_inv_query(top.func_index("double"), [1], [2])
    res1 = list(top.execute_by_name("half", [2], q))
    assert res1 == [1]
    n1 = (q.num_func_queries(top.func_index("half")), q.num_func_queries(top.func_index("double")))
    for _ in range(2):
        q.clean()
        assert list(top.execute_by_name("half", [2], q)) == res1
        assert (q.num_func_queries(top.func_index("half")), q.num_func_queries(top.func_index("double"))) == n1


def test_total_function_may_not_call_a_partial_one():
    """execute.rs:1005-1025 (`nonpartial_calls_partial`, should_panic "assertion failed: ctx.partial"): both compilers refuse."""
    from lurk_amd import lair
    from oracle import lair as ol

    source = "\n".join(_test_funcs("nonpartial_calls_partial"))
    with pytest.raises(Exception):
        lair.Toplevel(source)
    with pytest.raises(Exception):
        ol.Toplevel(source)
