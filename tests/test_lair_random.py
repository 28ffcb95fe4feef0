"""Differential test of the host Lair against the oracle's independent implementation on RANDOM programs (tests/lair_random.py):
same results, same number of queries per function and of cells per memory table, same byte-record count, same public values,
same column layout per function -- and the exported bytecode, imported again, behaves the same.  The GPU half (traces, word for
word) is tests/test_lair_gpu.py::test_random_programs_vs_oracle."""
import pytest

import lair_random as lr
from lurk_amd import lair
from oracle import lair as ol

SEEDS = list(range(120))


def _run(top, otop, calls):
    q, oq = lair.QueryRecord(top), ol.QueryRecord(otop)
    for name, args in calls:
        assert top.execute_by_name(name, args, q) == ol.execute(otop, name, args, oq), (name, args)
    return q, oq


@pytest.mark.parametrize("chunk", range(0, len(SEEDS), 20))
def test_random_programs_interpreter_and_layout_vs_oracle(chunk):
    total_queries = 0
    for seed in SEEDS[chunk:chunk + 20]:
        src, calls, _ = lr.program(seed)
        top, otop = lair.Toplevel.new_pure(src), ol.Toplevel(src)
        assert top.to_bytecode().tolist() == ol.to_bytecode(otop), (seed, "the product's compiler and the oracle's disagree")
        q, oq = _run(top, otop, calls)
        for i in range(top.num_funcs()):
            assert q.num_func_queries(i) == len(oq.func[i]), (seed, i)
            got, lay = top.func_info(i)["layout"], otop.layout(otop.funcs[i])
            assert (got.nonce, got.input, got.output, got.aux, got.sel) == (1, lay["input"], lay["output"], lay["aux"], lay["sel"]), (seed, i)
            total_queries += len(oq.func[i])
        for k, ml in enumerate(ol.MEM_TABLE_SIZES):
            assert q.num_mem_queries(ml) == len(oq.mem[k]), (seed, ml)
        assert q.num_byte_records() == len(oq.bytes), seed
        assert q.expect_public_values() == oq.public_values, seed
        # export -> import: the same machine
        again = lair.Toplevel.from_bytecode(top.to_bytecode())
        q2, _ = _run(again, otop, calls)
        assert [q2.num_func_queries(i) for i in range(top.num_funcs())] == [q.num_func_queries(i) for i in range(top.num_funcs())], seed
    assert total_queries > 100  # the chunk exercised something


def test_parser_and_compiler_survive_mutated_sources():
    """Token-level mutations of valid programs (dropped, duplicated, swapped and replaced tokens): the C++ parser / compiler
    answers with a program or an error status, never a crash; whatever still compiles round-trips through the bytecode.
    (tools/asan_host.sh runs this under AddressSanitizer + UBSan.)"""
    import random
    import re

    rnd = random.Random(0x4C55524B)
    vocab = ["{", "}", "(", ")", "[", "]", ";", ",", "=>", "=", "let", "match", "if", "!", "return", "call", "store", "load", "fn", "partial",
             "invertible", "preimg", "0", "1", "2013265921", "4294967295", "99999999999999999999", "x", "n", ":", "add", "mul", "assert_eq!", "range_u8!"]
    compiled = errors = 0
    for seed in range(40):
        src, _, _ = lr.program(seed)
        toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*!?|\d+|=>|\S", src)
        for _ in range(12):
            t = list(toks)
            for _ in range(rnd.randint(1, 3)):
                i = rnd.randrange(len(t))
                k = rnd.random()
                if k < 0.3:
                    del t[i]
                elif k < 0.5:
                    t.insert(i, t[i])
                elif k < 0.7:
                    j = rnd.randrange(len(t))
                    t[i], t[j] = t[j], t[i]
                else:
                    t[i] = rnd.choice(vocab)
            try:
                top = lair.Toplevel.new_pure(" ".join(t))
            except lair.LairError:
                errors += 1
                continue
            compiled += 1
            again = lair.Toplevel.from_bytecode(top.to_bytecode())
            assert again.num_funcs() == top.num_funcs()
    assert errors > 100 and compiled >= 3, (errors, compiled)
