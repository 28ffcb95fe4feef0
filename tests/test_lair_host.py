"""CPU tests of the product's host-side Lair (parser, compiler, layout, interpreter) through the C ABI.
No GPU: nothing here generates a trace."""
import pytest

from lair_helpers import PARTIAL_SRC, SHADOW_CALLS, SHADOW_SRC, U64_SRC, load_cases
from lurk_amd import lair
from oracle import lair as ol


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_layout_and_execution_match_golden_and_oracle(case):
    top = lair.Toplevel(case["source"], lurk_chips=case["lurk_chips"])
    otop = ol.Toplevel(case["source"])
    q, oq = lair.QueryRecord(top), ol.QueryRecord(otop)
    for name, args in case["calls"]:
        assert top.execute_by_name(name, args, q) == ol.execute(otop, name, args, oq)
    for i in range(top.num_funcs()):
        info = top.func_info(i)
        lay = otop.layout(otop.funcs[i])
        got = info["layout"]
        assert (got.nonce, got.input, got.output, got.aux, got.sel) == (1, lay["input"], lay["output"], lay["aux"], lay["sel"])
        assert q.num_func_queries(i) == len(oq.func[i])
    idx = top.func_index(case["func"])
    assert top.func_info(idx)["layout"].total() == case["width"]
    if case["layout"]:
        got = top.func_info(idx)["layout"]
        assert dict(nonce=got.nonce, input=got.input, aux=got.aux, sel=got.sel, output=got.output) == case["layout"]
    assert q.expect_public_values() == oq.public_values


def test_default_block_may_rebind_the_scrutinee():
    """Hand-computed answers; both compilers (the oracle's had the bug)."""
    top, otop = lair.Toplevel(SHADOW_SRC), ol.Toplevel(SHADOW_SRC)
    q, oq = lair.QueryRecord(top), ol.QueryRecord(otop)
    for (name, args), want in SHADOW_CALLS:
        assert top.execute_by_name(name, args, q) == want
        assert ol.execute(otop, name, args, oq) == want
    for i in range(top.num_funcs()):
        assert q.num_func_queries(i) == len(oq.func[i])


def test_iterative_interpreter_known_answers():
    demo = load_cases()[0]["source"]
    top = lair.Toplevel.new_pure(demo)
    q = lair.QueryRecord(top)
    # src/lair/execute.rs:826-834: fib(100000) mod p
    assert top.execute_by_name("fib", [100000], q) == [1123328132]
    assert q.num_func_queries(top.func_index("fib")) == 100001
    # default shard size 2^22 -> 1 shard; shard size 4 -> ceil(100001 / 4)
    assert len(lair.Shard.new(q).shard(lair.ShardingConfig())) == 1
    assert len(lair.Shard.new(q).shard(lair.ShardingConfig(4))) == 25001


def test_ackermann_sharding_count():
    # src/lair/trace.rs:655-692: A(3, n) = 2^(n+3) - 3; smaller n here to keep the CPU suite fast
    src = """
    fn ackermann(m, n): [1] {
        let one = 1;
        match m {
            0 => {
                let ret = add(n, one);
                return ret
            }
        };
        let m_minus_one = sub(m, one);
        match n {
            0 => {
                let ret = call(ackermann, m_minus_one, one);
                return ret
            }
        };
        let n_minus_one = sub(n, one);
        let inner = call(ackermann, m, n_minus_one);
        let ret = call(ackermann, m_minus_one, inner);
        return ret
    }
    """
    top = lair.Toplevel.new_pure(src)
    q = lair.QueryRecord(top)
    assert top.execute_by_name("ackermann", [3, 8], q) == [2**11 - 3]


def test_partial_functions_depth_and_public_values():
    top = lair.Toplevel(PARTIAL_SRC)
    otop = ol.Toplevel(PARTIAL_SRC)
    q, oq = lair.QueryRecord(top), ol.QueryRecord(otop)
    assert top.execute_by_name("top", [9], q) == ol.execute(otop, "top", [9], oq)
    pv = q.expect_public_values()
    assert pv == oq.public_values and len(pv) == 1 + 2 + 4  # partial: + 4 depth bytes (execute.rs:384-390)
    for i in range(3):
        got = top.func_info(i)["layout"]
        lay = otop.layout(otop.funcs[i])
        assert (got.aux, got.sel) == (lay["aux"], lay["sel"])


def test_lurk_chip_layout_widths():
    # SURVEY appendix B worked widths: hash3/4/5 = 493/655/815 (src/core/eval_direct.rs:2053-2055);
    top = lair.Toplevel(U64_SRC, lurk_chips=True)
    assert top.func_info(top.func_index("hash3"))["layout"].total() == 493
    assert top.func_info(top.func_index("hash4"))["layout"].total() == 655
    assert top.func_info(top.func_index("hash5"))["layout"].total() == 815


def test_errors_are_reported_not_thrown():
    with pytest.raises(lair.LairError) as e:
        lair.Toplevel("fn f(a): [1] { let b = call(nope, a); return b }")
    assert e.value.status == -7 and "Unknown function" in e.value.message
    with pytest.raises(lair.LairError):
        lair.Toplevel("fn f(a): [2] { return a }")  # return size mismatch (toplevel.rs:328-335)
    top = lair.Toplevel("fn f(a): [1] { let one = 1; assert_eq!(a, one); return a }")
    q = lair.QueryRecord(top)
    assert top.execute_by_name("f", [1], q) == [1]
    with pytest.raises(lair.LairError) as e:
        top.execute_by_name("f", [2], q)
    assert e.value.status == -6
    loop = lair.Toplevel("fn f(a): [1] { let b = call(f, a); return b }")
    with pytest.raises(lair.LairError) as e:
        loop.execute_by_name("f", [1], lair.QueryRecord(loop))
    assert "Loop detected" in e.value.message  # execute.rs:505-507


def test_air_programs_are_cut_into_interaction_pieces():
    """Host lowering (no GPU): the interaction program of a chip is also emitted as independent pieces at batch boundaries
    (one wave per piece in the prover kernels).  The pieces are compact: constant tuple elements are folded into per-interaction
    start values and runs of consecutive main columns are one instruction, so together they are much shorter than the whole
    program; small chips stay in one piece."""
    import numpy as np

    from lurk_amd import _native as N
    from lurk_amd import air
    from lurk_amd.context import _addr
    from lurk_amd.programs import synth_eval as se

    top = lair.Toplevel(se.SOURCE)

    def info(a):
        v = np.zeros(16, dtype=np.uint32)
        N.check(N.lib.lurkhip_air_info(a.handle, _addr(v)))
        return [int(x) for x in v]

    big = air.ChipAir.for_func(top, top.func_index(se.FUNC))
    i = info(big)
    n_inter = big.num_sends + big.num_receives
    assert n_inter == 46 and big.permutation_width == 24
    assert i[14] == 4  # a dozen interactions per piece
    assert i[13] == 480 and 46 * 3 <= i[15] <= (3 * i[13]) // 4  # whole program vs the compact pieces (>= begin + value + end each)
    small = air.ChipAir.for_mem(4)
    assert info(small)[14] == 1
    none = air.ChipAir.for_poseidon2(16)  # no lookups at all: one empty piece
    assert none.num_sends + none.num_receives == 0 and info(none)[14] == 1


def test_air_run_time_compiler_builds_without_a_device():
    """The run-time compiler of the AIR kernels (csrc/jit.cpp): source generation from the lowered program pieces + hiprtc against
    the embedded device headers needs no GPU -- a chip with lookups and one without compile to gfx950 code objects here."""
    import ctypes as C

    from lurk_amd import _native as N
    from lurk_amd import air

    top = lair.Toplevel(load_cases()[0]["source"])
    for a in (air.ChipAir.for_func(top, top.func_index("fib")), air.ChipAir.for_mem(4), air.ChipAir.for_poseidon2(16)):
        log = C.create_string_buffer(8192)
        size = N.lib.lurkhip_air_compile_check(a.handle, log, 8192)
        assert size > 1000, (a.name, log.value.decode(errors="replace"))


def test_trace_run_time_compiler_builds_without_a_device():
    """The generator of the per-function trace kernels (csrc/trace_jit.cpp): a function's micro-program unrolled along its block
    tree -- every variable an SSA value, every hint / require offset and aux column a literal -- compiles with hiprtc here; the
    program header carries the hash that names the compiled kernel."""
    import ctypes as C

    from lurk_amd import _native as N

    for case in load_cases()[:4]:
        top = lair.Toplevel(case["source"], lurk_chips=case["lurk_chips"])
        chip = lair.FuncChip.from_name(None, case["func"], top)
        src = chip.trace_kernel_source()
        assert "jit_row" in src and "jit_trace_staged" in src and "map[" not in src
        log = C.create_string_buffer(8192)
        size = N.lib.lurkhip_trace_compile_check(top.handle, chip.func_idx, log, 8192)
        assert size > 1000, (case["name"], log.value.decode(errors="replace"))
    # a match with several keys per arm and a default arm is one if / else-if chain
    top = lair.Toplevel(load_cases()[0]["source"])
    assert lair.FuncChip.from_name(None, "fib", top).trace_kernel_source().count("if (") >= 1


def test_memo_index_screens_and_deferred_seats_agree_with_the_oracle():
    """The query tables' Bloom screen and deferred index seats (lair.h: QueryMap, round 3): a key inserted a moment ago is
    looked up again before it has its seat (same store twice in a row, same call twice in a row), tables grow several times
    with entries waiting, and stores alternate between fresh and repeated tuples.  Pointers are outputs here, so a wrong
    index shows in the result; the query and memory counts must equal the oracle interpreter's."""
    src = """
    fn leaf(a, b): [2] {
        let p = store(a, b);
        let q = store(a, b);
        let d = sub(p, q);
        let (x, y) = load(p);
        let s = add(x, y);
        return (s, d)
    }
    fn walk(n, acc): [2] {
        if !n {
            let z = 0;
            return (acc, z)
        }
        let one = 1;
        let three = 3;
        let m = sub(n, one);
        let k = mul(n, three);
        let (s1, d1) = call(leaf, n, k);
        let (s2, d2) = call(leaf, n, k);
        let t = store(s1, s2, d1);
        let u = store(s1, s2, d1);
        let e = sub(t, u);
        let f = add(d1, d2);
        let g = add(e, f);
        let acc1 = add(acc, s1);
        let acc2 = add(acc1, g);
        let (r, w) = call(walk, m, acc2);
        let v = add(w, t);
        return (r, v)
    }
    """
    top, otop = lair.Toplevel.new_pure(src), ol.Toplevel(src)
    q, oq = lair.QueryRecord(top), ol.QueryRecord(otop)
    n = 5000  # 10 k table entries per kind: the index grows four times past its first 1024 slots
    got, want = top.execute_by_name("walk", [n, 7], q), ol.execute(otop, "walk", [n, 7], oq)
    assert got == want
    for i in range(top.num_funcs()):
        assert q.num_func_queries(i) == len(oq.func[i])
    for k, ml in enumerate(ol.MEM_TABLE_SIZES):
        assert q.num_mem_queries(ml) == len(oq.mem[k])
    assert q.num_mem_queries(2) == n and q.num_mem_queries(3) == n
