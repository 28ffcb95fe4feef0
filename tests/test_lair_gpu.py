"""GPU parity of trace generation (FuncChip / MemChip / BytesChip through the C ABI) against the reference's
golden traces and the independent Python oracle."""
import numpy as np
import pytest

from lair_helpers import PARTIAL_SRC, SHADOW_CALLS, SHADOW_SRC, U64_SRC, load_cases
from lurk_amd import field, lair
from oracle import lair as ol

pytestmark = pytest.mark.gpu
P = field.P


def oracle_chip_callbacks(oracle):
    def poseidon(width, inp):
        return [int(v) for v in oracle.p2_permute(width, np.array(inp, dtype=np.uint32))[0]]

    def le(v, n):
        return [(v >> (8 * i)) & 0xFF for i in range(n)]

    def u64(x):
        return sum(b << (8 * i) for i, b in enumerate(x))

    def witness(chip, inp):
        if chip.name.startswith("hasher"):
            w = chip.input_size
            x = np.array(inp, dtype=np.uint32)
            return [int(v) for v in oracle.p2_wide_witness(w, x)[0]], poseidon(w, inp)
        a, b = u64(inp[:8]), u64(inp[8:16]) if len(inp) >= 16 else 0
        if chip.name in ("u64_add", "u64_sub"):
            r = le((a + b if chip.name == "u64_add" else a - b) % (1 << 64), 8)
            return r, r
        if chip.name == "u64_lessthan":
            la, lb = le(a, 8), le(b, 8)
            wit = [0] * 12
            for i in reversed(range(8)):
                if la[i] != lb[i]:
                    wit[i] = 1
                    wit[8], wit[9] = la[i], lb[i]
                    wit[10] = pow((la[i] - lb[i]) % P, P - 2, P)
                    wit[11] = 1 if la[i] < lb[i] else 0
                    break
            return wit, [wit[11]]
        if chip.name == "u64_iszero":
            wit = [0] * 9
            for i, limb in enumerate(le(a, 8)):
                if limb:
                    wit[i] = pow(limb, P - 2, P)
                    break
            wit[8] = 1 if a == 0 else 0
            return wit, [wit[8]]
        if chip.name == "u64_mul":
            la, lb = le(a, 8), le(b, 8)
            carries, res, carry = [], [], 0
            for k in range(8):
                o = sum(la[i] * lb[k - i] for i in range(k + 1)) + carry
                res.append(o & 0xFF)
                carry = (o >> 8) & 0xFFFF
                carries.append(carry)
            return carries + res, res

        def msb_diff(x, y, n):
            lx, ly = le(x, n), le(y, n)
            for i in reversed(range(n)):
                if lx[i] != ly[i]:
                    return i, lx[i], ly[i]
            return -1, 0, 0

        def compare_witness(x, y, n):  # CompareWitness<_, n>: is_comp[n], lhs, rhs, diff_inv, is_less_than
            i, l, r = msb_diff(x, y, n)
            return [1 if k == i else 0 for k in range(n)] + [l, r, pow((l - r) % P, P - 2, P) if i >= 0 else 0, 1 if (i >= 0 and l < r) else 0]

        if chip.name == "u64_divrem":
            qv, rem = divmod(a, b)
            qb = qv * b
            wit = [0] * 8
            for i, limb in enumerate(le(b, 8)):
                if limb:
                    wit[i] = pow(limb, P - 2, P)
                    break
            wit += le(qv, 8)
            lq, lb = le(qv, 8), le(b, 8)
            carries, res, carry = [], [], 0
            for k in range(8):
                o = sum(lq[i] * lb[k - i] for i in range(k + 1)) + carry
                res.append(o & 0xFF)
                carry = (o >> 8) & 0xFFFF
                carries.append(carry)
            wit += carries + res + le(rem, 8)
            i, l, r = msb_diff(rem, b, 8)
            wit += [1 if k == i else 0 for k in range(8)] + [l, r]
            wit += compare_witness(qb, a, 8)
            return wit, le(qv, 8) + le(rem, 8)
        if chip.name == "big_num_lessthan":
            l = r = 0
            idx = -1
            for i in reversed(range(8)):
                if inp[i] != inp[8 + i]:
                    idx, l, r = i, inp[i], inp[8 + i]
                    break
            wit = [1 if k == idx else 0 for k in range(8)] + [l, r]
            for v in (l, r):
                wit += [1 if (v >> 24) < 0x78 else 0] + le(v, 4)
            cw = compare_witness(l, r, 4)
            return wit + cw, [cw[7]]
        raise NotImplementedError(chip.name)

    return poseidon, witness


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_golden_traces(ctx, case):
    top = lair.Toplevel(case["source"], lurk_chips=case["lurk_chips"])
    q = lair.QueryRecord(top)
    for name, args in case["calls"]:
        top.execute_by_name(name, args, q)
    chip = lair.FuncChip.from_name(ctx, case["func"], top)
    trace = chip.generate_trace(lair.Shard.new(q))
    assert trace.shape[1] == case["width"] == chip.width()
    assert trace.flatten().tolist() == case["trace"]
    # Montgomery output is the same matrix in the other encoding
    tm = chip.generate_trace(lair.Shard.new(q), repr=1)
    assert np.array_equal(field.from_monty(tm), trace)
    if case["mem"]:
        mt = lair.MemChip(ctx, case["mem"]["len"]).generate_trace(lair.Shard.new(q))
        assert mt.flatten().tolist() == case["mem"]["trace"]


def _compare_all_funcs(ctx, oracle, src, calls, lurk_chips=False, shard_sizes=(1 << 22,)):
    poseidon, witness = oracle_chip_callbacks(oracle)
    top = lair.Toplevel(src, lurk_chips=lurk_chips)
    otop = ol.Toplevel(src, chips=ol.lurk_chips() if lurk_chips else ())
    q, oq = lair.QueryRecord(top), ol.QueryRecord(otop)
    for name, args in calls:
        assert top.execute_by_name(name, args, q) == ol.execute(otop, name, args, oq, poseidon=poseidon)
    for size in shard_sizes:
        cfg = lair.ShardingConfig(size)
        shards = lair.Shard.new(q).shard(cfg)
        for sh in shards:
            for i, f in enumerate(otop.funcs):
                rows, width = ol.generate_trace(otop, f["name"], oq, sh.index, size, witness=witness)
                got = lair.FuncChip(ctx, i, top).generate_trace(sh)
                assert got.shape == (len(rows), width), (f["name"], sh.index)
                assert got.tolist() == rows, (f["name"], sh.index, size)
    for ml in lair.MEM_TABLE_SIZES:
        assert lair.MemChip(ctx, ml).generate_trace(lair.Shard.new(q)).tolist() == ol.mem_trace(oq, ml)
    for sidx in (0, 1):
        got = lair.BytesChip(ctx).generate_trace(lair.Shard(q, sidx))
        assert np.array_equal(got, np.array(ol.bytes_trace(oq, sidx), dtype=np.uint32))
    return top, q, oq


@pytest.mark.parametrize("workload,rows,shard", [("fib-mix", 40, 16), ("lurk-mix", 60, 32)])
def test_mix_machines_vs_oracle_interpreter_and_compiled(ctx, oracle, workload, rows, shard):
    """The bench machines themselves: every function of fib_mix(40) / lurk_mix(60) -- all 39 Lurk
    widths, partial functions with depth columns, u64 / hasher extern chips -- GPU trace == the oracle's independent trace
    generator, unsharded and sharded, first on the row interpreter, then on the compiled row kernels."""
    from lurk_amd.programs import lurk_mix as lm

    mix = lm.fib_mix(rows) if workload == "fib-mix" else lm.lurk_mix(rows)
    top, q, oq = _compare_all_funcs(ctx, oracle, mix.source, [[mix.entry, list(mix.main_args)]], lurk_chips=True, shard_sizes=(1 << 22, shard))
    poseidon, witness = oracle_chip_callbacks(oracle)
    otop = ol.Toplevel(mix.source, chips=ol.lurk_chips())
    assert q.num_func_queries(top.func_index("eval")) == rows
    for size in (1 << 22, shard):
        for sh in lair.Shard.new(q).shard(lair.ShardingConfig(size)):
            for i, f in enumerate(otop.funcs):
                chip = lair.FuncChip(ctx, i, top)
                chip.compile_trace()
                rows_, width = ol.generate_trace(otop, f["name"], oq, sh.index, size, witness=witness)
                got = chip.generate_trace(sh)
                assert got.shape == (len(rows_), width) and got.tolist() == rows_, (f["name"], sh.index, size, "compiled")


def test_demo_functions_vs_oracle_with_sharding(ctx, oracle):
    demo = load_cases()[0]["source"]
    # shard size 4 is what the reference's own tests use (src/core/tests/mod.rs:59-63)
    _compare_all_funcs(ctx, oracle, demo, [["fib", [40]], ["factorial", [11]], ["even", [9]]], shard_sizes=(1 << 22, 4))


def test_rebinding_default_vs_oracle(ctx, oracle):
    _compare_all_funcs(ctx, oracle, SHADOW_SRC, [[n, a] for (n, a), _ in SHADOW_CALLS], shard_sizes=(1 << 22, 2))


def test_partial_functions_vs_oracle(ctx, oracle):
    _compare_all_funcs(ctx, oracle, PARTIAL_SRC, [["top", [12]], ["top", [3]], ["pfib", [14]]], shard_sizes=(1 << 22, 4))


def test_extern_chips_vs_oracle(ctx, oracle):
    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    calls = [
        ["u64_ops", u64(5) + u64(7)],
        ["u64_ops", u64(2**64 - 1) + u64(1)],
        ["u64_ops", u64(0x0102030405060708) + u64(0x0102030405060708)],
        ["u64_ops", u64(3 << 40) + u64(3 << 32)],
        ["chain", [9, 8, 7, 6, 5, 4, 3, 2]],
        ["chain", [1, 0, 0, 0, 0, 0, 0, 0]],
        ["hash5", list(range(40))],
        ["u64_more", u64(0xFEDCBA9876543210) + u64(0x1234567)],
        ["u64_more", u64(77) + u64(77)],
        ["u64_more", u64(5) + u64(2**63)],
        ["big_lt", [1, 2, 3, 4, 5, 6, 7, 8] + [1, 2, 3, 4, 5, 6, 9, 8]],
        ["big_lt", [9] * 8 + [9] * 8],
        ["big_lt", [0, 0, 0, 0, 0, 0, 0, 2013265920] + [0, 0, 0, 0, 0, 0, 0, 5]],
    ]
    top, q, oq = _compare_all_funcs(ctx, oracle, U64_SRC, calls, lurk_chips=True)
    # byte lookups were recorded and provided
    assert q.num_byte_records() == len(oq.bytes) > 0


def test_bytes_preprocessed_trace(ctx):
    t = lair.BytesChip(ctx).generate_preprocessed_trace()
    i = np.arange(1 << 16)
    i1, i2 = i & 0xFF, i >> 8
    want = np.stack([i1, i2, (i1 < i2).astype(int), i1 & i2, i1 ^ i2, i1 | i2], axis=1).astype(np.uint32)
    assert np.array_equal(t, want)  # src/gadgets/bytes/trace.rs:49-72


def test_entrypoint_trace(ctx):
    top = lair.Toplevel(PARTIAL_SRC)
    q = lair.QueryRecord(top)
    top.execute_by_name("top", [6], q)
    t = lair.entrypoint_trace(q)
    assert t.shape == (1, 7)  # input 1 + output 2 + 4 depth bytes (lair_chip.rs:34-41)


def test_large_fib_trace_properties(ctx):
    """fib(100000): 100001 rows padded to 2^17.  Pure-Python oracle is too slow here, so check
    size-independent properties: known answer, nonce column, one-hot selectors, count_inv * (count+1) = 1,
    the recurrence out(n) = out(n-1) + out(n-2) across rows, and padding rows all zero."""
    demo = load_cases()[0]["source"]
    top = lair.Toplevel.new_pure(demo)
    q = lair.QueryRecord(top)
    assert top.execute_by_name("fib", [100000], q) == [1123328132]
    chip = lair.FuncChip.from_name(ctx, "fib", top)
    t = chip.generate_trace(lair.Shard.new(q)).astype(np.uint64)
    n = 100001
    assert t.shape == (1 << 17, 18)
    assert np.array_equal(t[:, 0], np.arange(1 << 17, dtype=np.uint64))
    assert not t[n:, 1:].any()
    real = t[:n]
    assert np.array_equal(real[:, 15:18].sum(axis=1), np.ones(n, dtype=np.uint64))
    assert np.array_equal(real[:, 1], np.arange(100000, -1, -1, dtype=np.uint64))  # args in call order
    # rows with the default selector: fib(n) = fib(n-1) + fib(n-2) and both requires are well formed
    d = real[real[:, 17] == 1]
    assert len(d) == 99999
    assert np.array_equal(d[:, 2], (d[:, 7] + d[:, 11]) % P)
    for c in (8, 12):
        assert np.array_equal(((d[:, c + 1] + 1) * d[:, c + 2]) % P, np.ones(len(d), dtype=np.uint64))
    # lookup chains (air/builder.rs:152-214): fib(k) is required by the rows of fib(k+1) and fib(k+2) (and fib(100000) by the
    # caller of the chip), each require holding the (nonce, count) the previous one left: per provided key the require records
    # are (0, 0) -> (nonce_a, 1) -> ..., and the row's provide record is the last link.  Row r holds the argument 100000 - r.
    arg_row = {int(a): r for r, a in enumerate(real[:, 1])}
    links = {}  # callee argument -> [(prev_nonce, prev_count, requiring row's nonce)]
    for row in d:
        for c, callee in ((8, int(row[1]) - 1), (12, int(row[1]) - 2)):
            links.setdefault(callee, []).append((int(row[c]), int(row[c + 1]), int(row[0])))
    for callee in (5, 4321, 99998):
        chain = sorted(links[callee], key=lambda x: x[1])
        assert [x[1] for x in chain] == list(range(len(chain)))          # counts 0, 1, ...
        assert chain[0][:2] == (0, 0)                                     # the first require starts the chain
        for prev, nxt in zip(chain, chain[1:]):
            assert nxt[0] == prev[2]                                      # prev_nonce = nonce of the row that required before
        prov = real[arg_row[callee]]
        assert (int(prov[3]), int(prov[4])) == (chain[-1][2], len(chain))  # provide record = the last link
    # inverse witnesses: n * (1/n) = 1 and (n-1) * 1/(n-1) = 1
    assert np.array_equal((d[:, 1] * d[:, 5]) % P, np.ones(len(d), dtype=np.uint64))
    assert np.array_equal(((d[:, 1] - 1) * d[:, 6]) % P, np.ones(len(d), dtype=np.uint64))


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_compiled_trace_kernels_golden(ctx, case, monkeypatch):
    """lurkhip_trace_compile: the function's micro-program as a straight-line row kernel (csrc/trace_jit.cpp).  The compiled
    kernel must give the reference's literal trace, like the interpreter (forced again by LURKHIP_TRACE_INTERPRET)."""
    top = lair.Toplevel(case["source"], lurk_chips=case["lurk_chips"])
    q = lair.QueryRecord(top)
    for name, args in case["calls"]:
        top.execute_by_name(name, args, q)
    chip = lair.FuncChip.from_name(ctx, case["func"], top)
    assert "jit_row" in chip.trace_kernel_source()
    chip.compile_trace()
    got = chip.generate_trace(lair.Shard.new(q))
    assert got.flatten().tolist() == case["trace"]
    monkeypatch.setenv("LURKHIP_TRACE_INTERPRET", "1")
    assert np.array_equal(chip.generate_trace(lair.Shard.new(q)), got)


@pytest.mark.parametrize("workload", ["fib-mix", "lurk-mix"])
def test_compiled_trace_kernels_every_function(ctx, workload, monkeypatch):
    """Every function of the bench machines (all 39 Lurk widths, partial functions, u64 / hasher extern chips, sharded):
    compiled row kernels == interpreter, bit for bit, Montgomery output."""
    from lurk_amd.programs import lurk_mix as lm

    mix = lm.fib_mix(1 << 9) if workload == "fib-mix" else lm.lurk_mix(1 << 9)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute(top.func_index(mix.entry), mix.main_args, q)
    shards = lair.Shard.new(q).shard(lair.ShardingConfig(1 << 8))
    chips = [lair.FuncChip(ctx, i, top) for i in range(top.num_funcs())]
    monkeypatch.setenv("LURKHIP_TRACE_INTERPRET", "1")
    want = [[c.generate_trace(sh, repr=1) for sh in shards] for c in chips]
    monkeypatch.delenv("LURKHIP_TRACE_INTERPRET")
    for c, w in zip(chips, want):
        c.compile_trace()
        for sh, t in zip(shards, w):
            assert np.array_equal(c.generate_trace(sh, repr=1), t), (c.func_idx, sh.index)


@pytest.mark.parametrize("seed,const_times_var", [(s, False) for s in (6, 9, 16, 22, 24, 29, 35, 41, 51, 52, 58, 62, 68)] + [(27, True), (52, True), (94, True)])
def test_random_programs_vs_oracle(ctx, oracle, seed, const_times_var):
    """Random programs (tests/lair_random.py: nested and array matches in any arm order, if / if !, every memory width,
    div / eq / not, asserts, preimages, partial functions): every function's GPU trace == the oracle's trace generator, unsharded
    and in shards of 3 rows, on the row interpreter and on the compiled row kernels; memory and byte chips too."""
    import lair_random as lr

    # const_times_var: products of a constant and a variable -- they have a trace (no aux column: func_chip.rs:202-211,
    # trace.rs:291-302) although the reference's AIR disagrees with its own layout about them (air.rs:345-358)
    src, calls, _ = lr.program(seed, const_times_var=const_times_var)
    top, q, oq = _compare_all_funcs(ctx, oracle, src, calls, shard_sizes=(1 << 22, 3))
    otop = ol.Toplevel(src)
    for size in (1 << 22, 3):
        for sh in lair.Shard.new(q).shard(lair.ShardingConfig(size)):
            for i, f in enumerate(otop.funcs):
                chip = lair.FuncChip(ctx, i, top)
                chip.compile_trace()
                rows_, width = ol.generate_trace(otop, f["name"], oq, sh.index, size)
                got = chip.generate_trace(sh)
                assert got.shape == (len(rows_), width) and got.tolist() == rows_, (f["name"], sh.index, size, "compiled")


def test_reference_proptest_regression_seeds(ctx, oracle):
    """The six failure cases the reference's property tests have found and keep re-running
    (/root/reference/proptest-regressions/gadgets/{unsigned/cmp,unsigned/div_rem,unsigned/field,comm/cmp}.txt), as explicit cases:
    compare a = b = 0 (twice), divide a = 0 by b = 1 (twice), a field element written as the modulus itself (2013265921 = 0: the
    boundary canonicalises it, so the case becomes its limb 0 -- beside it the limbs around BABYBEAR_MSB = 0x78 that the
    field-to-word witness of unsigned/field.rs:7-36 distinguishes), and the big-num comparison of all-zero limbs against a limb
    written as the modulus (equal after reduction: not less) with its neighbours 1 and p - 1."""
    P = 2013265921

    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    zero8 = [0] * 8
    calls = [
        ["u64_ops", u64(0) + u64(0)],                     # unsigned/cmp.txt: a = 0, b = 0 (both seeds)
        ["u64_more", u64(0) + u64(1)],                    # unsigned/div_rem.txt: a = 0, b = 1 (both seeds)
        ["big_lt", zero8 + [0, 0, 0, 0, 0, 0, P % P, 0]],  # comm/cmp.txt: rhs limb 6 = 2013265921 = 0
        ["big_lt", zero8 + [0, 0, 0, 0, 0, 0, 1, 0]],
        ["big_lt", zero8 + [0, 0, 0, 0, 0, 0, P - 1, 0]],
        ["big_lt", [0, 0, 0, 0, 0, 0, P - 1, 0] + zero8],
        ["big_lt", [P - 1] * 8 + [P - 1] * 8],            # unsigned/field.txt: around the modulus: msb byte 0x78 (p - 1 = 0x78000000) ...
        ["big_lt", [0x77FFFFFF] * 8 + [0x78000000] * 8],  # ... and just below it (msb byte 0x77)
        ["big_lt", [0x78000000] * 8 + [0x77FFFFFF] * 8],
    ]
    top, q, oq = _compare_all_funcs(ctx, oracle, U64_SRC, calls, lurk_chips=True)
    assert top.execute_by_name("u64_ops", u64(0) + u64(0), q)[16:] == [0, 1]      # not less, difference is zero
    assert top.execute_by_name("u64_more", u64(0) + u64(1), q) == [0] * 24         # 0 * 1, 0 / 1, 0 % 1
    assert top.execute_by_name("big_lt", zero8 + [0, 0, 0, 0, 0, 0, 0, 0], q) == [0]
    assert top.execute_by_name("big_lt", zero8 + [0, 0, 0, 0, 0, 0, 1, 0], q) == [1]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_extern_chips_on_random_and_edge_operands(ctx, oracle, seed):
    """The u64 / big-num / hasher chips on 60 random and edge operand pairs per seed (0, 1, 2^64 - 1, equal operands, divisor 1 and
    dividend < divisor, operands that differ in one byte only, field elements next to p): results, every function's trace, the
    memory and byte chips -- GPU == the oracle's generator."""
    import random

    rnd = random.Random(9000 + seed)
    P = 2013265921

    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    edge = [0, 1, 2, 255, 256, 2**32 - 1, 2**32, 2**63, 2**64 - 1, 2**64 - 2, 0x0101010101010101, 0xFF00FF00FF00FF00]

    def operand():
        k = rnd.random()
        if k < 0.35:
            return rnd.choice(edge)
        if k < 0.5:
            return rnd.getrandbits(rnd.choice([1, 8, 9, 16, 33, 63]))
        return rnd.getrandbits(64)

    calls = []
    for _ in range(20):
        a, b = operand(), operand()
        if rnd.random() < 0.15:
            b = a
        if rnd.random() < 0.15:
            b = a ^ (1 << (8 * rnd.randrange(8)))  # one differing byte
        calls.append(["u64_ops", u64(a) + u64(b)])
        calls.append(["u64_more", u64(a) + u64(b if b else 1)])
        x = [rnd.choice([0, 1, P - 1, P - 2, rnd.randrange(P)]) for _ in range(8)]
        y = list(x) if rnd.random() < 0.2 else [rnd.choice([0, 1, P - 1, rnd.randrange(P)]) for _ in range(8)]
        if rnd.random() < 0.3:
            y = list(x)
            y[rnd.randrange(8)] = rnd.randrange(P)
        calls.append(["big_lt", x + y])
    calls.append(["chain", [rnd.randrange(P) for _ in range(8)]])
    top, q, oq = _compare_all_funcs(ctx, oracle, U64_SRC, calls, lurk_chips=True)
    assert q.num_byte_records() == len(oq.bytes) > 0
