"""End to end: the GPU shard prover's proofs are accepted by the oracle's verifier (which recomputes the whole
transcript, every Merkle opening, every FRI query and the constraint identity at zeta through the oracle's own
numeric AIR), and tampered proofs are rejected."""
import copy

import numpy as np
import pytest

import lurk_amd
from lair_helpers import PARTIAL_SRC, load_cases
from lurk_amd import lair, prover
from lurk_amd.programs import synth_eval as se
from oracle import air as oa
from oracle import binding as ob
from oracle import lair as ol
from oracle import stark as os_

pytestmark = pytest.mark.gpu
DEMO = load_cases()[0]["source"]
LOG_ROWS_LARGE = 20  # BASELINE.json configs[2]


def oracle_airs(src, entry, n_public):
    otop = ol.Toplevel(src)
    airs = [oa.EntrypointAir(otop.index[entry], n_public)]
    airs += [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    airs += [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES]
    airs.append(oa.BytesAir())
    return airs


def prove(ctx, src, entry, args, num_queries=8, pow_bits=6, shard_size=None):
    top = lair.Toplevel(src)
    q = lair.QueryRecord(top)
    top.execute_by_name(entry, args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, entry, len(pv))
    root = m.setup()
    proofs = m.prove(q, lair.ShardingConfig(shard_size) if shard_size else None, num_queries=num_queries, pow_bits=pow_bits)
    assert m.verify(proofs)  # the product's own host verifier (csrc/verify.cpp) accepts what the oracle's is about to be asked
    return m, root, proofs, pv


def verify(src, entry, root, proofs, n_public):
    return os_.verify_machine(oracle_airs(src, entry, n_public), root, [16], [6], proofs, ob.merkle_verify)


@pytest.mark.parametrize("src,entry,args", [(DEMO, "fib", [12]), (PARTIAL_SRC, "top", [8]), (se.SOURCE, "synth_eval", [1, 50, 0])],
                         ids=["demo_fib", "partial_with_bytes", "synth_eval"])
def test_gpu_proof_verifies(ctx, src, entry, args):
    m, root, proofs, pv = prove(ctx, src, entry, args)
    assert len(proofs) == 1
    assert verify(src, entry, root, proofs, len(pv))


def test_tampered_proofs_are_rejected(ctx):
    m, root, proofs, pv = prove(ctx, DEMO, "fib", [9])
    airs = oracle_airs(DEMO, "fib", len(pv))

    from proof_words import encode_words

    def check(mutate):
        bad = copy.deepcopy(proofs)
        mutate(bad[0])
        with pytest.raises(os_.VerifyError):
            os_.verify_machine(airs, root, [16], [6], bad, ob.merkle_verify)
        with pytest.raises(prover.VerificationError):  # ... and so does the product's verifier
            m.verify([encode_words(b) for b in bad])

    def bump_opened(p):
        loc, nxt = p.chips[1].opened["main"]
        loc[2] = ((loc[2][0] + 1) % os_.P,) + loc[2][1:]

    def bump_final(p):
        p.final_poly = ((p.final_poly[0] + 1) % os_.P,) + p.final_poly[1:]

    def bump_cumsum(p):
        cs = p.chips[0].cumulative_sum
        p.chips[0].cumulative_sum = ((cs[0] + 1) % os_.P,) + cs[1:]

    def bump_row(p):
        rw, recs = p.round_openings[1]
        recs[0][0] = (recs[0][0] + 1) % os_.P

    def bump_pow(p):
        p.pow_witness = (p.pow_witness + 1) % os_.P

    for f in (bump_opened, bump_final, bump_cumsum, bump_row, bump_pow):
        check(f)


def test_proof_of_a_wrong_trace_is_rejected(ctx):
    """A trace that violates a constraint still commits and opens consistently (every committed matrix is low degree by
    construction), but C(zeta) / Z_H(zeta) no longer equals the quotient recomputed from the opened chunks: the verifier
    rejects exactly there.  The device-side debug check finds the row."""
    top = lair.Toplevel(DEMO)
    q = lair.QueryRecord(top)
    top.execute_by_name("fib", [9], q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, "fib", len(pv))
    root = m.setup()
    traces = m.shard_traces(lair.Shard.new(q))
    mi, air, lg, t = traces[1]
    t[1, 2] += 1
    assert air.check_trace(ctx, 1 << lg, t)[0] in (0, 1)
    handle, main_root = m.commit_shard(traces)
    ch = prover.Challenger(ctx)
    ch.observe(m.vk_root)
    ch.observe([0])
    ch.observe(main_root)
    ch.observe(pv)
    proof = m.prove_shard(handle, ch, pv, num_queries=4, pow_bits=2)
    m.free_shard(handle)
    with pytest.raises(os_.VerifyError, match="do not match the quotient"):
        verify(DEMO, "fib", root, [proof], len(pv))
    with pytest.raises(prover.VerificationError, match="do not match the quotient"):
        m.verify([proof])


def test_sharded_proof_verifies(ctx):
    """max_shard_size 8: several independent shard proofs sharing one transcript prefix; the chips' cumulative
    sums only cancel across all shards (the reference's sharding test uses size 4, src/core/tests/mod.rs:59-63)."""
    m, root, proofs, pv = prove(ctx, DEMO, "fib", [20], shard_size=8)
    assert len(proofs) >= 3
    assert verify(DEMO, "fib", root, proofs, len(pv))


def test_openings_64bit_index_kernels(ctx, monkeypatch):
    """Committed matrices of 4 GiB and more reduce their openings with 64-bit word indices; the test hook takes those
    kernels at small sizes and the proof must not change."""
    _, root, want, pv = prove(ctx, DEMO, "fib", [20])
    monkeypatch.setenv("LURKHIP_OPENINGS_FORCE_64BIT", "1")
    _, _, got, _ = prove(ctx, DEMO, "fib", [20])
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a.words, b.words)
    assert verify(DEMO, "fib", root, got, len(pv))


def test_pipelined_sharded_proof_equals_sequential(ctx):
    """Two machines on two contexts (two HIP streams) of the one GPU prove the shards of an execution concurrently
    (prover.prove_pipelined): the proofs are those of Machine.prove, shard by shard, and verify."""
    top = lair.Toplevel(DEMO)
    q = lair.QueryRecord(top)
    top.execute_by_name("fib", [20], q)
    pv = q.expect_public_values()
    cfg = lair.ShardingConfig(8)
    m1 = prover.Machine(ctx, top, "fib", len(pv))
    root = m1.setup()
    want = m1.prove(q, cfg, num_queries=4, pow_bits=2)
    with lurk_amd.Context(0) as ctx2:
        m2 = prover.Machine(ctx2, top, "fib", len(pv))
        got = prover.prove_pipelined([m1, m2], q, cfg, num_queries=4, pow_bits=2)
        m2.close()
    m1.close()
    assert len(got) == len(want) >= 3
    for a, b in zip(got, want):
        assert np.array_equal(a.words, b.words)
    assert verify(DEMO, "fib", root, got, len(pv))


def test_compiled_air_kernels(ctx):
    """ChipAir.compile: the chip's program pieces as straight-line device code (hiprtc) instead of the interpreter.  The
    permutation traces must be bit-identical, and a proof made with every chip compiled is the interpreter's proof."""
    import torch

    from lurk_amd import air, field

    top = lair.Toplevel(DEMO)
    q = lair.QueryRecord(top)
    top.execute_by_name("fib", [12], q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, "fib", len(pv))
    root = m.setup()
    want = m.prove(q, num_queries=4, pow_bits=2)
    # permutation trace of one chip, before and after compiling it
    shard = lair.Shard.new(q)
    idx = top.func_index("fib")
    chip = lair.FuncChip(ctx, idx, top)
    t = chip.generate_trace(shard, repr=1)
    td = torch.from_numpy(t.view(np.int32)).cuda()
    a = air.ChipAir.for_func(top, idx)
    ch = [3, 1, 4, 1, 5, 9, 2, 6]
    out0 = torch.zeros((t.shape[0], 4 * a.permutation_width), dtype=torch.int32, device="cuda")
    a.permutation_trace(ctx, t.shape[0], td, None, ch, out0)
    a.compile(ctx)
    out1 = torch.zeros_like(out0)
    a.permutation_trace(ctx, t.shape[0], td, None, ch, out1)
    assert torch.equal(out0, out1)
    # the whole machine compiled
    for _, _, chip_air in m.chips:
        chip_air.compile(ctx)
    got = m.prove(q, num_queries=4, pow_bits=2)
    m.close()
    assert len(got) == len(want)
    for x, y in zip(got, want):
        assert np.array_equal(x.words, y.words)
    assert verify(DEMO, "fib", root, got, len(pv))


def test_machine_with_extern_chips_proves_and_verifies(ctx):
    """hash3 / hash4 chips (Poseidon2 wide AIR, 493 / 655 columns) called through call / preimg, with the hash queries
    injected the way the reference's setup does (inject_inv_queries, /root/reference/src/lair/execute.rs:299)."""
    from lair_helpers import U64_SRC

    top = lair.Toplevel(U64_SRC, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name("chain", [9, 8, 7, 6, 5, 4, 3, 2], q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, "chain", len(pv))
    root = m.setup()
    proofs = m.prove(q, num_queries=6, pow_bits=4)
    otop = ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
    airs = [oa.EntrypointAir(otop.index["chain"], len(pv))] + [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    airs += [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES] + [oa.BytesAir()]
    assert os_.verify_machine(airs, root, [16], [6], proofs, ob.merkle_verify)
    assert m.verify(proofs)


def test_u64_gadget_machine_proves_and_verifies(ctx):
    from lair_helpers import U64_SRC

    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    top = lair.Toplevel(U64_SRC, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name("u64_ops", u64(0x0102030405060708) + u64(0x01020304FF060708), q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, "u64_ops", len(pv))
    root = m.setup()
    proofs = m.prove(q, num_queries=6, pow_bits=4)
    otop = ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
    airs = [oa.EntrypointAir(otop.index["u64_ops"], len(pv))] + [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    airs += [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES] + [oa.BytesAir()]
    assert os_.verify_machine(airs, root, [16], [6], proofs, ob.merkle_verify)
    assert m.verify(proofs)


def test_large_shard_proof_verifies(ctx):
    """The bench workload itself: a 2^20-row eval shard (LDE height 2^22 for the callee chip) with the bench FRI parameters: the verifier's work is
    independent of the trace height except for the Merkle path lengths, so the full-size pipeline (multi-chunk scans,
    3-pass NTTs, k_top tails, 18 FRI layers) is checked end to end."""
    top = lair.Toplevel(se.SOURCE)
    q = lair.QueryRecord(top)
    top.execute_by_name(se.FUNC, se.args_for_rows(1 << LOG_ROWS_LARGE), q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, se.FUNC, len(pv))
    root = m.setup()
    proofs = m.prove(q, num_queries=100, pow_bits=16)
    assert proofs[0].log_max_height == LOG_ROWS_LARGE + 2
    assert verify(se.SOURCE, se.FUNC, root, proofs, len(pv))
    assert m.verify(proofs)


@pytest.mark.parametrize("entry,args", [("u64_more", [0x10, 0x32, 0x54, 0x76, 0x98, 0xBA, 0xDC, 0xFE, 0x67, 0x45, 0x23, 0x01, 0, 0, 0, 0]),
                                        ("big_lt", [1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 6, 9, 8])], ids=["mul_divrem", "bignum"])
def test_mul_divrem_bignum_machines_prove_and_verify(ctx, entry, args):
    from lair_helpers import U64_SRC

    top = lair.Toplevel(U64_SRC, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(entry, args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, entry, len(pv))
    root = m.setup()
    proofs = m.prove(q, num_queries=6, pow_bits=4)
    otop = ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
    airs = [oa.EntrypointAir(otop.index[entry], len(pv))] + [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    airs += [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES] + [oa.BytesAir()]
    assert os_.verify_machine(airs, root, [16], [6], proofs, ob.merkle_verify)
    assert m.verify(proofs)


def test_prover_edge_parameters_and_errors(ctx):
    """One query / no proof of work; invalid arguments come back as status codes, not crashes."""
    import ctypes as C

    from lurk_amd import _native as N

    m, root, proofs, pv = prove(ctx, DEMO, "factorial", [5], num_queries=1, pow_bits=0)
    assert proofs[0].pow_witness == 0 and len(proofs[0].query_indices) == 1
    assert verify(DEMO, "factorial", root, proofs, len(pv))
    # null / out-of-range arguments
    out = C.c_void_p()
    assert N.lib.lurkhip_shard_prove(ctx.handle, None, None, None, None, 0, 1, 0, C.byref(out)) == N.ERR_INVALID_ARG
    assert N.lib.lurkhip_shard_commit(ctx.handle, 0, None, None, None, None, 1, C.byref(out), None) == N.ERR_INVALID_ARG
    assert N.lib.lurkhip_air_mem(7, C.byref(out)) != N.OK  # there is no 7-wide memory table (execute.rs:243-244)
    top = lair.Toplevel(DEMO)
    assert N.lib.lurkhip_air_func(top.handle, 99, C.byref(out)) == N.ERR_INVALID_ARG
    ch = prover.Challenger(ctx)
    ch.observe([1, 2, 3])
    a = ch.clone()
    assert a.sample_ext() == ch.sample_ext()  # clones continue identically


@pytest.mark.parametrize("seed,shard_size", [(s, (None, 2, None, 3, None, 4)[s % 6]) for s in range(48)])
def test_random_machines_prove_and_verify(ctx, seed, shard_size):
    """Random programs (tests/lair_random.py) as whole machines: chips of 1 .. 64 rows with every kind of column (nested and
    array matches, every memory width, preimages, partial functions with depth columns and byte lookups), proved -- some of them
    in shards of 2 .. 4 rows -- and verified by the oracle from its own AIRs; the compiled AIR kernels produce the same proof."""
    import lair_random as lr

    src, calls, _ = lr.program(seed)
    entry, args = calls[0]
    m, root, proofs, pv = prove(ctx, src, entry, args, shard_size=shard_size)
    assert verify(src, entry, root, proofs, len(pv))
    assert prover.grand_sum(proofs) == (0, 0, 0, 0)
    if len(pv) >= 4 and all(v <= 255 for v in pv[-4:]):
        # public values that end in four bytes (every partial program's: its depth) fit the reference's CryptoProof, whose `depth`
        # field they must equal: prove -> bincode -> the product's verifier from the bytes alone
        from lurk_amd import proofs as lp

        data = lp.CryptoProof.from_machine_proof(m, proofs, verifier_version="r").to_bytes()
        assert lp.verify_crypto_proof(m, data, pv, num_queries=8, pow_bits=6)
        with pytest.raises(prover.VerificationError):
            lp.verify_crypto_proof(m, data[:100] + bytes([data[100] ^ 1]) + data[101:], pv, num_queries=8, pow_bits=6)


@pytest.mark.parametrize("seed", [6, 16, 41, 52, 62])
def test_first_proofs_of_fresh_contexts_are_identical(seed):
    """A proof made of short chips spreads them over the context's side streams (ctx.h: SideLane); tables a context builds at
    first use (selector tables, twiddles, programs) must be complete before any of those streams reads them: the FIRST proof of
    three fresh contexts is the same proof (a race on the selector tables of chips of one height showed up here)."""
    import lair_random as lr

    src, calls, _ = lr.program(seed)
    entry, args = calls[0]
    words = []
    for _ in range(3):
        with lurk_amd.Context(0) as c:
            m, root, proofs, pv = prove(c, src, entry, args)
            words.append([p.words.copy() for p in proofs])
            m.close()
    for other in words[1:]:
        assert len(other) == len(words[0]) and all(np.array_equal(a, b) for a, b in zip(other, words[0]))


@pytest.mark.parametrize("seed", range(6))
def test_u64_and_bignum_machines_on_random_operands_prove_and_verify(ctx, seed):
    """u64_ops / u64_more / big_lt machines on random and edge operands: proved, accepted by the product's verifier and the oracle's."""
    import random

    from lair_helpers import U64_SRC

    rnd = random.Random(77 + seed)
    P = 2013265921

    def u64(v):
        return [(v >> (8 * i)) & 0xFF for i in range(8)]

    edge = [0, 1, 255, 256, 2**32 - 1, 2**63, 2**64 - 1]
    a = rnd.choice(edge) if rnd.random() < 0.4 else rnd.getrandbits(64)
    b = a if rnd.random() < 0.2 else (rnd.choice(edge) if rnd.random() < 0.4 else rnd.getrandbits(rnd.choice([8, 33, 64])))
    x = [rnd.choice([0, P - 1, rnd.randrange(P)]) for _ in range(8)]
    y = list(x)
    y[rnd.randrange(8)] = rnd.randrange(P)
    entry, args = [("u64_ops", u64(a) + u64(b)), ("u64_more", u64(a) + u64(b or 3)), ("big_lt", x + y)][seed % 3]
    top = lair.Toplevel(U64_SRC, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(entry, args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, entry, len(pv))
    root = m.setup()
    proofs = m.prove(q, num_queries=6, pow_bits=4)
    assert m.verify(proofs)
    otop = ol.Toplevel(U64_SRC, chips=ol.lurk_chips())
    airs = [oa.EntrypointAir(otop.index[entry], len(pv))] + [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    airs += [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES] + [oa.BytesAir()]
    assert os_.verify_machine(airs, root, [16], [6], proofs, ob.merkle_verify)
    m.close()


def test_alpha_drawn_on_the_device_gives_the_same_proof(ctx, monkeypatch):
    """LURKHIP_DEV_ALPHA=1 (round 5, off by default: measured, no gain): the constraint-folding challenge is drawn by the device's
    copy of the transcript (k_fri_challenge on the permutation root), its powers are built from device memory and the quotient
    kernels read the cumulative sums where the permutation stage left them; the host catches its transcript up at the quotient
    root's read-back and compares the two alphas.  Same proof words as the host-drawn route, and the verifier accepts them."""
    from lurk_amd.programs import lurk_mix as lm

    mix = lm.fib_mix(300)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    m.setup()
    host = m.prove(q, num_queries=6, pow_bits=4, parse=False)
    monkeypatch.setenv("LURKHIP_DEV_ALPHA", "1")
    dev = m.prove(q, num_queries=6, pow_bits=4, parse=False)
    assert len(host) == len(dev) == 1 and np.array_equal(host[0], dev[0])
    assert m.verify([prover.parse_proof(dev[0])])
