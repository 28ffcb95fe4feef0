"""The narrow Poseidon2 chip: one trace row per round.

CPU part: the oracle's restatement satisfies the two properties the reference tests
(/root/reference/src/poseidon/mod.rs:45-88): the tail of row R equals hasher.permute(input), and every constraint of
the chip's AIR vanishes on every row of a generated trace (next row taken cyclically, as p3's debug checker does).
GPU part: trace == oracle bit for bit, the AIR register program == the oracle's AIR on real and random rows, the
device-side trace check, and a whole STARK proof of a batch of permutations accepted by the oracle's verifier."""
import numpy as np
import pytest

from lurk_amd import field, synth
from oracle import air as oa

REFERENCE_TEST_WIDTHS = [4, 8, 12, 16, 24, 40]  # poseidon/mod.rs:59-66,80-87
ALL_WIDTHS = [4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48]


def oracle_constraints_vanish(width, trace):
    a = oa.Poseidon2NarrowAir(width)
    n = trace.shape[0]
    rows = [[int(v) for v in r] for r in trace]
    for i in range(n):
        b = oa.Builder(rows[i], rows[(i + 1) % n])
        a.eval(b)
        bad = [k for k, c in enumerate(b.constraints) if c]
        if bad:
            return i, bad[0]
    return None


@pytest.mark.parametrize("width", REFERENCE_TEST_WIDTHS)
def test_oracle_trace_eq_hash_and_air_constraints(oracle, width):
    rp = oracle.p2_params(width)[0]
    x = np.arange(width, dtype=np.uint32)[None, :]  # array::from_fn(from_canonical_usize), mod.rs:48,74
    t = oracle.p2_narrow_trace(width, x)
    assert t.shape == (32 if 8 + rp + 1 <= 32 else 64, 5 * width + 1 + 8 + rp)
    assert np.array_equal(t[8 + rp, -width:], oracle.p2_permute(width, x)[0])
    assert oracle_constraints_vanish(width, t) is None


def test_oracle_air_rejects_a_broken_trace(oracle):
    x = synth.field_elements((3, 16), seed=77)
    t = oracle.p2_narrow_trace(16, x)
    assert oracle_constraints_vanish(16, t) is None
    rp = oracle.p2_params(16)[0]
    bad = t.copy()
    bad[5, 3] = (bad[5, 3] + 1) % field.P  # an input lane of round row 5: breaks the link to row 4's output
    row, _ = oracle_constraints_vanish(16, bad)
    assert row == 4
    bad = t.copy()
    bad[8 + rp + 1 + 2, 16] = 1  # is_init set on a round row
    assert oracle_constraints_vanish(16, bad) is not None


# ---------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("width", ALL_WIDTHS)
def test_trace_matches_oracle(ctx, oracle, width):
    from lurk_amd.poseidon import Poseidon2Chip, PoseidonChipset

    chip = Poseidon2Chip(ctx, width)
    for n in (1, 5, 64, 300):
        x = synth.field_elements((n, width), seed=40 + n + width)
        got = chip.generate_trace(x)
        want = oracle.p2_narrow_trace(width, x)
        assert got.shape == want.shape == chip.shape(n)
        assert np.array_equal(got, want)
    rp = oracle.p2_params(width)[0]
    x = np.arange(width, dtype=np.uint32)[None, :]
    t = chip.generate_trace(x)
    assert np.array_equal(t[8 + rp, -width:], PoseidonChipset(ctx, width).permute_batch(x)[0])
    # Montgomery in / out is the same trace
    tm = chip.generate_trace(field.to_monty(x), repr=1)
    assert np.array_equal(field.from_monty(tm), t)


@pytest.mark.gpu
def test_empty_batch_is_one_zero_row(ctx):
    from lurk_amd.poseidon import Poseidon2Chip

    t = Poseidon2Chip(ctx, 24).generate_trace(np.zeros((0, 24), dtype=np.uint32))
    assert t.shape == (1, 150) and not t.any()


@pytest.mark.gpu
@pytest.mark.parametrize("width", [4, 16, 24, 40])
def test_air_program_matches_oracle(ctx, oracle, width):
    from lurk_amd.air import ChipAir
    from test_air_gpu import compare, rows_for

    a = ChipAir.for_poseidon2(width)
    o = oa.Poseidon2NarrowAir(width)
    assert (a.width, a.num_sends, a.num_receives, a.max_constraint_degree) == (o.width, 0, 0, 3)
    t = oracle.p2_narrow_trace(width, synth.field_elements((2, width), seed=9))
    local, nxt, sels = rows_for(o.width, [[int(v) for v in r] for r in t[:40]], seed=500 + width)
    compare(ctx, a, o, local, nxt, sels)


@pytest.mark.gpu
def test_check_trace_on_device(ctx):
    import torch

    from lurk_amd.poseidon import Poseidon2Chip

    chip = Poseidon2Chip(ctx, 24)
    n = 1000
    h, w = chip.shape(n)
    x = torch.from_numpy(field.to_monty(synth.field_elements((n, 24), seed=3)).view(np.int32)).cuda()
    t = torch.empty((h, w), dtype=torch.int32, device="cuda")
    chip.generate_trace_dev(x, t, n, repr=1)
    ctx.sync()
    a = chip.air()
    assert a.check_trace(ctx, h, t) == (-1, -1)
    t[12345, 7] += 1
    torch.cuda.synchronize()
    row, k = a.check_trace(ctx, h, t)
    assert row == 12344 and k >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("width,n", [(16, 20), (24, 1100)])
def test_batch_of_permutations_proves_and_verifies(ctx, oracle, width, n):
    import copy

    import torch

    from lurk_amd import prover
    from lurk_amd.poseidon import Poseidon2Chip
    from oracle import stark as os_

    chip = Poseidon2Chip(ctx, width)
    h, w = chip.shape(n)
    x = torch.from_numpy(field.to_monty(synth.field_elements((n, width), seed=21)).view(np.int32)).cuda()
    t = torch.empty((h, w), dtype=torch.int32, device="cuda")
    chip.generate_trace_dev(x, t, n, repr=1)
    ctx.sync()
    m = prover.StarkMachine(ctx, [chip.air()])
    root = m.setup()
    proof = m.prove([t], num_queries=8, pow_bits=4)
    airs = [oa.Poseidon2NarrowAir(width)]
    assert os_.verify_machine(airs, root, [], [], [proof], oracle.merkle_verify)
    assert m.verify(proof)  # the product's host verifier: a machine without preprocessed traces or interactions
    bad = copy.deepcopy(proof)
    loc, nxt = bad.chips[0].opened["main"]
    loc[1] = ((loc[1][0] + 1) % field.P,) + tuple(loc[1][1:])
    with pytest.raises(os_.VerifyError):
        os_.verify_machine(airs, root, [], [], [bad], oracle.merkle_verify)
    from proof_words import encode_words

    with pytest.raises(prover.VerificationError):
        m.verify(encode_words(bad))
