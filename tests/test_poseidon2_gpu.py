"""GPU parity of the Poseidon2 kernels (through the C ABI) against the oracle and the KATs."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import field, synth
from lurk_amd.poseidon import Hasher, PoseidonChipset

from kat_helpers import compute_kats, load_kats

pytestmark = pytest.mark.gpu

WIDTHS = [4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48]


def test_known_answer_digests_on_gpu(ctx):
    got = compute_kats(Hasher(ctx))
    kats = load_kats()
    for name, hexval in got.items():
        assert hexval == kats[name]["digest_hex"], name


@pytest.mark.parametrize("width", WIDTHS)
def test_permute_matches_oracle(ctx, oracle, width):
    chip = PoseidonChipset(ctx, width)
    n = 1000  # ragged: not a multiple of the 256-thread block
    x = synth.field_elements((n, width), seed=synth.SEED + width)
    # edge values in the first rows
    x[0, :] = 0
    x[1, :] = field.P - 1
    x[2, :] = 1
    got = chip.permute_batch(x)
    want = oracle.p2_permute(width, x)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("width", [w for w in WIDTHS if w >= 8])
def test_hash8_and_witness_match_oracle(ctx, oracle, width):
    chip = PoseidonChipset(ctx, width)
    for n in (1, 63, 64, 65, 300):
        x = synth.field_elements((n, width), seed=synth.SEED + 7 * width + n)
        assert np.array_equal(chip.hash_batch(x), oracle.p2_hash8(width, x))
        got = chip.witness_batch(x)
        want = oracle.p2_wide_witness(width, x)
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"width {width} n {n}"


def test_montgomery_repr_is_equivalent(ctx, oracle):
    for width in (16, 24, 32, 40):
        chip = PoseidonChipset(ctx, width)
        x = synth.field_elements((257, width), seed=99 + width)
        want = oracle.p2_hash8(width, x)
        got_m = chip.hash_batch(field.to_monty(x), repr=lurk_amd.REPR_MONTY)
        assert np.array_equal(field.from_monty(got_m), want)
        wit_m = chip.witness_batch(field.to_monty(x), repr=lurk_amd.REPR_MONTY)
        assert np.array_equal(field.from_monty(wit_m), oracle.p2_wide_witness(width, x))


def test_empty_and_bad_arguments(ctx):
    chip = PoseidonChipset(ctx, 24)
    assert chip.hash_batch(np.zeros((0, 24), dtype=np.uint32)).shape == (0, 8)
    with pytest.raises(ValueError):
        PoseidonChipset(ctx, 17)
    with pytest.raises(ValueError):
        chip.hash_batch(np.zeros((3, 23), dtype=np.uint32))


def test_chipset_interface_matches_reference_shapes(ctx, oracle):
    # core/poseidon.rs:44-72: sizes, hash = first 8 lanes, populate_witness returns the full state
    chip = PoseidonChipset(ctx, 32)
    assert (chip.input_size(), chip.output_size(), chip.witness_size(), chip.require_size()) == (32, 8, 611, 0)
    x = synth.field_elements((32,), seed=5)
    wit = np.zeros(chip.witness_size(), dtype=np.uint32)
    out = chip.populate_witness(x, wit)
    assert len(out) == 32
    assert out[:8] == chip.execute_simple(x) == [int(v) for v in wit[:8]]
    assert np.array_equal(wit, oracle.p2_wide_witness(32, x)[0])


def test_device_pointer_path_large(ctx, oracle):
    """2^20 width-24 hashes through the _dev entry points on torch tensors; checks a checksum of all
    digests against the oracle on a sample and linear-time invariants on the rest."""
    import torch

    n = 1 << 20
    x = synth.field_elements((n, 24), seed=2024)
    xd = torch.from_numpy(x.view(np.int32)).cuda()
    od = torch.empty((n, 8), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    chip = PoseidonChipset(ctx, 24)
    chip.hash_dev(xd, od, n)
    ctx.sync()
    got = od.cpu().numpy().view(np.uint32)
    assert got.max() < field.P
    idx = np.concatenate([np.arange(0, 512), np.arange(n - 512, n), np.arange(0, n, 4099)])
    assert np.array_equal(got[idx], oracle.p2_hash8(24, x[idx]))
    # determinism / no cross-row interference: same rows hashed alone give the same digests
    assert np.array_equal(chip.hash_batch(x[idx]), got[idx])


def test_batched_zstore_hashing_on_gpu(ctx):
    """Level-order batched interning through the HIP hash kernels == one-at-a-time interning."""
    from lurk_amd.poseidon import Hasher
    from lurk_amd.zstore import ZStore

    words = ["lurk", "lurk-user", "builtin", "nil", "t", "cons", "lambda", "letrec", "fib", "quote", "a-rather-long-symbol-name"]
    a, b = ZStore(Hasher(ctx)), ZStore(Hasher(ctx))
    assert [a.intern_string(w) for w in words] == b.intern_strings(words)
    assert a.hashes == b.hashes
    mixed = [[1] * 24, list(range(32)), [7] * 40, [2] * 24]
    assert b.hash_many(mixed) == [tuple(a.hash(p)) for p in mixed]


def test_baseline_config2_two_to_the_24_permutations(ctx, oracle):
    """BASELINE config 2 at full size: 2^24 width-24 states (splitmix64 lanes, seed "LURK", the KAT preimage
    first).  The oracle checks a spread sample bit-exactly; the whole batch is covered by size-independent properties:
    digests are canonical, equal the first 8 lanes of the full permutation of the same state, and do not depend on the
    position of a state in the batch (the batch reversed gives the digests reversed)."""
    import torch

    n = 1 << 24
    x = synth.field_elements((n, 24))
    x[0] = 0
    x[0, 8], x[0, 16] = 1, 123  # src/core/zstore.rs:1008-1019
    xd = torch.from_numpy(x.view(np.int32)).cuda()
    hd = torch.empty((n, 8), dtype=torch.int32, device="cuda")
    pd = torch.empty((n, 24), dtype=torch.int32, device="cuda")
    chip = PoseidonChipset(ctx, 24)
    chip.hash_dev(xd, hd, n)
    chip.permute_dev(xd, pd, n)
    ctx.sync()
    assert torch.equal(hd, pd[:, :8])
    assert int(hd.min()) >= 0 and int(hd.max()) < field.P
    idx = np.concatenate([np.arange(0, 256), np.arange(n - 256, n), np.arange(0, n, 65521)])
    got = hd[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, oracle.p2_hash8(24, x[idx]))
    assert format(field.digest_to_int([int(v) for v in got[0]]), "x") == load_kats()["hash3_num123"]["digest_hex"]
    rd = torch.flip(xd, dims=[0]).contiguous()
    h2 = torch.empty_like(hd)
    torch.cuda.synchronize()  # the flip ran on torch's stream, the library has its own
    chip.hash_dev(rd, h2, n)
    ctx.sync()
    assert torch.equal(torch.flip(h2, dims=[0]), hd)
