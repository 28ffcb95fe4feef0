"""GPU parity of the commit stage (coset LDE + Merkle) against the oracle, plus size-independent
properties at the benchmark shape."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import commit as cm
from lurk_amd import field, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,w,b", [(0, 3, 1), (1, 5, 1), (3, 78, 1), (6, 1, 1), (7, 13, 1), (8, 78, 1), (10, 9, 2), (12, 78, 1), (14, 130, 1), (15, 7, 0)])
def test_lde_matches_oracle(ctx, oracle, log_n, w, b):
    x = synth.field_elements((1 << log_n, w), seed=300 + log_n + w)
    got = cm.coset_lde(ctx, x, b)
    want = oracle.lde(x, b)
    assert np.array_equal(got, want)


# widths and heights that exercise every tiling of the NTT passes: adjacent-row grouping for narrow matrices (w = 4 ... 13 at
# heights with strided passes), single-chunk tiles up to 112 columns, evenly divided chunks (w = 224), ragged last chunks
# (113, 130, 493, 655, 815 = the hash chips), odd widths (one-column butterflies), 256 / 512 / 1024-thread tiles
@pytest.mark.parametrize("log_n,w", [(14, 4), (15, 8), (16, 9), (17, 6), (14, 13), (14, 96), (13, 112), (13, 113), (12, 224), (12, 493),
                                     (11, 655), (11, 815), (9, 815), (16, 78)])
def test_lde_tilings_match_oracle(ctx, oracle, log_n, w):
    x = synth.field_elements((1 << log_n, w), seed=900 + log_n + w)
    assert np.array_equal(cm.coset_lde(ctx, x, 1), oracle.lde(x, 1))


def test_lde_64bit_offset_kernels(ctx, oracle, monkeypatch):
    """Matrices of 4 GiB and more run kernel variants with 64-bit row offsets; the test hook forces them at small sizes."""
    monkeypatch.setenv("LURKHIP_NTT_FORCE_64BIT", "1")
    for log_n, w in [(12, 78), (14, 9), (10, 113), (15, 96)]:
        x = synth.field_elements((1 << log_n, w), seed=1200 + log_n + w)
        assert np.array_equal(cm.coset_lde(ctx, x, 1), oracle.lde(x, 1)), (log_n, w)


def test_lde_montgomery_repr(ctx, oracle):
    x = synth.field_elements((256, 10), seed=8)
    got = cm.coset_lde(ctx, field.to_monty(x), 1, repr=lurk_amd.REPR_MONTY)
    assert np.array_equal(field.from_monty(got), oracle.lde(x, 1))


def test_commit_single_matrix_root_and_openings(ctx, oracle):
    x = synth.field_elements((1 << 10, 78), seed=42)
    c = cm.commit(ctx, [x], log_blowup=1)
    lde = oracle.lde(x, 1)
    root, _ = oracle.merkle_commit([lde])
    assert np.array_equal(c.root, root)
    assert np.array_equal(c.lde_host(0), lde)
    for index in (0, 1, 1023, 2047):
        rows, path = c.open(index)
        assert np.array_equal(rows, lde[index])
        assert oracle.merkle_verify([11], [78], index, rows, path, c.root)
    c.close()


def test_commit_mixed_heights(ctx, oracle):
    # irregular width mix in the spirit of BASELINE config 5: func chips, mem chips, bytes, entrypoint
    shapes = [(9, 78), (9, 114), (6, 52), (4, 493), (9, 8), (2, 6), (0, 44), (6, 13)]
    mats = [synth.field_elements((1 << k, w), seed=500 + i) for i, (k, w) in enumerate(shapes)]
    c = cm.commit(ctx, mats, log_blowup=1)
    ldes = [oracle.lde(m, 1) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root)
    lh = [k + 1 for k, _ in shapes]
    ws = [w for _, w in shapes]
    for index in (0, 5, 777, 1023):
        rows, path = c.open(index)
        want = np.concatenate([ldes[i][index >> (10 - lh[i])] for i in range(len(mats))])
        assert np.array_equal(rows, want)
        assert oracle.merkle_verify(lh, ws, index, rows, path, c.root)
    c.close()


def test_commit_injection_at_cooperative_levels(ctx, oracle):
    """Levels of at most 16384 parents hash lane-cooperatively (16 lanes per node), the last 64 nodes inside one workgroup:
    matrices injected at such levels -- narrow, wider than one sponge block, and as wide as the hash chips -- and at the
    one-lane-per-node levels above them must give the oracle's tree."""
    shapes = [(15, 8), (14, 5), (13, 20), (12, 9), (10, 493), (9, 5), (8, 17), (4, 11), (3, 7), (0, 3)]
    mats = [synth.field_elements((1 << k, w), seed=1500 + i) for i, (k, w) in enumerate(shapes)]
    c = cm.commit(ctx, mats, log_blowup=1)
    ldes = [oracle.lde(m, 1) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root)
    lh = [k + 1 for k, _ in shapes]
    ws = [w for _, w in shapes]
    for index in (0, 4097, 65535):
        rows, path = c.open(index)
        assert oracle.merkle_verify(lh, ws, index, rows, path, c.root)
    c.close()


def test_commit_many_matrices_per_height_group(ctx, oracle):
    """The column table of a height group is written on the device from launch arguments, 24 matrices per launch: groups of
    more than 24 (and of exactly 24, 25 and 49) matrices -- as leaves, injected at a one-lane-per-node level, at a cooperative
    level and inside the one-workgroup top -- must give the oracle's tree, and the openings their rows."""
    shapes = [(15, 3)] * 25 + [(14, 2)] * 49 + [(9, 5)] * 24 + [(4, 7)] * 26 + [(2, 1)] * 30
    mats = [synth.field_elements((1 << k, w), seed=2600 + i) for i, (k, w) in enumerate(shapes)]
    c = cm.commit(ctx, mats, log_blowup=1)
    ldes = [oracle.lde(m, 1) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root)
    lh = [k + 1 for k, _ in shapes]
    ws = [w for _, w in shapes]
    for index in (0, 12345, 65535):
        rows, path = c.open(index)
        want = np.concatenate([ldes[i][index >> (16 - lh[i])] for i in range(len(mats))])
        assert np.array_equal(rows, want)
        assert oracle.merkle_verify(lh, ws, index, rows, path, c.root)
    c.close()


def test_commit_long_rows_hashed_by_sixteen_lanes(ctx, oracle):
    """Height groups of at most 4096 LDE rows with at least 16 permutations per row are hashed sixteen lanes to the row
    (merkle.hip: coop_sponge_row): widths around the threshold and around multiples of eight, one to 4096 rows, next to a group of
    the same width just above the row limit (one row per lane) -- the oracle's tree, and the openings' rows."""
    # (a 2^16-leaf tree: the row groups of trees above 16384 leaves share the sponge launch that has the sixteen-lane mode)
    shapes = [(15, 3), (12, 121), (11, 121), (10, 127), (9, 128), (8, 129), (7, 655), (3, 120), (1, 250), (0, 135)]
    mats = [synth.field_elements((1 << k, w), seed=3100 + i) for i, (k, w) in enumerate(shapes)]
    c = cm.commit(ctx, mats, log_blowup=1)
    ldes = [oracle.lde(m, 1) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root)
    lh = [k + 1 for k, _ in shapes]
    ws = [w for _, w in shapes]
    for index in (0, 4097, 65535):
        rows, path = c.open(index)
        want = np.concatenate([ldes[i][index >> (16 - lh[i])] for i in range(len(mats))])
        assert np.array_equal(rows, want)
        assert oracle.merkle_verify(lh, ws, index, rows, path, c.root)
    c.close()


def test_commit_same_shape_matrices_share_their_passes(ctx, oracle):
    """Device-resident matrices of one (height, width) go through the NTT passes in one launch per pass (up to 8 per launch):
    eleven 2^10 x 4 matrices (a batch of 8 and one of 3), three 2^9 x 12, pairs with an odd width and a single one."""
    import torch

    shapes = [(10, 4)] * 11 + [(9, 12)] * 3 + [(8, 7)] * 2 + [(11, 5)]
    mats = [synth.field_elements((1 << k, w), seed=1700 + i) for i, (k, w) in enumerate(shapes)]
    dev = [torch.from_numpy(m.view(np.int32)).cuda() for m in mats]
    c = cm.commit_dev(ctx, dev, [k for k, _ in shapes], [w for _, w in shapes], log_blowup=1)
    ldes = [oracle.lde(m, 1) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root)
    for i in (0, 7, 8, 10, 11, 13, 14, 15, 16):
        assert np.array_equal(c.lde_host(i), ldes[i]), i
    c.close()


def test_commit_buffer_reuse_across_shapes(ctx, oracle):
    """The pooled allocator hands a released LDE buffer to the next commit of the same byte size, and the context caches the
    Merkle leaf-column tables and the coset-shift power tables by (pointer, width) / (height, shift): commits of different
    shapes that land in the same buffers, and repeats of one shape, must each give the oracle's root."""
    shapes = [(10, 8), (9, 16), (8, 32), (10, 8), (9, 16), (11, 4), (10, 8)]  # all 2^13 words: one pool bucket
    for i, (k, w) in enumerate(shapes):
        x = synth.field_elements((1 << k, w), seed=900 + i)
        c = cm.commit(ctx, [x], log_blowup=1)
        lde = oracle.lde(x, 1)
        root, _ = oracle.merkle_commit([lde])
        assert np.array_equal(c.root, root), (i, k, w)
        assert np.array_equal(c.lde_host(0), lde)
        c.close()
    # two live commitments of the same shape at once: distinct buffers, distinct tables
    xs = [synth.field_elements((1 << 9, 16), seed=950 + i) for i in range(2)]
    cs = [cm.commit(ctx, [x], log_blowup=1) for x in xs]
    for x, c in zip(xs, cs):
        root, _ = oracle.merkle_commit([oracle.lde(x, 1)])
        assert np.array_equal(c.root, root)
        c.close()


def test_commit_height_one_and_two(ctx, oracle):
    for k in (0, 1):
        x = synth.field_elements((1 << k, 44), seed=70 + k)
        c = cm.commit(ctx, [x], log_blowup=1)
        root, _ = oracle.merkle_commit([oracle.lde(x, 1)])
        assert np.array_equal(c.root, root)
        c.close()


def test_custom_merkle_constants_change_the_root(ctx):
    import ctypes as C

    from lurk_amd import _native as N

    x = synth.field_elements((64, 8), seed=9)
    r0 = cm.commit(ctx, [x]).root.copy()
    ext = synth.field_elements((128,), seed=1)
    inn = synth.field_elements((13,), seed=2)
    diag = synth.field_elements((16,), seed=3)
    ctx.check(N.lib.lurkhip_set_merkle_poseidon2(ctx.handle, 13, ext.ctypes.data, inn.ctypes.data, diag.ctypes.data))
    r1 = cm.commit(ctx, [x]).root.copy()
    assert not np.array_equal(r0, r1)
    # restore the default table (in-tree BabyBearConfig16) for the other tests
    from oracle import binding as ob  # test-only: read the same numbers the oracle uses

    import re, os
    hdr = open(os.path.join(os.path.dirname(N.LIB_PATH), "csrc", "p2_params.h")).read()

    def table(name):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", hdr, re.S).group(1)
        return np.array([int(v) for v in re.findall(r"(\d+)u", body)], dtype=np.uint32)

    ext, inn, diag = table("LURK_P2_EXT_RC_16"), table("LURK_P2_INT_RC_16"), table("LURK_P2_DIAG_16")
    ctx.check(N.lib.lurkhip_set_merkle_poseidon2(ctx.handle, 13, ext.ctypes.data, inn.ctypes.data, diag.ctypes.data))
    assert np.array_equal(cm.commit(ctx, [x]).root, r0)


def test_lde_properties_at_bench_shape(ctx, oracle):
    """2^20 x 78 (BASELINE config 3) is too big for the O(N log N) CPU oracle on every column in test
    time, so: (a) 3 sampled columns are checked bit-exactly against the oracle FFT, (b) linearity
    LDE(a + 2b) = LDE(a) + 2 LDE(b) is checked on every entry, (c) Merkle openings verify."""
    import torch

    log_n, w = 20, 78
    n = 1 << log_n
    a = synth.field_elements((n, w), seed=11)
    a[:, 0] = np.arange(n, dtype=np.uint32)  # col 0 = row index as in the synthetic bench trace
    b = synth.field_elements((n, w), seed=12)
    P = field.P
    comb = ((a.astype(np.uint64) + 2 * b.astype(np.uint64)) % P).astype(np.uint32)

    def lde_dev(x):
        xd = torch.from_numpy(x.view(np.int32)).cuda()
        od = torch.empty((2 * n, w), dtype=torch.int32, device="cuda")
        cm.coset_lde_dev(ctx, log_n, w, 1, xd, od)
        ctx.sync()
        return od.cpu().numpy().view(np.uint32)

    la, lb, lc = lde_dev(a), lde_dev(b), lde_dev(comb)
    assert np.array_equal(lc, ((la.astype(np.uint64) + 2 * lb.astype(np.uint64)) % P).astype(np.uint32))
    for col in (0, 37, 77):
        assert np.array_equal(la[:, col], oracle.lde(a[:, col : col + 1], 1)[:, 0])
    del lb, lc, b, comb
    ad = torch.from_numpy(a.view(np.int32)).cuda()
    c = cm.commit_dev(ctx, [ad], [log_n], [w], log_blowup=1)
    for index in (0, 123456, 2 * n - 1):
        rows, path = c.open(index)
        assert np.array_equal(rows, la[index])
        assert oracle.merkle_verify([log_n + 1], [w], index, rows, path, c.root)
    c.close()


# BASELINE config 5: the width mix of the full Lurk machine - 39 funcs in the order of `test_widths`
# (src/core/tests/eval_direct.rs:2025-2063), six memory tables, the byte table, the entrypoint row.
LURK_FUNC_WIDTHS = [97, 188, 10, 78, 148, 110, 81, 79, 97, 115, 78, 107, 70, 68, 72, 94, 66, 54, 66, 9, 50, 86, 58, 61, 114, 52, 104, 81,
                    493, 655, 815, 53, 53, 85, 166, 44, 26, 38, 78]


def lurk_machine_shapes(log_max):
    """(log_height, width) per chip: eval / apply / env_lookup (widths 78, 114, 52) at the largest height, hash chips at most
    2^(log_max - 4), the rest drawn from splitmix64 between 2^2 and 2^(log_max - 1)."""
    draws = synth.splitmix64(len(LURK_FUNC_WIDTHS), synth.SEED + 5)
    shapes = []
    for i, w in enumerate(LURK_FUNC_WIDTHS):
        if i in (3, 24, 25):
            k = log_max
        elif w in (493, 655, 815):
            k = max(log_max - 4 - i % 3, 1)
        else:
            k = 2 + int(draws[i] % np.uint64(log_max - 2))
        shapes.append((k, w))
    shapes += [(log_max - 1 - i, w) for i, w in enumerate((6, 7, 8, 9, 10, 12))]
    shapes += [(16 if log_max >= 16 else log_max, 13), (0, 44)]
    return shapes


def test_baseline_config5_lurk_machine_width_mix(ctx, oracle):
    """One mixed-height commitment over all 47 matrices: root == oracle, openings at every height verify."""
    log_max = 12
    shapes = lurk_machine_shapes(log_max)
    mats = [synth.field_elements((1 << k, w), seed=5000 + i) for i, (k, w) in enumerate(shapes)]
    c = cm.commit(ctx, mats, log_blowup=1)
    ldes = [oracle.lde(m, 1) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root)
    lh = [k + 1 for k, _ in shapes]
    ws = [w for _, w in shapes]
    top = max(lh)
    for index in (0, 1, 4097, (1 << top) - 1):
        rows, path = c.open(index)
        want = np.concatenate([ldes[i][index >> (top - lh[i])] for i in range(len(mats))])
        assert np.array_equal(rows, want)
        assert oracle.merkle_verify(lh, ws, index, rows, path, c.root)
    c.close()


def test_baseline_config5_at_full_height(ctx, oracle):
    """The same mix with the big chips at 2^18 rows (≈ 1.6 GB of traces and LDEs): too slow for the oracle's tree on one core, so the root
    is tied to the oracle through openings - every opened row must equal the oracle's LDE of that row's column (sampled
    columns) and every path must verify against the root with the oracle's Merkle verifier."""
    import torch

    log_max = 18
    shapes = lurk_machine_shapes(log_max)
    mats = [synth.field_elements((1 << k, w), seed=6000 + i) for i, (k, w) in enumerate(shapes)]
    dev = [torch.from_numpy(m.view(np.int32)).cuda() for m in mats]
    c = cm.commit_dev(ctx, dev, [k for k, _ in shapes], [w for _, w in shapes], log_blowup=1)
    lh = [k + 1 for k, _ in shapes]
    ws = [w for _, w in shapes]
    top = max(lh)
    col_lde = {i: oracle.lde(np.ascontiguousarray(mats[i][:, 7:8]), 1)[:, 0] for i in (3, 24, 29, 30, 41)}
    offs = np.concatenate([[0], np.cumsum(ws)])
    for index in (0, 3, 99991, (1 << top) - 1):
        rows, path = c.open(index)
        assert oracle.merkle_verify(lh, ws, index, rows, path, c.root)
        for i, col in col_lde.items():
            assert rows[offs[i] + 7] == col[index >> (top - lh[i])]
    c.close()


@pytest.mark.parametrize("seed", range(40))
def test_commit_random_shape_sets(ctx, oracle, seed):
    """Random sets of 1 .. 14 matrices -- heights 2^0 .. 2^13 (repeated heights, gaps, one tall matrix or none), widths 1 .. 140
    with a few hash-chip-wide ones, blow-up 1 or 2 -- against the oracle: root, opened rows at random indices, Merkle paths."""
    import random

    rng = random.Random(7000 + seed)
    n = rng.randint(1, 14)
    top = rng.randint(0, 13)
    shapes = []
    for i in range(n):
        k = top if i == 0 else rng.choice([top, rng.randint(0, top), rng.randint(0, top), max(0, top - 1)])
        w = rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 17, 31, 32, 33, 52, 64, 78, 107, 140, rng.randint(1, 140)])
        if rng.random() < 0.08 and k <= 9:
            w = rng.choice([493, 655, 815])
        shapes.append((k, w))
    b = rng.choice([1, 1, 2])
    mats = [synth.field_elements((1 << k, w), seed=9000 + 40 * seed + i) for i, (k, w) in enumerate(shapes)]
    c = cm.commit(ctx, mats, log_blowup=b)
    ldes = [oracle.lde(m, b) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root), shapes
    lh = [k + b for k, _ in shapes]
    ws = [w for _, w in shapes]
    hmax = max(lh)
    for index in [0, (1 << hmax) - 1] + [rng.randrange(1 << hmax) for _ in range(3)]:
        rows, path = c.open(index)
        want = np.concatenate([ldes[i][index >> (hmax - lh[i])] for i in range(len(mats))])
        assert np.array_equal(rows, want), (shapes, index)
        assert oracle.merkle_verify(lh, ws, index, rows, path, c.root), (shapes, index)
    c.close()
