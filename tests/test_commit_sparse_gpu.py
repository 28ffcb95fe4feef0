"""Commitments that leave identically-zero columns out of the LDE (round 5: `commit_impl(live_runs)`, through
`lurkhip_commit_dev_sparse`, which finds the columns itself).  The extension of the zero polynomial is zero, so root, LDE words
and openings must equal `lurkhip_commit_dev`'s -- and the oracle's -- for every pattern of zero columns: none, all, the first
columns of a matrix, more runs than a launch group holds (LDE_MAX_MATS = 32: the narrowest gaps are bridged), several matrices of
one height, matrices below 2^11 rows (one LDE kernel: nothing is left out), both buffer layouts, both representations."""
import numpy as np
import pytest
import torch

import lurk_amd
from lurk_amd import commit as cm
from lurk_amd import field, synth

pytestmark = pytest.mark.gpu


def _with_zero_columns(shape, seed, zero_cols):
    x = synth.field_elements(shape, seed=seed)
    x[:, list(zero_cols)] = 0
    return x


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).cuda()


def _check(ctx, oracle, mats, aligned, expect_zero=None, repr=lurk_amd.REPR_CANONICAL):
    lh = [m.shape[0].bit_length() - 1 for m in mats]
    ws = [m.shape[1] for m in mats]
    given = [field.to_monty(m) if repr == lurk_amd.REPR_MONTY else m for m in mats]
    devs = [_dev(m) for m in given]
    torch.cuda.synchronize()
    dense = cm.commit_dev(ctx, devs, lh, ws, 1, repr=repr)
    sparse, n_zero = cm.commit_dev_sparse(ctx, devs, lh, ws, 1, repr=repr, aligned_groups=aligned)
    if expect_zero is not None:
        assert n_zero == expect_zero
    assert np.array_equal(sparse.root, dense.root)
    ldes = [oracle.lde(m, 1) for m in mats]
    want_root, _ = oracle.merkle_commit(ldes)
    root = sparse.root if repr == lurk_amd.REPR_CANONICAL else field.from_monty(np.asarray(sparse.root, dtype=np.uint32))
    assert np.array_equal(root, want_root)
    for i in range(len(mats)):
        got = sparse.lde_host(i)
        assert np.array_equal(got, dense.lde_host(i)), i
    h_max = 2 << max(lh)
    for index in (0, 1, h_max // 2 + 3, h_max - 1):
        ra, pa = sparse.open(index)
        rb, pb = dense.open(index)
        assert np.array_equal(ra, rb) and np.array_equal(pa, pb)
    sparse.close()
    dense.close()


@pytest.mark.parametrize("aligned", [False, True], ids=["dense-buffers", "aligned-groups"])
def test_zero_column_patterns(ctx, oracle, aligned):
    n = 1 << 12
    # no zero column at all; every column zero; the first columns; the last; scattered single columns
    _check(ctx, oracle, [_with_zero_columns((n, 40), 1, [])], aligned, 0)
    _check(ctx, oracle, [_with_zero_columns((n, 40), 2, range(40))], aligned, 40)
    _check(ctx, oracle, [_with_zero_columns((n, 40), 3, range(0, 9))], aligned, 9)
    _check(ctx, oracle, [_with_zero_columns((n, 77), 4, range(60, 77))], aligned, 17)
    _check(ctx, oracle, [_with_zero_columns((n, 77), 5, [0, 5, 6, 31, 32, 33, 64, 76])], aligned, 8)


@pytest.mark.parametrize("aligned", [False, True], ids=["dense-buffers", "aligned-groups"])
def test_more_runs_than_a_launch_holds(ctx, oracle, aligned):
    # every other column zero: 80 live runs in one matrix (LDE_MAX_MATS = 32): the narrowest gaps are bridged, the words do not change
    n = 1 << 11
    _check(ctx, oracle, [_with_zero_columns((n, 160), 6, range(0, 160, 2))], aligned, 80)
    # ... beside two more matrices of the same height, one of them all zero, and a shorter and a taller one
    mats = [_with_zero_columns((n, 160), 7, range(1, 160, 2)), _with_zero_columns((n, 12), 8, range(12)), _with_zero_columns((n, 33), 9, [4, 5, 6, 7]),
            _with_zero_columns((1 << 13, 20), 10, range(8, 16)), _with_zero_columns((1 << 8, 50), 11, range(10, 30))]
    _check(ctx, oracle, mats, aligned)


def test_montgomery_words_and_a_permutation_shaped_matrix(ctx, oracle):
    # extension-field columns (four words each), most of them dead: the shape of a permutation trace
    n = 1 << 13
    dead = [c for k in range(0, 79) if k % 3 for c in range(4 * k, 4 * k + 4)]
    _check(ctx, oracle, [_with_zero_columns((n, 316), 12, dead), _with_zero_columns((n, 92), 13, range(8, 88))], True, repr=lurk_amd.REPR_MONTY)
