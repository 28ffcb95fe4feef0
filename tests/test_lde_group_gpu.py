"""The grouped coset LDE (csrc/lde.hip) on the shapes that are specific to it: a height group larger than one launch group (more
than 16 matrices), every tile height (2^5 .. 2^12 rows: the one-kernel route up to 2^10, three kernels above), groups with
several coset shifts (a third shift closes a launch group), one-column and very wide rows, canonical against Montgomery input.
Bit-exact against the oracle; tests/test_commit_gpu.py covers the general shapes through the same route."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import commit as cm
from lurk_amd import field, synth
from oracle import stark as os_

pytestmark = pytest.mark.gpu
P = field.P


def _dev(mats):
    import torch

    return [torch.from_numpy(np.ascontiguousarray(m).view(np.int32)).cuda() for m in mats]


def test_more_matrices_of_one_height_than_a_launch_group_holds(ctx, oracle):
    widths = [1, 2, 3, 5, 8, 13, 31, 32, 33, 64, 7, 7, 9, 100, 4, 4, 4, 17, 21, 2, 1]  # 21 matrices at 2^9: two launch groups
    mats = [synth.field_elements((1 << 9, w), seed=8100 + i) for i, w in enumerate(widths)]
    c = cm.commit_dev(ctx, _dev(mats), [9] * len(mats), widths, log_blowup=1)
    ldes = [oracle.lde(m, 1) for m in mats]
    root, _ = oracle.merkle_commit(ldes)
    assert np.array_equal(c.root, root)
    for i in (0, 8, 15, 16, 20):
        assert np.array_equal(c.lde_host(i), ldes[i]), i
    c.close()


@pytest.mark.parametrize("log_n", list(range(5, 13)))
def test_every_tile_height(ctx, oracle, log_n):
    widths = [78, 1, 37]
    mats = [synth.field_elements((1 << log_n, w), seed=8200 + 10 * log_n + i) for i, w in enumerate(widths)]
    c = cm.commit_dev(ctx, _dev(mats), [log_n] * 3, widths, log_blowup=1)
    for i, m in enumerate(mats):
        assert np.array_equal(c.lde_host(i), oracle.lde(m, 1)), (log_n, i)
    c.close()


def test_three_coset_shifts_at_one_height(ctx):
    """Quotient chunks of a chip with four chunks: shifts w_Q^-c.  Two shifts share a launch group, the third opens the next."""
    log_n, lqd = 6, 2
    wq = os_.two_adic_generator(log_n + lqd)
    shifts = [pow(wq, (-c) % (P - 1), P) for c in range(4)]
    mats = [synth.field_elements((1 << log_n, 4), seed=8300 + c) for c in range(4)] + [synth.field_elements((1 << log_n, 8), seed=8310)]
    all_shifts = shifts + [shifts[1]]
    c = cm.commit_cosets_dev(ctx, _dev([field.to_monty(m) for m in mats]), [log_n] * 5, [4, 4, 4, 4, 8], all_shifts, log_blowup=1)
    for i, (m, sh) in enumerate(zip(mats, all_shifts)):
        want = os_.bit_reverse_rows(os_.coset_lde([[int(v) for v in r] for r in m], 1, shift=sh))
        assert c.lde_host(i).tolist() == want, i  # (lde_host returns canonical words)
    c.close()


def test_very_wide_rows_and_montgomery_input(ctx, oracle):
    m = synth.field_elements((1 << 7, 2067), seed=8400)  # the hash chips' height group of a fib machine: 65 column chunks
    c = cm.commit_dev(ctx, _dev([m]), [7], [2067], log_blowup=1)
    want = oracle.lde(m, 1)
    assert np.array_equal(c.lde_host(0), want)
    c.close()
    cmont = cm.commit_dev(ctx, _dev([field.to_monty(m)]), [7], [2067], log_blowup=1, repr=lurk_amd.REPR_MONTY)
    assert np.array_equal(cmont.lde_host(0), want)
    cmont.close()
