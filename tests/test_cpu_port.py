"""The CPU port timed as bench.py's cpu_baseline (oracle/cpu_port.c: Montgomery row-major NTTs, vectorised, OpenMP) against the
checker it must agree with (oracle/commit.c: canonical per-column FFT, the definition-level LDE, the Merkle tree)."""
import numpy as np
import pytest

from lurk_amd import synth
from oracle import binding as ob


# (11, 700) and (13, 300): rows so wide that the cache-blocked NTT takes its stages in two and three groups (round 5)
@pytest.mark.parametrize("log_n,w", [(0, 3), (1, 1), (3, 5), (6, 13), (9, 78), (11, 4), (11, 700), (13, 300)])
def test_cpu_port_lde_equals_checker(log_n, w):
    x = synth.field_elements((1 << log_n, w), seed=100 * log_n + w)
    assert np.array_equal(ob.cpu_port_lde(x, 1), ob.lde(x, 1))
    if log_n <= 6:
        assert np.array_equal(ob.cpu_port_lde(x, 2), ob.lde(x, 2, naive=True))


def test_cpu_port_commit_round_equals_checker():
    """Mixed heights: matrices injected at inner levels, several of one height, a one-row matrix."""
    shapes = [(8, 78), (8, 9), (7, 148), (5, 4), (5, 4), (2, 12), (0, 44)]
    mats = [synth.field_elements((1 << lg, w), seed=7 * i + 1) for i, (lg, w) in enumerate(shapes)]
    want, _ = ob.merkle_commit([ob.lde(m, 1) for m in mats])
    assert np.array_equal(ob.cpu_port_commit_round(mats, 1), want)


@pytest.mark.parametrize("width", [16, 24, 32, 40])
def test_cpu_port_p2_hash8_equals_checker(width):
    """The packed (sixteen rows per AVX-512 register set, all cores) hash of BASELINE config 2's CPU leg against the scalar oracle,
    which the reference's known answers pin (tests/test_oracle_kat.py): ragged row counts, the extreme words 0 and p - 1."""
    from lurk_amd import synth

    for n in (1, 15, 16, 17, 1000 + width):
        x = synth.field_elements((n, width), seed=width + n)
        x[0, :] = 2013265920
        if n > 1:
            x[1, :] = 0
        try:
            got = ob.cpu_port_p2_hash8(width, x)
        except RuntimeError:
            pytest.skip("no AVX-512 on this CPU: the bench falls back to the scalar oracle there")
        assert np.array_equal(got, ob.p2_hash8(width, x)), (width, n)
