"""CPU checks of the commit-stage oracle against its own definitions (parity vs the reference is
unpinned for this stage: no golden commitments exist in the reference)."""
import numpy as np

from lurk_amd import synth
from lurk_amd.field import P


def test_fft_lde_matches_quadratic_definition(oracle):
    for log_n, w, b in [(0, 3, 1), (1, 2, 1), (3, 5, 1), (5, 3, 2), (6, 2, 0)]:
        x = synth.field_elements((1 << log_n, w), seed=100 + log_n)
        assert np.array_equal(oracle.lde(x, b), oracle.lde(x, b, naive=True)), (log_n, w, b)


def test_lde_of_low_degree_polynomial(oracle):
    # evaluations of f(x) = 3 + 5x over H extend to f on the coset: row bitrev(j) = 3 + 5 * 31 * w^j
    log_n, b = 4, 1
    n = 1 << log_n
    w_n = pow(0x1A427A41, 1 << (27 - log_n), P)
    w_m = pow(0x1A427A41, 1 << (27 - log_n - b), P)
    evals = np.array([[(3 + 5 * pow(w_n, i, P)) % P] for i in range(n)], dtype=np.uint32)
    got = oracle.lde(evals, b)
    bits = log_n + b
    for j in range(n << b):
        r = int(format(j, f"0{bits}b")[::-1], 2)
        assert int(got[r, 0]) == (3 + 5 * 31 * pow(w_m, j, P)) % P


def test_merkle_openings_verify(oracle):
    mats = [synth.field_elements((16, 5), seed=1), synth.field_elements((16, 9), seed=2), synth.field_elements((4, 3), seed=3)]
    root, digests = oracle.merkle_commit(mats)
    lh, ws = [4, 4, 2], [5, 9, 3]
    for index in (0, 7, 15):
        rows = np.concatenate([mats[0][index], mats[1][index], mats[2][index >> 2]])
        path = []
        off = 0
        for l in range(4):
            sib = (index >> l) ^ 1
            path.append(digests[off + sib])
            off += 16 >> l
        assert oracle.merkle_verify(lh, ws, index, rows, np.array(path), root)
        bad = rows.copy()
        bad[-1] ^= 1  # corrupt the injected short matrix's row
        assert not oracle.merkle_verify(lh, ws, index, bad, np.array(path), root)
