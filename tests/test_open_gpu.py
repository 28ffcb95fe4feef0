"""Standalone Pcs::open through the C ABI (lurkhip_open): commitments opened at caller-chosen points.
* the opened values are the interpolants' values at the points (oracle: inverse NTT + Horner in the extension field);
* the oracle's PCS verifier (oracle/stark.py pcs_verify: p3 TwoAdicFriPcs::verify + FRI verifier as recalled) accepts the
  opening from the same transcript state, and leaves its transcript where the prover's is;
* tampered values / a transcript that observed something else are rejected."""
from types import SimpleNamespace

import numpy as np
import pytest

from lurk_amd import commit as cm
from lurk_amd import prover, synth
from oracle import stark as os_

pytestmark = pytest.mark.gpu
P = os_.P


def eval_columns_at(mat, z):
    """values at the extension point z of the polynomials interpolating the columns of `mat` on the subgroup of its height"""
    log_n = mat.shape[0].bit_length() - 1
    out = []
    for c in range(mat.shape[1]):
        coeffs = os_.interpolate([int(x) for x in mat[:, c]], log_n)
        acc = os_.ZERO
        for a in reversed(coeffs):
            acc = os_.ef_add(os_.ef_mul(acc, z), os_.ef(a))
        out.append(acc)
    return out


def test_open_two_rounds(ctx, oracle):
    shapes = [[(8, 20), (6, 5), (8, 4)], [(7, 8)]]
    mats = [[synth.field_elements((1 << lg, w), seed=900 + 10 * r + i) for i, (lg, w) in enumerate(rs)] for r, rs in enumerate(shapes)]
    commits = [cm.commit(ctx, ms, log_blowup=1) for ms in mats]
    z = (123456789, 987654321 % P, 5, 1 << 30)
    u = (7, 0, 3, 11)

    def nxt(pt, log_n):
        return os_.ef_scale(pt, os_.two_adic_generator(log_n))

    # sphinx's pattern (zeta and zeta * w_N) for most matrices, a single point for one, an unrelated second point for the last round
    points = [[[z, nxt(z, 8)], [z, nxt(z, 6)], [z]], [[u, z]]]
    ch, och = prover.Challenger(ctx), os_.Challenger(os_.default_permute16())
    for c in commits:
        ch.observe(c.root)
        och.observe([int(x) for x in c.root])
    op = cm.open_rounds(ctx, commits, points, ch, num_queries=12, pow_bits=5)
    assert op.log_max_height == 9 and len(op.fri_roots) == 8 and op.num_queries == 12
    assert op.shapes == [[(lg, w, len(points[r][i])) for i, (lg, w) in enumerate(rs)] for r, rs in enumerate(shapes)]
    for r, rs in enumerate(mats):
        for i, m in enumerate(rs):
            for k, pt in enumerate(points[r][i]):
                assert op.opened[r][i][k] == eval_columns_at(m, pt), (r, i, k)

    def rounds_of(opening):
        return [([int(x) for x in c.root], [(lg, w, list(zip(points[r][i], opening.opened[r][i]))) for i, (lg, w) in enumerate(shapes[r])])
                for r, c in enumerate(commits)]

    os_.pcs_verify(rounds_of(op), op, 1, och, oracle.merkle_verify)
    assert ch.sample(4) == [och.sample() for _ in range(4)]  # both transcripts end in the same state

    # a changed opened value, a changed final polynomial and a transcript that saw other roots are all rejected
    def fresh():
        o = os_.Challenger(os_.default_permute16())
        for c in commits:
            o.observe([int(x) for x in c.root])
        return o

    bad = cm.parse_opening(op.words)
    v = list(bad.opened[0][1][0][2])
    v[1] = (v[1] + 1) % P
    bad.opened[0][1][0][2] = tuple(v)
    with pytest.raises(os_.VerifyError):
        os_.pcs_verify(rounds_of(bad), bad, 1, fresh(), oracle.merkle_verify)
    bad = cm.parse_opening(op.words)
    bad.final_poly = (bad.final_poly[0] ^ 1,) + tuple(bad.final_poly[1:])
    with pytest.raises(os_.VerifyError):
        os_.pcs_verify(rounds_of(bad), bad, 1, fresh(), oracle.merkle_verify)
    with pytest.raises(os_.VerifyError):
        os_.pcs_verify(rounds_of(op), op, 1, os_.Challenger(os_.default_permute16()), oracle.merkle_verify)
    for c in commits:
        c.close()


@pytest.mark.parametrize("widths", [(17, 18, 19), (20, 21, 33, 5), (130, 257), (64, 16, 127, 4)])
def test_open_odd_widths_through_the_rows_kernel(ctx, oracle, widths):
    """Matrices wider than 16 columns are reduced four lanes to a row in 16-byte pieces (fri.hip: k_reduce_openings_rows): every
    remainder of the width modulo 4 and modulo 16, several matrices of one height in a launch, narrow ones beside them; the oracle's
    verifier recomputes the reduced openings from the opened values and checks the FRI queries against them."""
    lg = 5
    mats = [synth.field_elements((1 << lg, w), seed=1300 + w) for w in widths]
    c = cm.commit(ctx, mats, log_blowup=1)
    z = (1234567, 7654321, 99, 1 << 29)
    zn = os_.ef_scale(z, os_.two_adic_generator(lg))
    points = [[[z, zn] if i % 2 == 0 else [z] for i in range(len(widths))]]
    ch, och = prover.Challenger(ctx), os_.Challenger(os_.default_permute16())
    ch.observe(c.root)
    och.observe([int(x) for x in c.root])
    op = cm.open_rounds(ctx, [c], points, ch, num_queries=8, pow_bits=2)
    for i, m in enumerate(mats):
        for k, pt in enumerate(points[0][i]):
            assert op.opened[0][i][k] == eval_columns_at(m, pt), (i, k)
    rounds = [([int(x) for x in c.root], [(lg, w, list(zip(points[0][i], op.opened[0][i]))) for i, w in enumerate(widths)])]
    os_.pcs_verify(rounds, op, 1, och, oracle.merkle_verify)
    c.close()


def test_open_argument_checks(ctx):
    c = cm.commit(ctx, [synth.field_elements((64, 3), seed=5)], log_blowup=1)
    ch = prover.Challenger(ctx)
    z = (1, 2, 3, 4)
    with pytest.raises(Exception, match="one or two opening points"):
        cm.open_rounds(ctx, [c], [[[z, z, z]]], ch, num_queries=2, pow_bits=1)
    with pytest.raises(Exception, match="coincide"):
        cm.open_rounds(ctx, [c], [[[z, z]]], ch, num_queries=2, pow_bits=1)
    with pytest.raises(Exception, match="canonical"):
        cm.open_rounds(ctx, [c], [[[(P, 0, 0, 0)]]], ch, num_queries=2, pow_bits=1)
    c.close()
