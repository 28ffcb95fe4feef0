"""The restatement of the reference's dead LogUp module (oracle/logup.py <- /root/reference/src/logup/): nothing upstream pins
it (no tests, not compiled), so what is checked here is that the restatement has the properties the code's own comments
claim -- and where the code disagrees with itself, which variant satisfies its constraints.  No GPU."""
import pytest

from logup_helpers import GAMMA, R, Z, system
from oracle import logup as ol
from oracle import stark as os_

P = os_.P


def test_multiplicities_are_z_powers_weighted_counts():
    m = ol.multiplicities_trace([([0, 2], [[1, 2], [0, 5]]), ([1], [[7], [0]])], Z)
    z1, z2, z3 = Z, os_.ef_mul(Z, Z), os_.ef_mul(os_.ef_mul(Z, Z), Z)
    assert m[0][0] == os_.ef_add(os_.ef_scale(z1, 1), os_.ef_scale(z3, 2))      # traces 0 and 2 -> z^1, z^3 (the table upstream is one short)
    assert m[1][0] == os_.ef_scale(z3, 5) and m[0][1] == os_.ef_scale(z2, 7) and m[1][1] == os_.ZERO


def test_sums_of_a_balanced_system_cancel_and_constraints_hold_for_the_consistent_variant():
    s = system()
    mult = ol.multiplicities_trace(s["multiplicities"], Z)
    prov_rows, prov_sum = ol.permutation_trace(s["identity"].tolist(), s["prov_prep"].tolist(), s["prov_main"].tolist(), mult, s["provides"], [], Z, R, GAMMA)
    empty_mult = [[] for _ in range(s["height"])]
    req_rows, req_sum = ol.permutation_trace(s["identity"].tolist(), None, s["req_main"].tolist(), empty_mult, [], s["requires"], Z, R, GAMMA)
    # provide side: sum m_k / d_k with m = z * count; require side: sum -z / d per real lookup: the two cancel
    assert os_.ef_add(prov_sum, req_sum) == os_.ZERO and prov_sum != os_.ZERO
    # column 1 + k holds the INVERSE of d_k (not m_k / d_k), 0 where is_real is 0
    for i, row in enumerate(req_rows):
        d0 = ol.interaction_denominator(s["requires"][0], i, (), s["req_main"][i].tolist(), R, GAMMA)
        assert os_.ef_mul(row[1], d0) == os_.ONE
        assert (row[2] == os_.ZERO) == (int(s["req_main"][i][4]) == 0)
    # inclusive running sum as upstream computes it (trace.rs:142-148): the first-row constraint s_0 = 0 fails on it ...
    h = s["height"]
    sel = lambda i: (1 if i == 0 else 0, 1 if i == h - 1 else 0, 0 if i == h - 1 else 1)
    cons0 = ol.eval_constraints(req_rows[0], req_rows[1], [], 0, (), s["req_main"][0].tolist(), [], s["requires"], Z, R, GAMMA, req_sum, sel(0), air_order=False)
    assert cons0[-3] != os_.ZERO
    # ... the exclusive variant satisfies the inverse, first-row and transition constraints on every row
    ex_rows, ex_sum = ol.permutation_trace(s["identity"].tolist(), None, s["req_main"].tolist(), empty_mult, [], s["requires"], Z, R, GAMMA, exclusive=True)
    assert ex_sum == req_sum
    for i in range(h):
        c = ol.eval_constraints(ex_rows[i], ex_rows[(i + 1) % h], [], i, (), s["req_main"][i].tolist(), [], s["requires"], Z, R, GAMMA, ex_sum, sel(i), air_order=False)
        assert all(v == os_.ZERO for v in c[:-1]), i
    # (the last-row constraint compares the row's own t with the final sum, air.rs:73-76: it holds only for a one-row trace)
