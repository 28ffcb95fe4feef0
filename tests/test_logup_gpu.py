"""Device twin of the reference's in-tree LogUp module (lurk_amd/csrc/logup.hip <- /root/reference/src/logup/, SURVEY.md 8a row
L1) against its restatement (oracle/logup.py), bit for bit: multiplicity traces, permutation traces (inclusive as upstream,
and exclusive), constraint values in both interaction orders.  Parity upstream: unpinned (dead code, no tests)."""
import numpy as np
import pytest

from logup_helpers import GAMMA, R, Z, ef_rows, system
from lurk_amd import logup as ll
from oracle import logup as ol
from oracle import stark as os_

pytestmark = pytest.mark.gpu
P = os_.P


@pytest.mark.parametrize("height", [4, 16, 1 << 11])
def test_traces_match_the_restatement(ctx, height):
    s = system(height, seed=height)
    mult = ol.multiplicities_trace(s["multiplicities"], Z)
    got_mult = ll.multiplicities_trace(ctx, s["multiplicities"], Z)
    assert got_mult.tolist() == ef_rows(mult)
    for exclusive in (False, True):
        want, wsum = ol.permutation_trace(s["identity"].tolist(), s["prov_prep"].tolist(), s["prov_main"].tolist(), mult, s["provides"], [], Z, R, GAMMA, exclusive)
        got, gsum = ll.permutation_trace(ctx, s["identity"], s["prov_prep"], s["prov_main"], got_mult, s["provides"], [], Z, R, GAMMA, exclusive)
        assert got.tolist() == ef_rows(want) and gsum == wsum
        empty = [[] for _ in range(height)]
        want_r, wsum_r = ol.permutation_trace(s["identity"].tolist(), None, s["req_main"].tolist(), empty, [], s["requires"], Z, R, GAMMA, exclusive)
        got_r, gsum_r = ll.permutation_trace(ctx, s["identity"], None, s["req_main"], None, [], s["requires"], Z, R, GAMMA, exclusive)
        assert got_r.tolist() == ef_rows(want_r) and gsum_r == wsum_r
        assert os_.ef_add(gsum, gsum_r) == os_.ZERO


def test_mixed_provides_and_requires_in_one_trace_and_constraints(ctx):
    """One trace that both provides and requires (multiplicity witnesses first, then -z), constraint values on real and on
    perturbed rows in both interaction orders."""
    h = 32
    s = system(h, seed=7)
    main = np.concatenate([s["prov_main"], s["req_main"]], axis=1)  # [is_real | i1 v1 i2 v2 flag]
    shift = lambda inter: ([([(k, i + 1 if k == ol.MAIN else i, w) for k, i, w in t], c) for t, c in inter[0]],
                           None if inter[1] is None else ([(k, i + 1 if k == ol.MAIN else i, w) for k, i, w in inter[1][0]], inter[1][1]))
    provides, requires = s["provides"], [shift(x) for x in s["requires"]]
    mult = ol.multiplicities_trace(s["multiplicities"], Z)
    for exclusive in (False, True):
        want, wsum = ol.permutation_trace(s["identity"].tolist(), s["prov_prep"].tolist(), main.tolist(), mult, provides, requires, Z, R, GAMMA, exclusive)
        got, gsum = ll.permutation_trace(ctx, s["identity"], s["prov_prep"], main, np.array(ef_rows(mult), dtype=np.uint32), provides, requires, Z, R, GAMMA, exclusive)
        assert got.tolist() == ef_rows(want) and gsum == wsum
        assert wsum == os_.ZERO  # the trace's own provides and requires balance
    rows, total = ol.permutation_trace(s["identity"].tolist(), s["prov_prep"].tolist(), main.tolist(), mult, provides, requires, Z, R, GAMMA, True)
    rng = np.random.default_rng(3)
    local = np.array(ef_rows(rows), dtype=np.uint64)
    nxt = np.roll(local, -1, axis=0)
    local[5, 2, 1] = (local[5, 2, 1] + 1) % P  # a wrong inverse
    nxt[9, 0, 0] = (nxt[9, 0, 0] + 5) % P      # a broken running sum
    sels = rng.integers(0, P, size=(h, 3))
    for air_order in (False, True):
        want = [ol.eval_constraints([tuple(int(x) for x in c) for c in local[i]], [tuple(int(x) for x in c) for c in nxt[i]], mult[i], i, s["prov_prep"][i].tolist(),
                                    main[i].tolist(), provides, requires, Z, R, GAMMA, total, tuple(int(x) for x in sels[i]), air_order) for i in range(h)]
        got = ll.eval_constraints(ctx, local, nxt, np.array(ef_rows(mult), dtype=np.uint32), s["identity"], s["prov_prep"], main, provides, requires, Z, R, GAMMA, total,
                                  sels, air_order)
        assert got.tolist() == ef_rows(want), air_order
        if not air_order:
            nz = [(i, k) for i in range(h) for k in range(len(want[i]) - 1) if want[i][k] != os_.ZERO and tuple(int(x) for x in sels[i]) != (0, 0, 0)]
            assert (5, 1) in nz  # the perturbed inverse is caught


def test_malformed_programs_are_refused(ctx):
    s = system(4)
    with pytest.raises(Exception):
        ll.permutation_trace(ctx, s["identity"], None, s["req_main"], None, [], [([([(ol.MAIN, 99, 1)], 0)], None)], Z, R, GAMMA)
