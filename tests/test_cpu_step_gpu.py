"""The CPU port of the whole step (oracle/cpu_prover.py: bench.py's `cpu_baseline`) against the HIP prover on the same
program: same traces, same transcript -> the SAME proof, field by field (roots, cumulative sums, opened values, FRI
commitments, final polynomial, proof-of-work witness, every query opening)."""
import numpy as np
import pytest

import lurk_amd
from lair_helpers import PARTIAL_SRC, load_cases
from lurk_amd import lair, prover
from oracle import binding as ob
from oracle import cpu_prover as cpv
from oracle import stark as os_
from test_cpu_step import bytes_preprocessed, machine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src,entry,args", [(load_cases()[0]["source"], "fib", [11]), (PARTIAL_SRC, "top", [10])], ids=["demo", "partial"])
def test_cpu_port_and_hip_prover_produce_the_same_proof(ctx, oracle, src, entry, args):
    top = lair.Toplevel(src)
    q = lair.QueryRecord(top)
    top.execute_by_name(entry, args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, entry, len(pv))
    root = m.setup()
    (gp,) = m.prove(q, num_queries=7, pow_bits=5)
    m.close()

    airs, names, opv, traces = machine(src, entry, args)
    assert opv == [int(x) for x in pv]
    pr = cpv.CpuProver(airs, names, len(pv), threads=8)
    prep_m, pc = pr.setup({len(airs) - 1: bytes_preprocessed()})
    assert pc["root"] == [int(x) for x in root]
    ch = os_.Challenger(os_.default_permute16())
    ch.observe(pc["root"])
    ch.observe(0)
    ch.observe(gp.main_root)
    ch.observe(opv)
    cp = pr.prove_shard(traces, prep_m, pc, opv, ch, num_queries=7, pow_bits=5)
    assert [(c.machine_index, c.log_n, c.width) for c in cp.chips] == [(c.machine_index, c.log_n, c.width) for c in gp.chips]
    assert cp.main_root == gp.main_root
    assert [tuple(c.cumulative_sum) for c in cp.chips] == [tuple(c.cumulative_sum) for c in gp.chips]
    assert (cp.perm_root, cp.quot_root) == (gp.perm_root, gp.quot_root)
    assert os_.verify_machine(airs, pc["root"], [16], [6], [cp], ob.merkle_verify)
    assert [(c.machine_index, c.log_n, c.width, c.prep_width, c.perm_width, c.quotient_degree, c.prep_index, tuple(c.cumulative_sum)) for c in cp.chips] == \
           [(c.machine_index, c.log_n, c.width, c.prep_width, c.perm_width, c.quotient_degree, c.prep_index, tuple(c.cumulative_sum)) for c in gp.chips]
    for a, b in zip(cp.chips, gp.chips):
        for key in ("main", "perm") + (("prep",) if "prep" in b.opened else ()):
            assert [list(map(tuple, x)) for x in a.opened[key]] == [list(map(tuple, x)) for x in b.opened[key]], (a.machine_index, key)
        assert [list(map(tuple, x)) for x in a.opened["quotient"]] == [list(map(tuple, x)) for x in b.opened["quotient"]]
    assert cp.fri_roots == gp.fri_roots and tuple(cp.final_poly) == tuple(gp.final_poly)
    assert cp.pow_witness == gp.pow_witness and cp.query_indices == gp.query_indices
    assert [rw for rw, _ in cp.round_openings] == [rw for rw, _ in gp.round_openings]
    for (_, ra), (_, rb) in zip(cp.round_openings, gp.round_openings):
        assert [list(r) for r in ra] == [list(r) for r in rb]
    # the GPU's layer records carry the pair (8 words) + the path, like the port's
    for (wa, ra), (wb, rb) in zip(cp.layer_openings, gp.layer_openings):
        assert wa == wb and [list(r) for r in ra] == [list(r) for r in rb]
