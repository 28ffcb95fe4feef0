"""GPU side of the compiled-toplevel exchange (tests/test_bytecode.py has the host side): a toplevel imported from "LBC1"
bytecode (lurkhip_toplevel_from_bytecode) -- what a host with its own compiler hands over instead of source text -- drives
the same trace programs, AIRs and proofs as the toplevel compiled from source: bit-identical traces, bit-identical proof
words, and the oracle's verifier accepts the proof."""
import numpy as np
import pytest

from lurk_amd import lair, prover
from lurk_amd.programs import lurk_mix as lm
from oracle import binding as ob
from oracle import stark as os_
from test_workloads_gpu import oracle_airs

pytestmark = pytest.mark.gpu


def test_imported_toplevel_gives_the_same_traces_and_proof(ctx):
    mix = lm.fib_mix(300)
    src_top = lair.Toplevel(mix.source, lurk_chips=True)
    imp_top = lair.Toplevel.from_bytecode(src_top.to_bytecode())
    results = []
    for top in (src_top, imp_top):
        q = lair.QueryRecord(top)
        top.execute_by_name(mix.entry, mix.main_args, q)
        pv = q.expect_public_values()
        m = prover.Machine(ctx, top, mix.entry, len(pv))
        root = m.setup()
        shard = lair.Shard.new(q)
        traces = [(air.name, t.cpu().numpy().copy()) for _, air, _, t in m.shard_traces(shard)]
        proofs = m.prove(q, None, num_queries=8, pow_bits=6)
        results.append((root, traces, proofs, pv))
        m.close()
    (root_a, tr_a, pr_a, pv_a), (root_b, tr_b, pr_b, pv_b) = results
    assert list(root_a) == list(root_b) and pv_a == pv_b
    assert [n for n, _ in tr_a] == [n for n, _ in tr_b] and len(tr_a) > 14
    for (name, a), (_, b) in zip(tr_a, tr_b):
        assert np.array_equal(a, b), name
    assert len(pr_a) == len(pr_b) == 1
    assert np.array_equal(pr_a[0].words, pr_b[0].words)
    assert os_.verify_machine(oracle_airs(mix, len(pv_b)), root_b, [16], [6], pr_b, ob.merkle_verify)
