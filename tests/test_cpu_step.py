"""The CPU port of the whole proving step (oracle/cpu_step.c + oracle/cpu_emit.py + oracle/cpu_prover.py: bench.py's
`cpu_baseline`), checked against the oracle's Python definition (oracle/stark.py) stage by stage, word for word, and as a
whole: the oracle's verifier accepts its proofs and rejects tampered ones.  No GPU (tests/test_cpu_step_gpu.py compares it
with the HIP prover)."""
import numpy as np
import pytest

import upstream_helpers as uh
from lair_helpers import PARTIAL_SRC, load_cases
from oracle import air as oa
from oracle import binding as ob
from oracle import cpu_prover as cpv
from oracle import lair as ol
from oracle import stark as os_

P = os_.P


def bytes_preprocessed():
    i = np.arange(1 << 16)
    i1, i2 = i & 0xFF, i >> 8
    return np.stack([i1, i2, (i1 < i2).astype(int), i1 & i2, i1 ^ i2, i1 | i2], axis=1).astype(np.uint32)


def machine(src, entry, args, lurk_chips=False):
    """Oracle machine of `entry(args)`: (airs, names, public values, traces [(machine index, canonical matrix)])."""
    otop, oq = uh.oracle_machine(src, entry, args, lurk_chips)
    pv = [int(x) for x in oq.public_values]
    airs, names = uh.oracle_airs_and_names(otop, entry, len(pv))
    traces = [(0, np.array([pv], dtype=np.uint32))]
    mi = 1
    for f in otop.funcs:
        if oq.func[f["index"]]:  # LairChip::included: a function chip takes part iff it has queries (lair_chip.rs:124-129)
            rows, _ = ol.generate_trace(otop, f["name"], oq)
            traces.append((mi, np.array(rows, dtype=np.uint32)))
        mi += 1
    for ml in ol.MEM_TABLE_SIZES:
        traces.append((mi, np.array(ol.mem_trace(oq, ml), dtype=np.uint32)))
        mi += 1
    traces.append((mi, np.array(ol.bytes_trace(oq, 0), dtype=np.uint32)))
    return airs, names, pv, traces


@pytest.fixture(scope="module")
def demo(oracle):
    src = load_cases()[0]["source"]
    airs, names, pv, traces = machine(src, "fib", [9])
    prover = cpv.CpuProver(airs, names, len(pv), threads=4)
    return airs, names, pv, traces, prover


def nat_lde(rows):
    return os_.coset_lde([[int(x) for x in r] for r in rows], 1)


def test_stages_match_the_python_definition(demo):
    airs, names, pv, traces, pr = demo
    L = pr.L
    pub_m = cpv.to_m(np.array(pv + [0], dtype=np.uint64))
    alpha, beta, fold_alpha = (11, 22, 33, 44), (5, 6, 7, P - 1), (3, 1, 4, 1)
    checked = 0
    for mi, t in traces:
        air = airs[mi]
        if getattr(air, "prep_width", 0) or t.shape[0] > 64:
            continue  # the byte table (2^16 rows) is covered by the whole-proof test: the Python LDE below is O(N^2)-ish slow
        # ---- permutation trace
        m = pr.monty(t)
        pw = L.cp2_perm_width(mi, 2)
        out = np.empty((t.shape[0], 4 * pw), dtype=np.uint32)
        cs = np.zeros(4, dtype=np.uint32)
        import ctypes as C
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert L.cp2_perm_trace(mi, t.shape[0].bit_length() - 1, p(m), None, p(pub_m), p(cpv._ef_m(alpha)), p(cpv._ef_m(beta)), 2, p(out), p(cs)) == 0
        want = os_.permutation_trace(air, t.tolist(), None, alpha, beta, 2, public=pv)
        assert cpv.from_m(out).reshape(-1).tolist() == [x for r in want for c in r for x in c], names[mi]
        assert tuple(int(x) for x in cpv.from_m(cs)) == want[-1][-1]
        # ---- LDE (committed order) == the Python coset LDE, bit-reversed
        lde = pr.lde(m)
        want_lde = os_.bit_reverse_rows(nat_lde(t))
        assert cpv.from_m(lde).tolist() == want_lde
        # ---- quotient chunks
        perm_lde = pr.lde(out)
        qout = np.empty((2, t.shape[0], 4), dtype=np.uint32)
        lg = t.shape[0].bit_length() - 1
        assert L.cp2_quotient(mi, lg, p(lde), None, p(perm_lde), p(pub_m), p(cpv._ef_m(alpha)), p(cpv._ef_m(beta)), p(cpv._ef_m(fold_alpha)), p(cs), 1, p(qout)) == 0
        perm_nat = nat_lde(cpv.from_m(out))
        perm_nat = [[tuple(r[4 * j:4 * j + 4]) for j in range(pw)] for r in perm_nat]
        want_q = os_.quotient_chunks(air, lg, nat_lde(t), None, perm_nat, alpha, beta, fold_alpha, want[-1][-1], public=pv, lqd=1)
        assert cpv.from_m(qout).tolist() == [[list(v) for v in chunk] for chunk in want_q], names[mi]
        # ---- opened values: the polynomial of column c at an extension point, by definition (interpolate + Horner)
        z = (123, 456, 789, 1011)
        zs = np.ascontiguousarray(np.stack([cpv._ef_m(z)]))
        ov = np.empty((1, t.shape[1], 4), dtype=np.uint32)
        assert L.cp2_open(lg, t.shape[1], p(lde), 1, p(zs), p(ov)) == 0
        for c in range(0, t.shape[1], max(1, t.shape[1] // 5)):
            coef = os_.interpolate([int(x) for x in t[:, c]], lg)
            acc = os_.ZERO
            for k in reversed(coef):
                acc = os_.ef_add(os_.ef_mul(acc, z), os_.ef(k))
            assert tuple(int(x) for x in cpv.from_m(ov[0, c])) == acc
        checked += 1
    assert checked >= 3


def verify(airs, prover_out, vk_root):
    return os_.verify_machine(airs, vk_root, [16], [6], prover_out, ob.merkle_verify)


def prove(pr, traces, pv, num_queries=5, pow_bits=4, timings=None):
    prep_m, pc = pr.setup({len(pr.airs) - 1: bytes_preprocessed()})
    ch = os_.Challenger(os_.default_permute16())
    ch.observe(pc["root"])
    ch.observe(0)
    # phase 1 of machine.prove: the shard's main root (recomputed inside prove_shard) and the public values
    main_root = pr.commit([pr.lde(pr.monty(t)) for _, t in pr.prover_order(traces)])[1]
    ch.observe(main_root)
    ch.observe(pv)
    shard = pr.prove_shard(traces, prep_m, pc, pv, ch, num_queries=num_queries, pow_bits=pow_bits, timings=timings)
    assert shard.main_root == main_root
    return shard, pc["root"]


def test_whole_proof_is_accepted_by_the_oracle_verifier_and_tampering_is_not(demo):
    airs, names, pv, traces, pr = demo
    tm = {}
    shard, vk = prove(pr, traces, pv, timings=tm)
    assert vk == uh.oracle_vk_root()
    assert verify(airs, [shard], vk)
    assert set(tm) >= {"commit_main", "permutation", "commit_perm", "quotient_all", "commit_quotient", "open", "fri_commit", "pow", "fri_query"}
    # a flipped opened value, a flipped cumulative sum, a wrong witness
    import copy

    bad = copy.deepcopy(shard)
    v = list(bad.chips[1].opened["main"][0][0])
    v[0] = (v[0] + 1) % P
    bad.chips[1].opened["main"][0][0] = tuple(v)
    with pytest.raises(os_.VerifyError):
        verify(airs, [bad], vk)
    bad = copy.deepcopy(shard)
    bad.pow_witness += 1
    with pytest.raises(os_.VerifyError):
        verify(airs, [bad], vk)
    bad = copy.deepcopy(shard)
    bad.final_poly = tuple((x + 1) % P for x in bad.final_poly)
    with pytest.raises(os_.VerifyError):
        verify(airs, [bad], vk)


def test_partial_functions_with_byte_lookups_prove(oracle):
    """A machine whose functions are `partial` (depth columns, byte-table requires) and use memory: the byte chip's
    preprocessed round takes part."""
    airs, names, pv, traces = machine(PARTIAL_SRC, "top", [9])
    pr = cpv.CpuProver(airs, names, len(pv), threads=4)
    shard, vk = prove(pr, traces, pv)
    assert verify(airs, [shard], vk)
    assert any(c.prep_index == 0 for c in shard.chips)
