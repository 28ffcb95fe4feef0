"""The product's host verifier (csrc/verify.cpp: lurkhip_machine_verify, no device) on proofs made WITHOUT the GPU: the CPU port
of the prover (oracle/cpu_prover.py) proves small machines, the proofs are written in the product's word format and the product
verifier must accept them, reject every tampered variant the oracle's verifier rejects, and agree with it under a non-default
protocol profile.  (tests/test_prover_gpu.py runs it on the HIP prover's proofs.)"""
import copy

import numpy as np
import pytest

import upstream_helpers as uh
from lair_helpers import PARTIAL_SRC, load_cases
from lurk_amd import lair, prover
from lurk_amd.air import ChipAir
from lurk_amd.profile import ProtocolProfile
from oracle import binding as ob
from oracle import cpu_prover as cpv
from oracle import lair as ol
from oracle import stark as os_
from proof_words import encode_words
from test_cpu_step import machine, prove

P = os_.P


def product_airs(src, entry, n_public, lurk_chips=False):
    top = lair.Toplevel(src, lurk_chips=lurk_chips)
    airs = [ChipAir.for_entrypoint(top.func_index(entry), n_public)]
    airs += [ChipAir.for_func(top, i) for i in range(top.num_funcs())]
    airs += [ChipAir.for_mem(ml) for ml in lair.MEM_TABLE_SIZES]
    airs.append(ChipAir.for_bytes())
    return airs


@pytest.fixture(scope="module")
def proved(oracle):
    src = load_cases()[0]["source"]
    airs, names, pv, traces = machine(src, "fib", [9])
    pr = cpv.CpuProver(airs, names, len(pv), threads=4)
    shard, vk = prove(pr, traces, pv)
    return src, airs, pv, shard, vk


def test_round_trip_of_the_word_format(proved):
    _, _, _, shard, _ = proved
    words = encode_words(shard)
    back = prover.parse_proof(words)
    assert encode_words(back).tolist() == words.tolist()


def test_accepts_the_cpu_provers_proof_and_rejects_what_the_oracle_rejects(proved):
    src, oairs, pv, shard, vk = proved
    pairs = product_airs(src, "fib", len(pv))
    assert prover.verify_machine_proof(pairs, vk, [16], [6], [encode_words(shard)])

    def both_reject(mutate, what):
        bad = copy.deepcopy(shard)
        mutate(bad)
        with pytest.raises(os_.VerifyError):
            os_.verify_machine(oairs, vk, [16], [6], [bad], ob.merkle_verify)
        with pytest.raises(prover.VerificationError):
            prover.verify_machine_proof(pairs, vk, [16], [6], [encode_words(bad)])

    def bump(t, k=0):
        t = list(t)
        t[k] = (t[k] + 1) % P
        return tuple(t)

    def m_opened(b):
        b.chips[1].opened["main"][0][0] = bump(b.chips[1].opened["main"][0][0])

    def m_next(b):
        b.chips[1].opened["main"][1][2] = bump(b.chips[1].opened["main"][1][2], 3)

    def m_perm(b):
        b.chips[2].opened["perm"][0][1] = bump(b.chips[2].opened["perm"][0][1])

    def m_quot(b):
        b.chips[0].opened["quotient"][0][0] = bump(b.chips[0].opened["quotient"][0][0])

    def m_cumsum(b):
        b.chips[1].cumulative_sum = bump(b.chips[1].cumulative_sum)

    def m_pow(b):
        b.pow_witness += 1

    def m_final(b):
        b.final_poly = bump(b.final_poly, 2)

    def m_root(b):
        b.perm_root = list(bump(b.perm_root, 5))

    def m_fri_root(b):
        b.fri_roots[1] = list(bump(b.fri_roots[1], 7))

    def m_public(b):
        b.public_values = list(bump(b.public_values, 1))

    def m_path(b):
        rw, recs = b.round_openings[1]
        recs = [list(r) for r in recs]
        recs[2][-3] = (recs[2][-3] + 1) % P
        b.round_openings[1] = (rw, recs)

    def m_layer(b):
        rw, recs = b.layer_openings[0]
        recs = [list(r) for r in recs]
        recs[0][5] = (recs[0][5] + 1) % P
        b.layer_openings[0] = (rw, recs)

    for k, m in enumerate((m_opened, m_next, m_perm, m_quot, m_cumsum, m_pow, m_final, m_root, m_fri_root, m_public, m_path, m_layer)):
        both_reject(m, k)
    # malformed words are rejections too, never crashes
    words = encode_words(shard)
    for bad in (words[:-1], words[:200], np.concatenate([words, [0]]).astype(np.uint32), np.zeros(4, dtype=np.uint32)):
        with pytest.raises(prover.VerificationError):
            prover.verify_machine_proof(pairs, vk, [16], [6], [bad])
    flipped = words.copy()
    flipped[40] = 0xFFFFFFFF  # not a canonical field element
    with pytest.raises(prover.VerificationError):
        prover.verify_machine_proof(pairs, vk, [16], [6], [flipped])
    # a wrong verifying key / a different machine
    with pytest.raises(prover.VerificationError):
        prover.verify_machine_proof(pairs, [(x + 1) % P for x in vk], [16], [6], [words])
    with pytest.raises(prover.VerificationError):
        prover.verify_machine_proof(product_airs(PARTIAL_SRC, "top", len(pv)), vk, [16], [6], [words])


def test_forged_openings_of_a_one_row_chip_are_rejected(proved):
    """a chip proved at height 1 -- the entrypoint chip, always -- has LDE height 2^log_blowup, below the first
    height the FRI query chain folds in, so its reduced openings were never checked: any opened values with a matching forged
    quotient passed both verifiers (the default profile does not observe opened values into the transcript).  Forge exactly that --
    garbage opened values of the entrypoint chip, quotient chosen so that constraints(zeta) / Z_H(zeta) == quotient(zeta) -- and
    require both verifiers to reject it: the reduced opening at height 2^log_blowup must be zero."""
    src, oairs, pv, shard, vk = proved
    pairs = product_airs(src, "fib", len(pv))
    bad = copy.deepcopy(shard)
    k = next(i for i, c in enumerate(bad.chips) if c.log_n == 0)
    chip = bad.chips[k]
    # replay the transcript up to zeta (oracle/stark.py: verify_machine, verify_shard)
    ch = os_.Challenger(os_.default_permute16(), None)
    ch.observe(vk)
    ch.observe(0)
    ch.observe(bad.main_root)
    ch.observe(bad.public_values)
    perm_alpha, perm_beta = ch.sample_ext(), ch.sample_ext()
    ch.observe(bad.perm_root)
    alpha = ch.sample_ext()
    ch.observe(bad.quot_root)
    zeta = ch.sample_ext()
    chip.opened["main"] = tuple([tuple((7 * j + 3 * e + side + 1) % P for e in range(4)) for j in range(chip.width)] for side in (0, 1))
    sels = os_.selectors_at_point(zeta, 0)
    folded = os_.eval_constraints_at(oairs[chip.machine_index], chip, sels, alpha, perm_alpha, perm_beta, bad.public_values)
    target = os_.ef_mul(folded, sels[3])
    # the recomputed quotient is affine in the first word of the first chunk: solve for it
    zero4 = [os_.ZERO] * 4
    chip.opened["quotient"] = [list(zero4) for _ in chip.opened["quotient"]]
    t0 = os_.recompute_quotient(chip, zeta)
    chip.opened["quotient"][0][0] = os_.ONE
    t1 = os_.recompute_quotient(chip, zeta)
    chip.opened["quotient"][0][0] = os_.ef_mul(os_.ef_sub(target, t0), os_.ef_inv(os_.ef_sub(t1, t0)))
    assert os_.recompute_quotient(chip, zeta) == target  # the forgery passes the constraint check at zeta ...
    with pytest.raises(os_.VerifyError, match="reduced opening"):  # ... and is caught by the opening argument alone
        os_.verify_machine(oairs, vk, [16], [6], [bad], ob.merkle_verify)
    with pytest.raises(prover.VerificationError, match="reduced opening"):
        prover.verify_machine_proof(pairs, vk, [16], [6], [encode_words(bad)])


def test_partial_machine_with_the_preprocessed_round(oracle):
    airs, names, pv, traces = machine(PARTIAL_SRC, "top", [9])
    pr = cpv.CpuProver(airs, names, len(pv), threads=4)
    shard, vk = prove(pr, traces, pv)
    assert any(c.prep_index == 0 for c in shard.chips)
    assert prover.verify_machine_proof(product_airs(PARTIAL_SRC, "top", len(pv)), vk, [16], [6], [encode_words(shard)])


def test_verification_from_the_reference_wire_format(oracle):
    """CryptoProof bytes (lurk_amd/csrc/wire.cpp) of a CPU-port proof back through the product's decoder and verifier: the wire
    format drops the query indices, the queried half of every FRI pair, the public values and every size that a length prefix
    carries -- the verifier re-derives them."""
    import ctypes as C

    from lurk_amd import _native as N
    from lurk_amd import proofs

    airs, names, pv, traces = machine(PARTIAL_SRC, "top", [9])
    assert len(pv) >= 4
    pr = cpv.CpuProver(airs, names, len(pv), threads=4)
    shard, vk = prove(pr, traces, pv)
    pairs = product_airs(PARTIAL_SRC, "top", len(pv))
    cp = proofs.CryptoProof([encode_words(shard)], [a.name for a in pairs], verifier_version="test")
    data = cp.to_bytes()

    def run(data, pv, nq=shard.num_queries, pow_bits=shard.pow_bits):
        air_ptrs = (C.c_void_p * len(pairs))(*[a.handle for a in pairs])
        cnames = (C.c_char_p * len(pairs))(*[a.name.encode() for a in pairs])
        vk_a, lh, ws, pv_a = (np.array(x, dtype=np.uint32) for x in (vk, [16], [6], list(pv)))
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        err = C.create_string_buffer(256)
        st = N.lib.lurkhip_crypto_proof_verify(None, C.cast(air_ptrs, C.c_void_p), C.cast(cnames, C.c_void_p), len(pairs), vk_a.ctypes.data, lh.ctypes.data,
                                               ws.ctypes.data, 1, C.cast(buf, C.c_void_p), len(data), pv_a.ctypes.data, pv_a.size, nq, pow_bits, 1, err, 256)
        return st, err.value.decode()

    assert run(data, pv) == (N.OK, "")
    flipped = bytearray(data)
    flipped[8 + 40] ^= 1  # inside the permutation root
    for bad, bad_pv, kw in ((bytes(flipped), pv, {}), (data[:-1], pv, {}), (data + b"\0", pv, {}), (data, pv[:-1] + [(pv[-1] + 1) % 256], {}),
                            (data, pv, {"nq": shard.num_queries + 1}), (data, pv, {"pow_bits": shard.pow_bits + 9})):
        st, why = run(bad, bad_pv, **kw)
        assert st == -8 and why, (st, why)
    # byte-level fuzz of the decoder: flips, truncations, length fields blown up -- a status, never a crash (tools/asan_host.sh runs
    # this under AddressSanitizer); only a flip inside the informational version string may still verify
    import random

    rnd = random.Random(5)
    accepted = 0
    for _ in range(400):
        b = bytearray(data)
        k = rnd.random()
        if k < 0.5:
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif k < 0.7:
            del b[rnd.randrange(len(b)):]
        elif k < 0.9:
            at = rnd.randrange(len(b) - 8)
            b[at:at + 8] = rnd.choice([b"\xff" * 8, (1 << 40).to_bytes(8, "little"), (0).to_bytes(8, "little")])
        else:
            at = rnd.randrange(len(b))
            b[at:at] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 9)))
        st, why = run(bytes(b), pv)
        assert st in (N.OK, -8), (st, why)
        accepted += st == N.OK
    assert accepted <= 8
