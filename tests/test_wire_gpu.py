"""Proof wire format: the shard proofs of a Lurk-shaped machine re-encoded as the reference's
`bincode::serialize(&CryptoProof)` (/root/reference/src/core/cli/proofs.rs:22-35, repl.rs:200-203) by the C ABI
(lurkhip_crypto_proof_bincode), decoded by the oracle's independent reader (oracle/wire.py) and VERIFIED from the bytes alone:
  flat words -> bincode (product, C++) -> decode + verify_machine (oracle, Python).
Also the 44-lane public values the verifier rebuilds (proofs.rs:46-56, stark_machine.rs:16-17) and `CachedProof` with the ZDag
of its public data (proofs.rs:137-169, cli/zdag.rs:12-55)."""
import struct

import numpy as np
import pytest

from lurk_amd import lair, proofs, prover
from lurk_amd import zstore as zs
from lurk_amd.programs import lurk_mix as lm
from lurk_amd.zstore import BatchedZStore
from oracle import binding as ob
from oracle import stark as os_
from oracle import wire as ow
from test_workloads_gpu import oracle_airs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def proved(ctx):
    mix = lm.fib_mix(1 << 9)
    top = lair.Toplevel(mix.source, lurk_chips=True)
    q = lair.QueryRecord(top)
    top.execute_by_name(mix.entry, mix.main_args, q)
    pv = q.expect_public_values()
    m = prover.Machine(ctx, top, mix.entry, len(pv))
    root = m.setup()
    shard_proofs = m.prove(q, lair.ShardingConfig(1 << 7), num_queries=6, pow_bits=4)
    yield mix, m, root, shard_proofs, pv
    m.close()


def test_crypto_proof_bytes_verify(proved):
    mix, m, root, shard_proofs, pv = proved
    assert len(pv) == 44 and len(shard_proofs) == 4
    cp = proofs.CryptoProof.from_machine_proof(m, shard_proofs, verifier_version="0123abcd")
    data = cp.to_bytes()
    names = [air.name for _, _, air in m.chips]
    # the verifier rebuilds the public values: here the first 40 lanes are the synthetic machine's, the last 4 the depth bytes
    depth = sum(b << (8 * i) for i, b in enumerate(pv[-4:]))
    assert cp.depth == depth
    shards, version, got_depth = ow.decode_crypto_proof(data, names, lambda d: pv[:40] + [(d >> (8 * i)) & 0xFF for i in range(4)], pow_bits=4)
    assert (version, got_depth, len(shards)) == ("0123abcd", depth, 4)
    # field by field against the flat proof
    for s, p in zip(shards, shard_proofs):
        assert (s.main_root, s.perm_root, s.quot_root, s.fri_roots) == (p.main_root, p.perm_root, p.quot_root, p.fri_roots)
        assert (s.final_poly, s.pow_witness, s.num_queries, s.log_max_height) == (p.final_poly, p.pow_witness, p.num_queries, p.log_max_height)
        assert [(c.machine_index, c.log_n, c.width, c.prep_width, c.perm_width, c.quotient_degree, c.prep_index, c.cumulative_sum) for c in s.chips] == \
               [(c.machine_index, c.log_n, c.width, c.prep_width, c.perm_width, c.quotient_degree, c.prep_index, c.cumulative_sum) for c in p.chips]
        for a, b in zip(s.chips, p.chips):
            assert a.opened["main"] == tuple(b.opened["main"]) or list(a.opened["main"]) == list(b.opened["main"])
            assert [list(x) for x in a.opened["quotient"]] == [list(x) for x in b.opened["quotient"]]
        assert [r[0] for r in s.round_openings] == [r[0] for r in p.round_openings]
    # ... and the oracle's verifier accepts the machine proof decoded from the bytes alone
    assert os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], shards, ob.merkle_verify)
    # the Montgomery spelling of the field elements decodes to the same proof
    mont = proofs.CryptoProof.from_machine_proof(m, shard_proofs, verifier_version="0123abcd", serialize_montgomery=True).to_bytes()
    assert mont != data and len(mont) == len(data)
    shards_m, _, _ = ow.decode_crypto_proof(mont, names, lambda d: pv[:40] + [(d >> (8 * i)) & 0xFF for i in range(4)], pow_bits=4, montgomery=True)
    assert [s.main_root for s in shards_m] == [s.main_root for s in shards] and shards_m[0].round_openings == shards[0].round_openings
    # a flipped field element inside an opening is caught by the verifier, a truncated file by the decoder
    bad = bytearray(data)
    off = 8 + 3 * 32 + 8 + 8 + 8 + 8 + 4  # shard 0: roots, #chips, chip 0: empty prep.local / prep.next lengths, main.local length, first lane
    (v,) = struct.unpack_from("<I", bad, off)
    struct.pack_into("<I", bad, off, (v + 1) % os_.P)
    shards_bad, _, _ = ow.decode_crypto_proof(bytes(bad), names, lambda d: pv[:40] + [(d >> (8 * i)) & 0xFF for i in range(4)], pow_bits=4)
    with pytest.raises(os_.VerifyError):
        os_.verify_machine(oracle_airs(mix, len(pv)), root, [16], [6], shards_bad, ob.merkle_verify)
    with pytest.raises(Exception):
        ow.decode_crypto_proof(data[:-3], names, lambda d: pv, pow_bits=4)
    # the product's own verifier, from the bytes alone (csrc/verify.cpp: lurkhip_crypto_proof_verify): both spellings accepted, the
    # flipped element, a truncation, other public values and another query count rejected
    from lurk_amd.profile import ProtocolProfile

    assert proofs.verify_crypto_proof(m, data, pv, num_queries=6, pow_bits=4)
    mprof = ProtocolProfile.preset("default")
    mprof.serialize_montgomery = 1
    assert proofs.verify_crypto_proof(m, mont, pv, num_queries=6, pow_bits=4, profile=mprof)
    for bad_data, bad_pv, nq in ((bytes(bad), pv, 6), (data[:-3], pv, 6), (data, pv[:7] + [(pv[7] + 1) % os_.P] + pv[8:], 6), (data, pv, 7), (mont, pv, 6)):
        with pytest.raises(prover.VerificationError):
            proofs.verify_crypto_proof(m, bad_data, bad_pv, num_queries=nq, pow_bits=4)


def test_bincode_framing(proved):
    """The parts of the layout that are the reference's own (not recalled): Vec length prefix, String, trailing u32 depth."""
    mix, m, root, shard_proofs, pv = proved
    data = proofs.CryptoProof.from_machine_proof(m, shard_proofs[:1], verifier_version="abc").to_bytes()
    assert struct.unpack_from("<Q", data, 0)[0] == 1                       # shard_proofs: Vec -> u64 length
    assert data[-4 - 3 - 8:-4] == struct.pack("<Q", 3) + b"abc"           # verifier_version: String
    assert struct.unpack_from("<I", data, len(data) - 4)[0] == sum(b << (8 * i) for i, b in enumerate(pv[-4:]))  # depth: u32
    assert list(struct.unpack_from("<8I", data, 8)) == shard_proofs[0].main_root  # commitment.main_commit: [F; 8], canonical u32


def test_cached_proof_and_public_values(ctx, proved):
    mix, m, root, shard_proofs, pv = proved
    st = BatchedZStore(ctx)
    b = st.batch()
    expr = b.syntax(zs.syn_list(zs.syn_builtin("cons"), zs.syn_u64(1), zs.syn_u64(2)))
    env = b.env(b.symbol((zs.USER_PACKAGE, "x")), b.num(5), b.empty_env())
    result = b.cons(b.u64(1), b.u64(2))
    b.run()
    lanes = proofs.public_values(b[expr], b[env], b[result], 0x01020304)
    assert len(lanes) == 44 and lanes[0] == zs.TAG["Cons"] and lanes[1:8] == [0] * 7 and lanes[16:24] == list(b[env].digest)
    assert lanes[24] == zs.TAG["Cons"] and lanes[40:] == [4, 3, 2, 1]
    cp = proofs.CryptoProof.from_machine_proof(m, shard_proofs, verifier_version="v")
    cached = proofs.CachedProof(cp, b[expr], b[env], b[result], st)
    data = cached.to_bytes()
    names = [air.name for _, _, air in m.chips]
    shards, version, depth, e, v, r, zdag = ow.decode_cached_proof(data, names, pow_bits=4)
    assert (e, v, r) == tuple((z.tag, z.digest) for z in (b[expr], b[env], b[result]))
    assert len(shards) == 4 and version == "v" and depth == cp.depth
    assert [(z, k, kids) for z, k, kids in zdag] == [((z.tag, z.digest), k, [(c.tag, c.digest) for c in kids]) for z, k, kids in cached.zdag]
    assert len(zdag) > 20 and zdag[-1][0] == (b[result].tag, b[result].digest)
    # the decoder rebuilt the public values from expr / env / result / depth exactly as into_machine_proof does
    assert shards[0].public_values == proofs.public_values(b[expr], b[env], b[result], depth)
    st.close()


def test_cached_proof_verifies_from_bytes_alone(ctx):
    """The reference CLI's `verify`: a machine whose 44 public values have the claim's shape (expr flat 16 | env digest 8 | result
    flat 16 | depth bytes) is proved, stored as CachedProof bytes and verified by the product from those bytes alone
    (lurkhip_cached_proof_verify rebuilds the public values from expr / env / result / depth); another claim is rejected."""
    import ctypes as C

    from lurk_amd import _native as N
    from lurk_amd.zstore import ZPtr

    src = """
    partial fn lurk_main(e: [16], v: [8]): [16] {
        let (t, z1, z2, z3, z4, z5, z6, z7, d0, d1, d2, d3, d4, d5, d6, d7) = e;
        let tag = 1;
        let z = 0;
        let (v0, v1, v2, v3, v4, v5, v6, v7) = v;
        let s0 = add(d0, v0);
        let p = store(d1, v1, t);
        let (l0, l1, l2) = load(p);
        let s1 = mul(l0, l1);
        return (tag, z, z, z, z, z, z, z, s0, s1, d2, d3, v4, v5, d6, v7)
    }
    """
    top = lair.Toplevel(src)
    q = lair.QueryRecord(top)
    expr = ZPtr(11, (5, 6, 7, 8, 9, 10, 11, 12))
    env = ZPtr(12, (21, 22, 23, 24, 25, 26, 27, 28))
    top.execute_by_name("lurk_main", expr.flatten() + list(env.digest), q)
    pv = q.expect_public_values()
    assert len(pv) == 44
    result = ZPtr(pv[24], tuple(pv[32:40]))
    m = prover.Machine(ctx, top, "lurk_main", 44)
    m.setup()
    shard_proofs = m.prove(q, num_queries=7, pow_bits=5)
    cp = proofs.CryptoProof.from_machine_proof(m, shard_proofs, verifier_version="deadbeef").to_bytes()

    def cached(e, v, r):
        zp = [np.asarray([z.tag] + list(z.digest), dtype=np.uint32) for z in (e, v, r)]
        p = C.c_void_p
        args = (cp, len(cp), zp[0].ctypes.data_as(p), zp[1].ctypes.data_as(p), zp[2].ctypes.data_as(p), 0, None, 0)
        size = N.lib.lurkhip_cached_proof_bincode(*args, None, 0)
        buf = (C.c_uint8 * size)()
        assert N.lib.lurkhip_cached_proof_bincode(*args, C.cast(buf, p), size) == size
        return bytes(buf)

    assert proofs.verify_cached_proof(m, cached(expr, env, result), num_queries=7, pow_bits=5) == pv
    for other in (cached(ZPtr(10, expr.digest), env, result), cached(expr, ZPtr(12, (1,) * 8), result), cached(expr, env, ZPtr(1, (0,) * 8)),
                  cached(expr, env, result)[:-1]):
        with pytest.raises(prover.VerificationError):
            proofs.verify_cached_proof(m, other, num_queries=7, pow_bits=5)
    m.close()


def test_the_example_script_runs():
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "prove_and_verify.py")
    spec = importlib.util.spec_from_file_location("example_prove_and_verify", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a, b = 0, 1
    for _ in range(300):
        a, b = b, (a + b) % 2013265921
    assert mod.main(300) == a
