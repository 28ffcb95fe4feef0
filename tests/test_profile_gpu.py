"""The protocol profile on the GPU: flipping ONE field changes the HIP prover and the oracle together -- the proof made under
the flipped profile differs from the default one, is accepted by the oracle verifier running the mirrored profile, and is
rejected by the oracle running the default profile (and vice versa).  Plus the GPU side of the upstream vector harness."""
import numpy as np
import pytest

import lurk_amd
import upstream_helpers as uh
from lair_helpers import PARTIAL_SRC
from lurk_amd import commit as lcommit
from lurk_amd import lair, prover
from lurk_amd.profile import ProtocolProfile
from oracle import air as oa
from oracle import binding as ob
from oracle import lair as ol
from oracle import stark as os_

pytestmark = pytest.mark.gpu
P = 2013265921

VARIANTS = {
    "squeeze16": {"challenger_squeeze": 16},
    "pop_front": {"challenger_pop_front": 1},
    "observe_openings": {"observe_openings": 1},
    "observe_chip_meta": {"observe_chip_meta": 1},
    "alpha_ascending": {"constraint_alpha_ascending": 1},
    "fri_alpha_global": {"fri_alpha_global": 1},
    "monty_diffusion": "p3-monty-diffusion",
    "hardened": "hardened",
}


def profiles_for(variant):
    """(library profile, oracle profile) of a variant."""
    v = VARIANTS[variant]
    lib = ProtocolProfile.preset(v) if isinstance(v, str) else ProtocolProfile.from_dict(v)
    return lib, os_.Profile.from_dict(lib.to_dict())


def oracle_airs(src, entry, n_public):
    otop = ol.Toplevel(src)
    airs = [oa.EntrypointAir(otop.index[entry], n_public)] + [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    return airs + [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES] + [oa.BytesAir()]


def prove_with(profile, src=PARTIAL_SRC, entry="top", args=(8,)):
    with lurk_amd.Context(0) as ctx:
        if profile is not None:
            profile.install(ctx)
        top = lair.Toplevel(src)
        q = lair.QueryRecord(top)
        top.execute_by_name(entry, list(args), q)
        pv = q.expect_public_values()
        m = prover.Machine(ctx, top, entry, len(pv))
        root = m.setup()
        proofs = m.prove(q, num_queries=6, pow_bits=4)
        assert m.verify(proofs, profile=ProtocolProfile.of(ctx))  # the product's verifier under the profile the proofs were made with
        if profile is not None and profile.to_dict() != ProtocolProfile.preset("default").to_dict():
            with pytest.raises(prover.VerificationError):  # ... and not under another one
                m.verify(proofs)
        m.close()
    return root, proofs, pv


def oracle_accepts(oprof, root, proofs, pv, src=PARTIAL_SRC, entry="top"):
    oprof.install()
    try:
        return os_.verify_machine(oracle_airs(src, entry, len(pv)), root, [16], [6], proofs, ob.merkle_verify, profile=oprof)
    finally:
        os_.Profile().install()


@pytest.fixture(scope="module")
def baseline():
    return prove_with(None)


def test_default_profile_roundtrip(ctx, baseline):
    root, proofs, pv = baseline
    assert ProtocolProfile.of(ctx).to_dict() == ProtocolProfile.preset("default").to_dict()
    assert oracle_accepts(os_.Profile(), root, proofs, pv)


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_flipping_a_field_moves_gpu_and_oracle_together(variant, baseline):
    lib, oprof = profiles_for(variant)
    root, proofs, pv = prove_with(lib)
    base_root, base_proofs, _ = baseline
    assert not np.array_equal(proofs[0].words, base_proofs[0].words), "the field has no effect on the GPU prover"
    assert oracle_accepts(oprof, root, proofs, pv)
    with pytest.raises(os_.VerifyError):  # the default oracle does not accept the flipped prover ...
        oracle_accepts(os_.Profile(), root, proofs, pv)
    with pytest.raises(os_.VerifyError):  # ... nor the flipped oracle the default prover
        oracle_accepts(oprof, base_root, base_proofs, pv)


def test_host_transcript_matches_oracle_under_every_challenger_setting():
    for squeeze in (8, 16):
        for front in (0, 1):
            lib = ProtocolProfile.from_dict({"challenger_squeeze": squeeze, "challenger_pop_front": front})
            oprof = os_.Profile.from_dict(lib.to_dict())
            with lurk_amd.Context(0) as ctx:
                lib.install(ctx)
                ch = prover.Challenger(ctx)
                och = os_.Challenger(os_.default_permute16(), oprof)
                for obs, n in (([3, 1, 4, 1, 5], 3), (list(range(20)), 18), ([9], 1)):
                    ch.observe(obs)
                    och.observe(obs)
                    assert ch.sample(n) == [och.sample() for _ in range(n)], (squeeze, front)


def test_default_transcript_tripwire_on_the_library(ctx):
    """tests/golden/default_transcript.json through the library's host challenger under its default profile."""
    import json
    import os

    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "default_transcript.json")))
    ch, got = prover.Challenger(ctx), []
    for op, arg in doc["ops"]:
        if op == "observe":
            ch.observe(list(arg))
        elif op == "sample":
            got += ch.sample(arg)
        else:
            got.append(ch.sample_bits(arg))
    assert got == doc["outputs"]


def test_unsupported_profiles_are_refused(ctx):
    import ctypes as C

    from lurk_amd import _native as N

    p = ProtocolProfile.preset("default")
    p.fri_log_arity = 2
    assert N.lib.lurkhip_set_protocol_profile(ctx.handle, C.byref(p)) == N.ERR_UNSUPPORTED
    p = ProtocolProfile.preset("default")
    p.struct_bytes -= 4
    assert N.lib.lurkhip_set_protocol_profile(ctx.handle, C.byref(p)) == N.ERR_INVALID_ARG
    p = ProtocolProfile.preset("default")
    p.challenger_squeeze = 12
    assert N.lib.lurkhip_set_protocol_profile(ctx.handle, C.byref(p)) == N.ERR_INVALID_ARG
    assert ProtocolProfile.of(ctx).to_dict() == ProtocolProfile.preset("default").to_dict()  # nothing stuck


# ---------------------------------------------------------------- upstream vectors, GPU side
DOCS = uh.load_all() + uh.load_all(uh.SELFCHECK_DIR)


@pytest.mark.skipif(not DOCS, reason="no vectors in tests/golden/upstream/ or tests/golden/selfcheck/")
@pytest.mark.parametrize("name,doc", DOCS or [("none", {})], ids=[n for n, _ in DOCS] or ["none"])
def test_upstream_vectors_on_the_gpu(name, doc):
    """Every vector file -- real upstream dumps (tests/golden/upstream/, none yet: S1 parity unpinned) and the repository's own
    schema self-check file (tests/golden/selfcheck/, made by the HIP prover: exercises every key of the schema, pins nothing
    upstream) -- through the C ABI on the GPU."""
    lib = ProtocolProfile.from_dict(uh.profile_dict(doc), base=uh.preset_of(doc))
    with lurk_amd.Context(0) as ctx:
        lib.install(ctx)
        for v in doc.get("poseidon2_16", []):  # the profile's permutation on the device (lurkhip_perm16)
            x = np.array([v["input"]], dtype=np.uint32)
            out = np.zeros_like(x)
            ctx.check(lurk_amd._native.lib.lurkhip_perm16(ctx.handle, 1, x.ctypes.data, out.ctypes.data, 0))
            assert out[0].tolist() == [int(t) for t in v["output"]]
        for v in doc.get("challenger", []):
            ch, got = prover.Challenger(ctx), []
            for op, arg in v["ops"]:
                if op == "observe":
                    ch.observe(list(arg))
                elif op == "sample":
                    got += ch.sample(arg)
                else:
                    got.append(ch.sample_bits(arg))
            assert got == [int(x) for x in v["outputs"]]
        for v in doc.get("coset_lde", []):
            m = np.array(v["values"], dtype=np.uint32).reshape(1 << v["log_n"], v["width"])
            assert lcommit.coset_lde(ctx, m, v["log_blowup"]).reshape(-1).tolist() == [int(x) for x in v["lde_bit_reversed"]]
        for key, blow in (("pcs_commit", None), ("mmcs_commit", 0)):
            for v in doc.get(key, []):
                mats = [np.array(m["values"], dtype=np.uint32).reshape(1 << m["log_height"], m["width"]) for m in v["matrices"]]
                c = lcommit.commit(ctx, mats, log_blowup=v["log_blowup"]) if blow is None else lcommit.mmcs_commit(ctx, mats)
                assert [int(x) for x in c.root] == [int(x) for x in v["root"]]
                c.close()
        _gpu_permutation_trace_and_shard_proof(ctx, doc)


def _gpu_permutation_trace_and_shard_proof(ctx, doc):
    """GPU side of the `permutation_trace` and `shard_proof` keys: the device permutation trace of the chip equals the vector,
    and the HIP prover's ShardProof bincode equals the vector's bytes."""
    import torch

    from lurk_amd import air, field, proofs
    from lurk_amd import _native as N

    for v in doc.get("permutation_trace", []):
        top = lair.Toplevel(v["program"], lurk_chips=v.get("lurk_chips", False))
        q = lair.QueryRecord(top)
        top.execute_by_name(v["entry"], list(v["args"]), q)
        i = top.func_index(v["chip"])
        a = air.ChipAir.for_func(top, i)
        t_host = lair.FuncChip(ctx, i, top).generate_trace(lair.Shard.new(q), repr=N.REPR_MONTY)
        h = t_host.shape[0]
        t = torch.from_numpy(np.ascontiguousarray(t_host).view(np.int32)).cuda()
        out = torch.zeros((h, 4 * a.permutation_width), dtype=torch.int32, device="cuda")
        cs = a.permutation_trace(ctx, h, t, None, [int(x) for x in v["challenges"]], out)
        got = field.from_monty(out.cpu().numpy().view(np.uint32)).reshape(-1)
        assert got.tolist() == [int(x) for x in v["trace"]]
        assert [int(x) for x in cs] == [int(x) for x in v["cumulative_sum"]]
    for v in doc.get("shard_proof", []):
        top = lair.Toplevel(v["program"], lurk_chips=v.get("lurk_chips", False))
        q = lair.QueryRecord(top)
        top.execute_by_name(v["entry"], list(v["args"]), q)
        pv = q.expect_public_values()
        m = prover.Machine(ctx, top, v["entry"], len(pv))
        root = m.setup()
        if "vk_root" in v:
            assert [int(x) for x in root] == [int(x) for x in v["vk_root"]]
        (p,) = m.prove(q, num_queries=v["num_queries"], pow_bits=v["pow_bits"])
        names = [ch_air.name for _, _, ch_air in m.chips]
        prof = ProtocolProfile.of(ctx).to_dict()
        got = proofs.shard_proof_bincode(p.words, names, serialize_montgomery=bool(prof["serialize_montgomery"]))
        assert got.hex() == v["bincode_hex"]
        m.close()
