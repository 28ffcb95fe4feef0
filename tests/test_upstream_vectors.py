"""Pins for S1 from the real crates: loads every tests/golden/upstream/*.json (README.md there) and checks the ORACLE against
it.  While that directory holds no vectors parity of S1 is "unpinned"; the repository's own file in the same schema
(tests/golden/selfcheck/, made by make_selfcheck_vectors.py from the oracle and the HIP prover) keeps every checker below
running -- it pins nothing upstream."""
import numpy as np
import pytest

import upstream_helpers as uh
from oracle import binding as ob
from oracle import stark as os_

UPSTREAM_DOCS = uh.load_all()
DOCS = UPSTREAM_DOCS + uh.load_all(uh.SELFCHECK_DIR)
pytestmark = pytest.mark.skipif(not DOCS, reason="no vectors in tests/golden/upstream/ or tests/golden/selfcheck/")


def test_upstream_pins_present_or_declared_absent():
    """Bookkeeping, not parity: with no real upstream file the S1 rows stay "parity unpinned" (DESIGN.md 5)."""
    for name, d in UPSTREAM_DOCS:
        assert "self-check" not in d.get("source", ""), f"{name}: self-made vectors belong in tests/golden/selfcheck/"


def oracle_profile_of(d):
    prof = os_.Profile.from_dict(uh.profile_dict(d))
    if uh.preset_of(d) == "p3-monty-diffusion" and "p16_diag" not in d.get("profile", {}):
        P = os_.P
        prof.p16_diag = [P - 2] + [1 << s for s in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15)]
        prof.p16_internal_scale = pow(pow(2, 32, P), P - 2, P)
    if prof.p16_ext_rc is None and (prof.p16_diag is not None or prof.p16_internal_scale != 1):
        from lurk_amd.profile import ProtocolProfile

        base = ProtocolProfile.preset("default").to_dict()
        prof.p16_ext_rc, prof.p16_int_rc = base["p16_ext_rc"], base["p16_int_rc"]
        prof.p16_diag = prof.p16_diag or base["p16_diag"]
    return prof


@pytest.fixture(params=DOCS or [("none", {})], ids=lambda d: d[0])
def doc(request, oracle):
    name, d = request.param
    prof = oracle_profile_of(d).install()
    yield d, prof
    os_.Profile().install()


CHECKS = []


def check(f):
    CHECKS.append(f)
    return f


@check
def test_poseidon2_16(doc):
    d, _ = doc
    for v in d.get("poseidon2_16", []):
        assert ob.perm16(v["input"]) == [int(x) for x in v["output"]]


@check
def test_challenger(doc):
    d, prof = doc
    for v in d.get("challenger", []):
        ch = os_.Challenger(os_.default_permute16(), prof)
        got = []
        for op, arg in v["ops"]:
            if op == "observe":
                ch.observe(list(arg))
            elif op == "sample":
                got += [ch.sample() for _ in range(arg)]
            else:
                got.append(ch.sample_bits(arg))
        assert got == [int(x) for x in v["outputs"]]


def _mats(v):
    return [np.array(m["values"], dtype=np.uint32).reshape(1 << m["log_height"], m["width"]) for m in v["matrices"]]


@check
def test_mmcs_commit(doc):
    d, _ = doc
    for v in d.get("mmcs_commit", []):
        root, _ = ob.merkle_commit(_mats(v))
        assert [int(x) for x in root] == [int(x) for x in v["root"]]


@check
def test_coset_lde(doc):
    d, _ = doc
    for v in d.get("coset_lde", []):
        m = np.array(v["values"], dtype=np.uint32).reshape(1 << v["log_n"], v["width"])
        got = ob.lde(m, v["log_blowup"])
        assert got.reshape(-1).tolist() == [int(x) for x in v["lde_bit_reversed"]]


@check
def test_pcs_commit(doc):
    d, _ = doc
    for v in d.get("pcs_commit", []):
        root, _ = ob.merkle_commit([ob.lde(m, v["log_blowup"]) for m in _mats(v)])
        assert [int(x) for x in root] == [int(x) for x in v["root"]]


@check
def test_permutation_trace(doc):
    """sphinx generate_permutation_trace of one chip under given challenges: the extension-field matrix (flattened to base,
    row-major) and its cumulative sum."""
    from oracle import air as oa
    from oracle import lair as ol

    d, _ = doc
    for v in d.get("permutation_trace", []):
        otop, oq = uh.oracle_machine(v["program"], v["entry"], v["args"], v.get("lurk_chips", False))
        rows, width = ol.generate_trace(otop, v["chip"], oq)
        air = oa.FuncAir(otop, v["chip"])
        alpha, beta = tuple(int(x) for x in v["challenges"][:4]), tuple(int(x) for x in v["challenges"][4:8])
        want = os_.permutation_trace(air, rows, None, alpha, beta, v.get("batch_size", 2), public=oq.public_values)  # batch = 1 << log_quotient_degree
        assert [x for r in want for c in r for x in c] == [int(x) for x in v["trace"]]
        assert list(want[-1][-1]) == [int(x) for x in v["cumulative_sum"]]


@check
def test_shard_proof(doc):
    """`bincode::serialize(&ShardProof)`: decoded by the oracle's reader (oracle/wire.py) and ACCEPTED by the oracle's verifier
    under the file's profile -- the verifier side of "same transcript, same proof bytes"."""
    from oracle import wire as ow

    d, prof = doc
    for v in d.get("shard_proof", []):
        otop, oq = uh.oracle_machine(v["program"], v["entry"], v["args"], v.get("lurk_chips", False))
        airs, names = uh.oracle_airs_and_names(otop, v["entry"], len(oq.public_values))
        shard = ow.decode_shard_proof(bytes.fromhex(v["bincode_hex"]), names, log_blowup=prof.fri_log_blowup, pow_bits=v["pow_bits"],
                                      montgomery=bool(prof.serialize_montgomery))
        assert shard.num_queries == v["num_queries"]
        assert shard.public_values == [int(x) for x in oq.public_values]
        vk_root = [int(x) for x in v["vk_root"]] if "vk_root" in v else uh.oracle_vk_root()
        assert os_.verify_machine(airs, vk_root, [16], [6], [shard], ob.merkle_verify, profile=prof)
