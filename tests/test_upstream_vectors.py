"""Pins for S1 from the real crates: loads every tests/golden/upstream/*.json (README.md there) and checks the ORACLE against
it.  Skipped while the directory holds no vectors (parity of S1 is then "unpinned")."""
import numpy as np
import pytest

import upstream_helpers as uh
from oracle import binding as ob
from oracle import stark as os_

DOCS = uh.load_all()
pytestmark = pytest.mark.skipif(not DOCS, reason="no upstream vectors in tests/golden/upstream/ (S1 parity unpinned)")


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
