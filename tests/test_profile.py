"""The protocol profile on the CPU: the C structure and its ctypes / oracle mirrors agree, presets are what they say, the
oracle's restated width-16 permutation matches the table-driven one, and the transcript reacts to every challenger field."""
import ctypes as C

import pytest

from lurk_amd import _native as N
from lurk_amd.profile import ProtocolProfile
from oracle import binding as ob
from oracle import stark as os_

P = 2013265921


def test_struct_layout_and_presets():
    p = ProtocolProfile.preset("default")
    assert p.struct_bytes == C.sizeof(ProtocolProfile)  # the C side wrote its own sizeof
    d = p.to_dict()
    # the oracle mirrors every scalar field with the same default
    for k, v in os_.Profile.FIELDS.items():
        if v is not None:
            assert d[k] == v, k
    assert set(d) == set(os_.Profile.FIELDS)
    assert d["challenger_squeeze"] == 8 and d["p16_internal_scale"] == 1 and d["fri_log_arity"] == 1
    h = ProtocolProfile.preset("hardened").to_dict()
    assert (h["observe_openings"], h["observe_chip_meta"], h["challenger_squeeze"]) == (1, 1, 8)
    assert ProtocolProfile.preset("whole-state-squeeze").to_dict()["challenger_squeeze"] == 16
    m = ProtocolProfile.preset("p3-monty-diffusion").to_dict()
    assert m["p16_diag"] == [P - 2] + [1 << s for s in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15)]
    assert m["p16_internal_scale"] * pow(2, 32, P) % P == 1
    bad = ProtocolProfile()
    assert N.lib.lurkhip_protocol_profile_preset(b"no-such-preset", C.byref(bad)) == N.ERR_INVALID_ARG
    # round trip through the dict form
    assert ProtocolProfile.from_dict(m).to_dict() == m


def test_oracle_custom_perm16_equals_table_driven_one(oracle):
    """oracle/commit.c restates the width-16 permutation on its own for profile tables: with the default tables and scale 1 it
    must be the permutation pinned (through widths 24/32/40 of the same code) by the reference's KATs."""
    d = ProtocolProfile.preset("default").to_dict()
    states = [[(7 * i + 3 * k * k + 1) % P for i in range(16)] for k in range(5)]
    want = [[int(x) for x in ob.p2_permute(16, s)[0]] for s in states]
    try:
        ob.set_p16(d["p16_rounds_p"], d["p16_ext_rc"], d["p16_int_rc"] + [0] * (32 - len(d["p16_int_rc"])), d["p16_diag"], 1)
        assert [ob.perm16(s) for s in states] == want
        # a scale s multiplies every internal layer: different permutation
        ob.set_p16(d["p16_rounds_p"], d["p16_ext_rc"], d["p16_int_rc"] + [0] * 19, d["p16_diag"], 5)
        assert ob.perm16(states[0]) != want[0]
    finally:
        ob.set_p16()
    assert ob.perm16(states[1]) == want[1]


@pytest.mark.parametrize("field,value", [("challenger_squeeze", 16), ("challenger_pop_front", 1)])
def test_oracle_transcript_reacts_to_challenger_fields(oracle, field, value):
    def run(profile):
        ch = os_.Challenger(os_.default_permute16(), profile)
        ch.observe(list(range(1, 12)))
        out = [ch.sample() for _ in range(20)]
        ch.observe(5)
        return out + [ch.sample_bits(10)]

    base, other = run(os_.Profile()), run(os_.Profile(**{field: value}))
    assert base != other
    if field == "challenger_squeeze":
        # 16 outputs per permutation instead of the default 8: the first 8 samples of the rate-8 transcript are lanes 7..0, of
        # the whole-state one lanes 15..8
        assert base[:8] != other[:8]


def test_upstream_loader_mechanics(oracle, tmp_path, monkeypatch):
    """The vector loader itself (not parity): a file written from the ORACLE's own outputs under a non-default profile must
    pass every check, and fail once a value in it is altered.  Real pins come from files dumped from sphinx (README.md in
    tests/golden/upstream/)."""
    import json

    import numpy as np

    import test_upstream_vectors as tv
    import upstream_helpers as uh

    doc = {"source": "oracle self-check", "profile": {"preset": "p3-monty-diffusion", "challenger_squeeze": 8}}
    prof = tv.oracle_profile_of(doc).install()
    try:
        st = [(i * i + 5) % P for i in range(16)]
        doc["poseidon2_16"] = [{"input": st, "output": ob.perm16(st)}]
        ch = os_.Challenger(os_.default_permute16(), prof)
        ch.observe([1, 2, 3])
        outs = [ch.sample() for _ in range(9)] + [ch.sample_bits(7)]
        doc["challenger"] = [{"ops": [["observe", [1, 2, 3]], ["sample", 9], ["sample_bits", 7]], "outputs": outs}]
        m = (np.arange(8 * 3, dtype=np.uint32).reshape(8, 3) * 77 + 1) % P
        lde = ob.lde(m, 1)
        root, _ = ob.merkle_commit([lde])
        doc["coset_lde"] = [{"log_n": 3, "width": 3, "values": m.reshape(-1).tolist(), "log_blowup": 1, "lde_bit_reversed": lde.reshape(-1).tolist()}]
        doc["pcs_commit"] = [{"matrices": [{"log_height": 3, "width": 3, "values": m.reshape(-1).tolist()}], "log_blowup": 1, "root": root.tolist()}]
        doc["mmcs_commit"] = [{"matrices": [{"log_height": 4, "width": 3, "values": lde.reshape(-1).tolist()}], "root": root.tolist()}]
        # the permutation_trace key, from the oracle's own generator on the reference's demo fib
        from lair_helpers import load_cases
        from oracle import air as oa
        from oracle import lair as ol

        demo = load_cases()[0]["source"]
        otop = ol.Toplevel(demo)
        oq = ol.QueryRecord(otop)
        ol.execute(otop, "fib", [7], oq)
        rows, _ = ol.generate_trace(otop, "fib", oq)
        alpha, beta = (1, 2, 3, 4), (9, 8, 7, 6)
        pt = os_.permutation_trace(oa.FuncAir(otop, "fib"), rows, None, alpha, beta, 2, public=oq.public_values)
        doc["permutation_trace"] = [{"program": demo, "entry": "fib", "args": [7], "chip": "fib", "challenges": list(alpha + beta),
                                     "trace": [x for r in pt for c in r for x in c], "cumulative_sum": list(pt[-1][-1])}]
    finally:
        os_.Profile().install()
    (tmp_path / "selfcheck.json").write_text(json.dumps(doc))
    monkeypatch.setattr(uh, "DIR", str(tmp_path))
    docs = uh.load_all()
    assert [n for n, _ in docs] == ["selfcheck.json"]
    for _, d in docs:
        pr = tv.oracle_profile_of(d).install()
        try:
            for chk in tv.CHECKS:
                chk((d, pr))
            d["mmcs_commit"][0]["root"][0] ^= 1
            with pytest.raises(AssertionError):
                tv.test_mmcs_commit((d, pr))
            d["permutation_trace"][0]["trace"][5] ^= 1
            with pytest.raises(AssertionError):
                tv.test_permutation_trace((d, pr))
        finally:
            os_.Profile().install()


def test_default_transcript_tripwire(oracle):
    """The default profile cannot move silently: a fixed transcript under it is committed (tests/golden/default_transcript.json,
    written by make_default_transcript.py from the oracle).  The library's default preset must carry the same scalars."""
    import importlib.util
    import json
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    doc = json.load(open(os.path.join(here, "default_transcript.json")))
    spec = importlib.util.spec_from_file_location("make_default_transcript", os.path.join(here, "make_default_transcript.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.OPS == doc["ops"]
    assert gen.run() == doc["outputs"]
    lib = ProtocolProfile.preset("default").to_dict()
    for k, v in doc["profile_scalars"].items():
        assert lib[k] == v, k
    assert gen.run(os_.Profile(challenger_squeeze=16)) != doc["outputs"]


def test_sphinx_preset_takes_its_round_constants_from_a_vector_file(tmp_path):
    """tools/upstream_dump writes RC_16_30 into the vector file's profile; `ProtocolProfile.sphinx` builds the profile from it
    (external rounds 0..3 and 17..20, internal rounds 4..16 lane 0) on top of the p3-monty-diffusion preset -- and refuses to
    invent the constants when no file exists (none ships: they are not in the reference tree)."""
    import json

    from lurk_amd.profile import ProtocolProfile

    rc = [[(1000 * r + c) % 2013265921 for c in range(16)] for r in range(30)]
    f = tmp_path / "sphinx.json"
    f.write_text(json.dumps({"source": "synthetic", "profile": {"rc_16_30": rc, "preset": "p3-monty-diffusion"}}))
    p = ProtocolProfile.sphinx(str(f)).to_dict()
    base = ProtocolProfile.preset("p3-monty-diffusion").to_dict()
    assert p["p16_ext_rc"] == [x for r in rc[0:4] + rc[17:21] for x in r]
    assert p["p16_int_rc"] == [r[0] for r in rc[4:17]] and p["p16_rounds_p"] == 13
    assert p["p16_diag"] == base["p16_diag"] and p["p16_internal_scale"] == base["p16_internal_scale"]
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({"profile": {}}))
    with pytest.raises(ValueError):
        ProtocolProfile.sphinx(str(bad))


def test_committed_pmc_pass_belongs_to_this_tree():
    """bench.py prices the hashing launches' live time with the instruction count of the committed PMC pass (profiles/pmc_traffic.json)
    and falls back to that pass's own rate when the hashing kernels' sources have changed since (round 5's driver line went out on the
    fallback).  The committed pass must therefore be one of THIS tree: the SHA-256 over the files the hashing kernels are made of is
    recomputed here -- a round that edits them re-runs tools/final_profiles.sh + tools/summarise_profiles.py before it is done."""
    import hashlib
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_traffic.json")) as f:
        pmc = json.load(f)
    assert "lurk_amd/csrc/commit.h" not in pmc["hash_kernel_sources"]  # (host-side declarations: not what the kernels are made of)
    h = hashlib.sha256()
    for rel in pmc["hash_kernel_sources"]:
        with open(os.path.join(root, rel), "rb") as src:
            h.update(src.read())
    assert h.hexdigest() == pmc["hash_kernel_sources_sha256"], "the hashing kernels changed after the committed PMC pass: re-run the profiles"
