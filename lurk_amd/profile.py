"""The protocol profile (include/lurkhip.h: lurkhip_protocol_profile): every choice of the commit / transcript / FRI layer
that is restated from memory of the absent third-party sources (sphinx-core @ 8a39b951, Plonky3 @ a0b92870) in ONE
structure per context.  The test oracle keeps a field-by-field mirror (class Profile of its stark module); tests/golden/upstream/ pins it once vectors
dumped from the real crates are available ("Pinning S1")."""
from __future__ import annotations

import ctypes as C

from . import _native as N


class ProtocolProfile(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32),
        ("p16_rounds_p", C.c_uint32),
        ("p16_ext_rc", C.c_uint32 * 128),
        ("p16_int_rc", C.c_uint32 * 32),
        ("p16_diag", C.c_uint32 * 16),
        ("p16_internal_scale", C.c_uint32),
        ("challenger_squeeze", C.c_uint32),
        ("challenger_pop_front", C.c_uint32),
        ("observe_openings", C.c_uint32),
        ("observe_chip_meta", C.c_uint32),
        ("constraint_alpha_ascending", C.c_uint32),
        ("fri_alpha_global", C.c_uint32),
        ("fri_log_arity", C.c_uint32),
        ("fri_log_blowup", C.c_uint32),
        ("fri_num_queries", C.c_uint32),
        ("fri_pow_bits", C.c_uint32),
        ("serialize_montgomery", C.c_uint32),
    ]

    SCALARS = ("p16_rounds_p", "p16_internal_scale", "challenger_squeeze", "challenger_pop_front", "observe_openings", "observe_chip_meta",
               "constraint_alpha_ascending", "fri_alpha_global", "fri_log_arity", "fri_log_blowup", "fri_num_queries", "fri_pow_bits",
               "serialize_montgomery")

    @classmethod
    def preset(cls, name: str = "default") -> "ProtocolProfile":
        p = cls()
        N.check(N.lib.lurkhip_protocol_profile_preset(name.encode(), C.byref(p)))
        return p

    def to_dict(self) -> dict:
        """Plain-Python form: what the checker's mirror class and the upstream vector files use."""
        d = {k: int(getattr(self, k)) for k in self.SCALARS}
        d["p16_ext_rc"] = [int(x) for x in self.p16_ext_rc]
        d["p16_int_rc"] = [int(x) for x in self.p16_int_rc][: d["p16_rounds_p"]]
        d["p16_diag"] = [int(x) for x in self.p16_diag]
        return d

    @classmethod
    def from_dict(cls, d: dict, base: str = "default") -> "ProtocolProfile":
        p = cls.preset(base)
        for k in cls.SCALARS:
            if k in d:
                setattr(p, k, int(d[k]))
        if "p16_ext_rc" in d:
            flat = [int(x) for row in d["p16_ext_rc"] for x in (row if isinstance(row, (list, tuple)) else [row])]
            assert len(flat) == 128, "p16_ext_rc needs 8 x 16 values"
            p.p16_ext_rc[:] = flat
        if "p16_int_rc" in d:
            rc = [int(x) for x in d["p16_int_rc"]]
            assert len(rc) <= 32
            p.p16_int_rc[:] = rc + [0] * (32 - len(rc))
            p.p16_rounds_p = len(rc)
        if "p16_diag" in d:
            p.p16_diag[:] = [int(x) for x in d["p16_diag"]]
        return p

    @classmethod
    def sphinx(cls, vector_file: str | None = None) -> "ProtocolProfile":
        """sphinx's `BabyBearPoseidon2` as far as this repository can state it: the `p3-monty-diffusion` preset (power-of-two internal
        diagonal, layer scaled by 2^-32) with the width-16 round constants RC_16_30 of an upstream vector file
        (tests/golden/upstream/*.json, key "profile.rc_16_30", written by tools/upstream_dump from the real crates -- tools/pin_s1.sh).
        The constants are not in the reference tree, so no such file ships: without one this raises."""
        import glob
        import json
        import os

        if vector_file is None:
            here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            found = sorted(glob.glob(os.path.join(here, "tests", "golden", "upstream", "*.json")))
            if not found:
                raise FileNotFoundError("no upstream vector file: run tools/pin_s1.sh on a machine with cargo (sphinx's RC_16_30 are not in the reference tree)")
            vector_file = found[0]
        with open(vector_file) as fh:
            prof = dict(json.load(fh).get("profile", {}))
        base = prof.pop("preset", "p3-monty-diffusion")
        rc = prof.pop("rc_16_30", None)
        if rc is None:
            raise ValueError(f"{vector_file} carries no profile.rc_16_30")
        if len(rc) != 30 or any(len(r) != 16 for r in rc):
            raise ValueError("RC_16_30 is 30 rounds of 16 constants")
        # rounds 0..3 and 17..20 are the external ones, 4..16 the internal ones (lane 0) [UPSTREAM-RECALL: sphinx inner_perm]
        prof["p16_ext_rc"] = [list(r) for r in rc[0:4]] + [list(r) for r in rc[17:21]]
        prof["p16_int_rc"] = [r[0] for r in rc[4:17]]
        return cls.from_dict(prof, base=base)

    def install(self, ctx) -> None:
        ctx.check(N.lib.lurkhip_set_protocol_profile(ctx.handle, C.byref(self)))

    @classmethod
    def of(cls, ctx) -> "ProtocolProfile":
        p = cls()
        ctx.check(N.lib.lurkhip_get_protocol_profile(ctx.handle, C.byref(p)))
        return p
