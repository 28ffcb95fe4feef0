"""Loader of tests/golden/upstream/*.json (schema: tests/golden/upstream/README.md)."""
import glob
import json
import os

DIR = os.path.join(os.path.dirname(__file__), "golden", "upstream")


def vector_files():
    return sorted(glob.glob(os.path.join(DIR, "*.json")))


def load_all():
    out = []
    for f in vector_files():
        with open(f) as fh:
            out.append((os.path.basename(f), json.load(fh)))
    return out


def profile_dict(doc):
    """The vector file's profile overrides as the plain dict both Profile classes take (rc_16_30 split by the rule of the
    README)."""
    d = dict(doc.get("profile", {}))
    d.pop("preset", None)
    rc = d.pop("rc_16_30", None)
    if rc is not None:
        assert len(rc) == 30 and all(len(r) == 16 for r in rc)
        d["p16_ext_rc"] = [list(r) for r in rc[0:4]] + [list(r) for r in rc[17:21]]
        d["p16_int_rc"] = [r[0] for r in rc[4:17]]
        d["p16_rounds_p"] = 13
    return d


def preset_of(doc):
    return doc.get("profile", {}).get("preset", "default")
