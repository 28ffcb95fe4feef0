"""Loader of tests/golden/upstream/*.json (schema: tests/golden/upstream/README.md)."""
import glob
import json
import os

DIR = os.path.join(os.path.dirname(__file__), "golden", "upstream")
# the repository's own file in the same schema (made by tests/golden/make_selfcheck_vectors.py on a GPU box from the HIP prover
# and the oracle): exercises every key of the loader; NOT an upstream pin
SELFCHECK_DIR = os.path.join(os.path.dirname(__file__), "golden", "selfcheck")


def vector_files(directory=None):
    return sorted(glob.glob(os.path.join(directory or DIR, "*.json")))


def load_all(directory=None):
    out = []
    for f in vector_files(directory):
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


# ---- helpers of the `permutation_trace` and `shard_proof` keys (oracle side; no GPU)
def oracle_machine(program, entry, args, lurk_chips=False):
    """The oracle's toplevel + query record of `entry(args)`, its chip names in machine order (lair_chip.rs:85-93,196-211) and
    its AIRs by machine index."""
    from oracle import air as oa
    from oracle import lair as ol

    otop = ol.Toplevel(program, chips=ol.lurk_chips() if lurk_chips else ())
    oq = ol.QueryRecord(otop)
    ol.execute(otop, entry, list(args), oq)
    return otop, oq


def oracle_airs_and_names(otop, entry, n_public):
    from oracle import air as oa
    from oracle import lair as ol

    idx = otop.index[entry]
    airs = [oa.EntrypointAir(idx, n_public)] + [oa.FuncAir(otop, f["name"]) for f in otop.funcs]
    airs += [oa.MemAir(ml) for ml in ol.MEM_TABLE_SIZES] + [oa.BytesAir()]
    names = [f"Entrypoint[{idx}]"] + [f"Func[{f['name']}]" for f in otop.funcs] + [f"Mem[{ml}-wide]" for ml in ol.MEM_TABLE_SIZES] + ["CPU"]
    return airs, names


def oracle_vk_root():
    """The machine's preprocessed commitment (one preprocessed chip: the byte table, 2^16 x 6,
    /root/reference/src/gadgets/bytes/trace.rs:49-72) through the oracle's LDE + Merkle tree."""
    import numpy as np

    from oracle import binding as ob

    i = np.arange(1 << 16)
    i1, i2 = i & 0xFF, i >> 8
    t = np.stack([i1, i2, (i1 < i2).astype(int), i1 & i2, i1 ^ i2, i1 | i2], axis=1).astype(np.uint32)
    root, _ = ob.merkle_commit([ob.lde(t, 1)])
    return [int(x) for x in root]
