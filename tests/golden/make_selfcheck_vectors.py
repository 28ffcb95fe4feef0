#!/usr/bin/env python3
"""Writes tests/golden/selfcheck/selfcheck.json: ONE file in the schema of tests/golden/upstream/README.md that carries EVERY
key, produced by this repository itself -- the oracle for the stage vectors, the HIP prover (needs a GPU) for `shard_proof`.
It pins nothing upstream (S1 parity stays unpinned); it exists so that every loader / checker of the schema runs in the CPU
suite (oracle side, tests/test_upstream_vectors.py via tests/test_profile.py) and in the GPU suite
(tests/test_profile_gpu.py) instead of being skipped until somebody has cargo.  Run on a GPU box:
    python tests/golden/make_selfcheck_vectors.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import lurk_amd  # noqa: E402
from lurk_amd import lair, proofs, prover  # noqa: E402
from lurk_amd.profile import ProtocolProfile  # noqa: E402
from oracle import air as oa  # noqa: E402
from oracle import binding as ob  # noqa: E402
from oracle import lair as ol  # noqa: E402
from oracle import stark as os_  # noqa: E402

P = 2013265921
PROFILE = {"challenger_pop_front": 1, "observe_openings": 1}  # a non-default profile: the loaders must honour the overrides

PROGRAM = """
fn fib(n): [1] {
    let one = 1;
    match n {
        0 => {
            return one
        }
        1 => {
            return one
        }
    };
    let a = sub(n, one);
    let b = sub(a, one);
    let x = call(fib, a);
    let y = call(fib, b);
    let r = add(x, y);
    return r
}
"""


def main():
    ob.build()
    doc = {"source": "lurkhip self-check (oracle + HIP prover of this repository) -- NOT an upstream vector", "profile": dict(PROFILE)}
    prof = os_.Profile(**PROFILE).install()
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
        # permutation trace of the fib chip of fib(7) (the reference's own golden trace program shape, src/lair/trace.rs:483-514)
        otop = ol.Toplevel(PROGRAM)
        oq = ol.QueryRecord(otop)
        ol.execute(otop, "fib", [7], oq)
        rows, _ = ol.generate_trace(otop, "fib", oq)
        alpha, beta = (11, 22, 33, 44), (5, 6, 7, P - 1)
        pt = os_.permutation_trace(oa.FuncAir(otop, "fib"), rows, None, alpha, beta, 2, public=oq.public_values)
        doc["permutation_trace"] = [{"program": PROGRAM, "entry": "fib", "args": [7], "chip": "fib", "challenges": list(alpha + beta), "batch_size": 2,
                                     "trace": [x for r in pt for c in r for x in c], "cumulative_sum": list(pt[-1][-1])}]
    finally:
        os_.Profile().install()
    # the whole ShardProof from the HIP prover under the same profile
    with lurk_amd.Context(0) as ctx:
        ProtocolProfile.from_dict(PROFILE).install(ctx)
        top = lair.Toplevel(PROGRAM)
        q = lair.QueryRecord(top)
        top.execute_by_name("fib", [7], q)
        pv = q.expect_public_values()
        mach = prover.Machine(ctx, top, "fib", len(pv))
        vk = mach.setup()
        (p,) = mach.prove(q, num_queries=2, pow_bits=3)
        names = [a.name for _, _, a in mach.chips]
        data = proofs.shard_proof_bincode(p.words, names)
        doc["shard_proof"] = [{"program": PROGRAM, "entry": "fib", "args": [7], "num_queries": 2, "pow_bits": 3, "vk_root": [int(x) for x in vk],
                               "bincode_hex": data.hex()}]
        mach.close()
    out = os.path.join(HERE, "selfcheck", "selfcheck.json")
    with open(out, "w") as f:
        json.dump(doc, f)
        f.write("\n")
    print("wrote", out, os.path.getsize(out), "bytes; ShardProof", len(data), "bytes")


if __name__ == "__main__":
    main()
