"""Writes a shard proof object in the product's flat word layout (test helper; lurk_amd/prover.py: parse_proof is the inverse)."""
import numpy as np

from lurk_amd.prover import PROOF_MAGIC


def encode_words(sh):
    """A shard proof object (oracle/wire.py: Shard, or lurk_amd.prover.ShardProof) in the product's flat layout
    (lurk_amd/prover.py: parse_proof is the inverse)."""
    w = [PROOF_MAGIC, len(sh.chips), sh.log_blowup, sh.num_queries, sh.pow_bits, len(sh.public_values), len(sh.fri_roots), sh.log_max_height,
         sh.n_preprocessed, sum(c.quotient_degree for c in sh.chips)]
    for c in sh.chips:
        w += [c.machine_index, c.log_n, c.width, c.prep_width, c.perm_width, c.quotient_degree, c.prep_index + 1] + list(c.cumulative_sum)
    w += list(sh.public_values) + list(sh.main_root) + list(sh.perm_root) + list(sh.quot_root)
    flat = lambda vals: [x for v in vals for x in v]
    if sh.n_preprocessed:
        by = {c.prep_index: c for c in sh.chips if c.prep_index >= 0}
        for m in range(sh.n_preprocessed):
            w += flat(by[m].opened["prep"][0]) + flat(by[m].opened["prep"][1])
    for key in ("main", "perm"):
        for c in sh.chips:
            w += flat(c.opened[key][0]) + flat(c.opened[key][1])
    for c in sh.chips:
        for chunk in c.opened["quotient"]:
            w += flat(chunk)
    for r in sh.fri_roots:
        w += list(r)
    w += list(sh.final_poly) + [sh.pow_witness] + list(sh.query_indices)
    for rw, recs in list(sh.round_openings) + list(sh.layer_openings):
        w.append(rw)
        for r in recs:
            w += list(r)
    return np.array([int(x) % (1 << 32) for x in w], dtype=np.uint32)
