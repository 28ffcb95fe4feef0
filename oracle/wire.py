"""ORACLE (test infrastructure, not product code): an independent DECODER of the reference's serialised proofs.

Reads `bincode::serialize(&CryptoProof)` / `CachedProof` bytes (/root/reference/src/core/cli/proofs.rs:22-35,137-143;
bincode 1.x defaults: little-endian, lengths and usize as u64, enum variants as u32) back into the structure the oracle's
verifier consumes (oracle/stark.py verify_machine), so that the product's encoder (lurk_amd/csrc/wire.cpp) is checked by
something that shares no code with it: flat words -> bincode (product) -> decode + verify (oracle).

The inner sphinx / Plonky3 types are [UPSTREAM-RECALL]; their field order is restated here on its own:
  CryptoShardProof { commitment {main, permutation, quotient: [F; 8]},
                     opened_values { chips: Vec<{preprocessed, main, permutation: {local, next: Vec<[F; 4]>},
                                                 quotient: Vec<Vec<[F; 4]>>, cumulative_sum: [F; 4], log_degree: usize}> },
                     opening_proof { fri_proof { commit_phase_commits: Vec<[F; 8]>,
                                                 query_proofs: Vec<{commit_phase_openings: Vec<{sibling_value: [F; 4], opening_proof: Vec<[F; 8]>}>}>,
                                                 final_poly: [F; 4], pow_witness: F },
                                     query_openings: Vec<Vec<{opened_values: Vec<Vec<F>>, opening_proof: Vec<[F; 8]>}>> },
                     chip_ordering: HashMap<String, usize> }
Only tests/ may import anything under oracle/."""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

P = 2013265921
R_INV = pow(pow(2, 32, P), P - 2, P)


class Reader:
    def __init__(self, data: bytes, montgomery: bool = False):
        self.d, self.pos, self.monty = data, 0, montgomery

    def u32(self):
        (v,) = struct.unpack_from("<I", self.d, self.pos)
        self.pos += 4
        return v

    def u64(self):
        (v,) = struct.unpack_from("<Q", self.d, self.pos)
        self.pos += 8
        return v

    def f(self):
        v = self.u32()
        if v >= P:
            raise ValueError("field element out of range")
        return v * R_INV % P if self.monty else v

    def fs(self, n):
        return [self.f() for _ in range(n)]

    def ef(self):
        return tuple(self.fs(4))

    def vec(self, item):
        n = self.u64()
        if n > len(self.d):
            raise ValueError("implausible length")
        return [item() for _ in range(n)]

    def string(self):
        n = self.u64()
        s = self.d[self.pos:self.pos + n].decode()
        self.pos += n
        return s


@dataclass
class Chip:
    machine_index: int
    log_n: int
    width: int
    prep_width: int
    perm_width: int
    quotient_degree: int
    prep_index: int
    cumulative_sum: tuple
    opened: dict = field(default_factory=dict)
    name: str = ""


@dataclass
class Shard:
    log_blowup: int
    num_queries: int
    pow_bits: int
    log_max_height: int
    chips: list
    public_values: list
    main_root: list
    perm_root: list
    quot_root: list
    fri_roots: list
    final_poly: tuple
    pow_witness: int
    round_openings: list
    layer_openings: list
    n_preprocessed: int
    query_indices: object = None   # not in the upstream format: the verifier derives them from its transcript
    sibling_only: bool = True      # FRI steps carry the sibling of the queried element, not the pair


def _air_values(r: Reader):
    return r.vec(r.ef), r.vec(r.ef)


def decode_shard(r: Reader, chip_names, public_values, log_blowup, pow_bits) -> Shard:
    main_root, perm_root, quot_root = r.fs(8), r.fs(8), r.fs(8)
    chips = []
    for _ in range(r.u64()):
        prep, main, perm = _air_values(r), _air_values(r), _air_values(r)
        quotient = r.vec(lambda: r.vec(r.ef))
        cs = r.ef()
        log_n = r.u64()
        c = Chip(-1, log_n, len(main[0]), len(prep[0]), len(perm[0]), len(quotient), -1, cs)
        c.opened = {"main": main, "perm": perm, "quotient": quotient}
        if prep[0]:
            c.opened["prep"] = prep
        if len(main[0]) != len(main[1]) or len(perm[0]) != len(perm[1]) or len(prep[0]) != len(prep[1]) or any(len(q) != 4 for q in quotient):
            raise ValueError("opened values shape")
        chips.append(c)
    fri_roots = r.vec(lambda: r.fs(8))
    n_layers = len(fri_roots)
    log_max = n_layers + log_blowup
    query_proofs = r.vec(lambda: r.vec(lambda: (r.ef(), r.vec(lambda: r.fs(8)))))
    final_poly = r.ef()
    pow_witness = r.f()
    query_openings = r.vec(lambda: r.vec(lambda: (r.vec(lambda: r.vec(r.f)), r.vec(lambda: r.fs(8)))))
    ordering = {}
    for _ in range(r.u64()):
        name = r.string()
        ordering[name] = r.u64()
    if sorted(ordering.values()) != list(range(len(chips))):
        raise ValueError("chip_ordering is not a permutation of the chips")
    for name, i in ordering.items():
        chips[i].name = name
        chips[i].machine_index = chip_names.index(name)
    k = 0
    for c in chips:  # preprocessed matrices: in chip order (one preprocessed chip in the Lurk machine, the byte table)
        if c.prep_width:
            c.prep_index = k
            k += 1
    nq = len(query_proofs)
    if len(query_openings) != nq:
        raise ValueError("query counts differ")
    n_rounds = len(query_openings[0]) if nq else 0
    rounds = []
    for ri in range(n_rounds):
        recs = []
        for q in range(nq):
            rows, path = query_openings[q][ri]
            recs.append([x for row in rows for x in row] + [x for d in path for x in d])
        if len({len(x) for x in recs}) > 1:
            raise ValueError("ragged round records")
        rounds.append((len(recs[0]) if recs else 0, recs))
    layers = []
    for li in range(n_layers):
        recs = []
        for q in range(nq):
            if len(query_proofs[q]) != n_layers:
                raise ValueError("FRI step count")
            sib, path = query_proofs[q][li]
            recs.append(list(sib) + [x for d in path for x in d])
        layers.append((len(recs[0]) if recs else 0, recs))
    return Shard(log_blowup, nq, pow_bits, log_max, chips, list(public_values), main_root, perm_root, quot_root, fri_roots, final_poly,
                 pow_witness, rounds, layers, k)


def decode_shard_proof(data: bytes, chip_names, log_blowup=1, pow_bits=16, montgomery=False) -> Shard:
    """One sphinx `ShardProof` (bincode): the CryptoShardProof fields followed by `public_values: Vec<Val>`
    (/root/reference/src/core/cli/proofs.rs:61-74,94-101 show the two structs side by side)."""
    r = Reader(data, montgomery)
    s = decode_shard(r, chip_names, [], log_blowup, pow_bits)
    s.public_values = r.vec(r.f)
    if r.pos != len(data):
        raise ValueError("trailing bytes after the ShardProof")
    return s


def decode_crypto_proof(data: bytes, chip_names, public_values_of_depth, log_blowup=1, pow_bits=16, montgomery=False, reader=None):
    """-> (shards, verifier_version, depth).  `public_values_of_depth(depth)` rebuilds the public values the way
    CryptoProof::into_machine_proof does (proofs.rs:44-79); they are not in the bytes."""
    r = reader or Reader(data, montgomery)
    n = r.u64()
    start = r.pos
    # two passes: the depth comes after the shard proofs, the public values depend on it
    shards = [decode_shard(r, chip_names, [], log_blowup, pow_bits) for _ in range(n)]
    version = r.string()
    depth = r.u32()
    pv = public_values_of_depth(depth)
    for s in shards:
        s.public_values = list(pv)
    if reader is None and r.pos != len(data):
        raise ValueError("trailing bytes")
    del start
    return shards, version, depth


def decode_zptr(r: Reader):
    return (r.u32(), tuple(r.fs(8)))


def decode_cached_proof(data: bytes, chip_names, log_blowup=1, pow_bits=16, montgomery=False):
    """-> (shards, version, depth, expr, env, result, zdag) with zdag = [(zptr, kind, children)] in file order; the public
    values are rebuilt from expr / env / result / depth (proofs.rs:46-56)."""
    r = Reader(data, montgomery)
    holder = {}
    shards, version, depth = decode_crypto_proof(data, chip_names, lambda d: holder.setdefault("d", d) and [] or [], log_blowup, pow_bits, montgomery, reader=r)
    expr, env, result = decode_zptr(r), decode_zptr(r), decode_zptr(r)
    zdag = []
    for _ in range(r.u64()):
        z = decode_zptr(r)
        kind = r.u32()
        if kind > 2:
            raise ValueError("unknown ZPtrType variant")
        zdag.append((z, kind, [decode_zptr(r) for _ in range({0: 0, 1: 2, 2: 3}[kind])]))
    if r.pos != len(data):
        raise ValueError("trailing bytes")

    def flat(z):
        return [z[0]] + [0] * 7 + list(z[1])

    pv = flat(expr) + list(env[1]) + flat(result) + [(depth >> (8 * i)) & 0xFF for i in range(4)]
    for s in shards:
        s.public_values = list(pv)
    return shards, version, depth, expr, env, result, zdag
