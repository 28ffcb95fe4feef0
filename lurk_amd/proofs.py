"""Host-side mirror of /root/reference/src/core/cli/proofs.rs: the serialised proof types a Lurk user stores and exchanges.

`CryptoProof.from_machine_proof` = `impl From<MachineProof<BabyBearPoseidon2>> for CryptoProof` (proofs.rs:87-131);
`CryptoProof.to_bytes` = `bincode::serialize` (repl.rs:200-203); `public_values` = the 44-lane layout the verifier rebuilds
(proofs.rs:46-56, stark_machine.rs:16-17); `CachedProof` = proofs.rs:137-169 with the ZDag exported from the native ZStore.
The bytes are produced by the C ABI (lurkhip_crypto_proof_bincode / lurkhip_cached_proof_bincode, lurk_amd/csrc/wire.cpp)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .zstore import ZPtr

DEPTH_W = 4                  # lair/provenance.rs:11
VERIFIER_VERSION = "lurkhip"  # the reference stores env!("VERGEN_GIT_SHA") (proofs.rs:37-40)


def public_values(expr: ZPtr, env: ZPtr, result: ZPtr, depth: int) -> list[int]:
    """[expr flat 16 | env digest 8 | result flat 16 | depth as 4 little-endian bytes] (proofs.rs:52-56)."""
    return expr.flatten() + list(env.digest) + result.flatten() + [(depth >> (8 * i)) & 0xFF for i in range(DEPTH_W)]


class CryptoProof:
    def __init__(self, shard_words, chip_names, verifier_version: str = VERIFIER_VERSION, serialize_montgomery: bool = False):
        self.shard_words = [np.ascontiguousarray(w, dtype=np.uint32) for w in shard_words]
        self.chip_names = list(chip_names)
        self.verifier_version = verifier_version
        self.serialize_montgomery = serialize_montgomery

    @classmethod
    def from_machine_proof(cls, machine, proofs, **kw) -> "CryptoProof":
        """machine: lurk_amd.prover.Machine (its chips' names, by machine index); proofs: what Machine.prove returned."""
        return cls([p.words for p in proofs], [air.name for _, _, air in machine.chips], **kw)

    @property
    def depth(self) -> int:
        w = self.shard_words[0]
        n_chips, n_public = int(w[1]), int(w[5])
        pv = w[10 + 11 * n_chips:10 + 11 * n_chips + n_public]
        return sum(int(b) << (8 * i) for i, b in enumerate(pv[-DEPTH_W:]))

    def to_bytes(self) -> bytes:
        n = len(self.shard_words)
        ptrs = (C.c_void_p * n)(*[w.ctypes.data for w in self.shard_words])
        lens = np.asarray([w.size for w in self.shard_words], dtype=np.uint64)
        names = (C.c_char_p * len(self.chip_names))(*[s.encode() for s in self.chip_names])
        args = (n, C.cast(ptrs, C.c_void_p), lens.ctypes.data_as(C.c_void_p), len(self.chip_names), C.cast(names, C.c_void_p),
                self.verifier_version.encode(), int(self.serialize_montgomery))
        size = N.lib.lurkhip_crypto_proof_bincode(*args, None, 0)
        if size < 0:
            raise ValueError(f"malformed proof words (status {size})")
        buf = (C.c_uint8 * size)()
        assert N.lib.lurkhip_crypto_proof_bincode(*args, C.cast(buf, C.c_void_p), size) == size
        return bytes(buf)


def verify_crypto_proof(machine, data: bytes, public_values, num_queries: int, pow_bits: int, log_blowup: int = 1, profile=None) -> bool:
    """The reference CLI's `verify` on the host (csrc/verify.cpp: lurkhip_crypto_proof_verify): `data` = the bincode of a CryptoProof,
    `public_values` = the 44 lanes rebuilt from the claim (`public_values` above), `machine` = a lurk_amd.prover.Machine of the same
    toplevel (its AIRs, chip names and verifying key).  Raises lurk_amd.prover.VerificationError when the proof is rejected."""
    from .prover import VerificationError

    if machine.pk is None:
        machine.setup()
    airs = [air for _, _, air in machine.chips]
    air_ptrs = (C.c_void_p * len(airs))(*[a.handle for a in airs])
    names = (C.c_char_p * len(airs))(*[a.name.encode() for a in airs])
    vk = np.array(machine.vk_root, dtype=np.uint32)
    lh, ws = np.array([16], dtype=np.uint32), np.array([6], dtype=np.uint32)
    pv = np.array(list(public_values), dtype=np.uint32)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    err = C.create_string_buffer(512)
    u32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    st = N.lib.lurkhip_crypto_proof_verify(C.byref(profile) if profile is not None else None, C.cast(air_ptrs, C.c_void_p), C.cast(names, C.c_void_p),
                                           len(airs), u32p(vk), u32p(lh), u32p(ws), 1, C.cast(buf, C.c_void_p), len(data), u32p(pv), pv.size,
                                           num_queries, pow_bits, log_blowup, err, len(err))
    if st != N.OK:
        raise VerificationError(f"lurkhip status {st}: {err.value.decode('utf-8', 'replace')}")
    return True


def verify_cached_proof(machine, data: bytes, num_queries: int, pow_bits: int, log_blowup: int = 1, profile=None) -> list[int]:
    """`lurk verify` after loading the file (lurkhip_cached_proof_verify): the bincode of a CachedProof carries the claim, the 44
    public values are rebuilt from it and returned.  Raises lurk_amd.prover.VerificationError when the proof is rejected."""
    from .prover import VerificationError

    if machine.pk is None:
        machine.setup()
    airs = [air for _, _, air in machine.chips]
    air_ptrs = (C.c_void_p * len(airs))(*[a.handle for a in airs])
    names = (C.c_char_p * len(airs))(*[a.name.encode() for a in airs])
    vk, lh, ws = np.array(machine.vk_root, dtype=np.uint32), np.array([16], dtype=np.uint32), np.array([6], dtype=np.uint32)
    pv = np.zeros(44, dtype=np.uint32)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    err = C.create_string_buffer(512)
    st = N.lib.lurkhip_cached_proof_verify(C.byref(profile) if profile is not None else None, C.cast(air_ptrs, C.c_void_p), C.cast(names, C.c_void_p),
                                           len(airs), vk.ctypes.data, lh.ctypes.data, ws.ctypes.data, 1, C.cast(buf, C.c_void_p), len(data), num_queries,
                                           pow_bits, log_blowup, pv.ctypes.data, err, len(err))
    if st != N.OK:
        raise VerificationError(f"lurkhip status {st}: {err.value.decode('utf-8', 'replace')}")
    return [int(x) for x in pv]


def shard_proof_bincode(words, chip_names, serialize_montgomery: bool = False) -> bytes:
    """`bincode::serialize(&ShardProof)` of one shard proof (sphinx's struct, public values included) from its flat words."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    names = (C.c_char_p * len(chip_names))(*[s.encode() for s in chip_names])
    args = (w.ctypes.data_as(C.c_void_p), w.size, len(chip_names), C.cast(names, C.c_void_p), int(serialize_montgomery))
    size = N.lib.lurkhip_shard_proof_bincode(*args, None, 0)
    if size < 0:
        raise ValueError(f"malformed proof words (status {size})")
    buf = (C.c_uint8 * size)()
    assert N.lib.lurkhip_shard_proof_bincode(*args, C.cast(buf, C.c_void_p), size) == size
    return bytes(buf)


class CachedProof:
    """crypto_proof + the Lurk data of its public values, fully specified (proofs.rs:137-169)."""

    def __init__(self, crypto_proof: CryptoProof, expr: ZPtr, env: ZPtr, result: ZPtr, zstore):
        self.crypto_proof, self.expr, self.env, self.result = crypto_proof, expr, env, result
        self.zdag = zstore.dag_export([expr, env, result])  # ZDag::populate_with_many

    def to_bytes(self) -> bytes:
        cp = self.crypto_proof.to_bytes()
        ent = np.zeros((max(len(self.zdag), 1), 37), dtype=np.uint32)
        for i, (z, kind, kids) in enumerate(self.zdag):
            ent[i, 0], ent[i, 1:9], ent[i, 9] = z.tag, z.digest, kind
            for k, c in enumerate(kids):
                ent[i, 10 + 9 * k], ent[i, 11 + 9 * k:19 + 9 * k] = c.tag, c.digest
        zp = [np.asarray([z.tag] + list(z.digest), dtype=np.uint32) for z in (self.expr, self.env, self.result)]
        p = C.c_void_p
        args = (cp, len(cp), zp[0].ctypes.data_as(p), zp[1].ctypes.data_as(p), zp[2].ctypes.data_as(p), len(self.zdag), ent.ctypes.data_as(p),
                int(self.crypto_proof.serialize_montgomery))
        size = N.lib.lurkhip_cached_proof_bincode(*args, None, 0)
        if size < 0:
            raise ValueError(f"malformed cached proof (status {size})")
        buf = (C.c_uint8 * size)()
        assert N.lib.lurkhip_cached_proof_bincode(*args, C.cast(buf, p), size) == size
        return bytes(buf)
