"""BabyBear constants and tiny host helpers (no kernels here)."""
from __future__ import annotations

import numpy as np

P = 2013265921  # 15 * 2^27 + 1
MONTY_BITS = 32
GENERATOR = 31
TWO_ADICITY = 27


def to_monty(a: np.ndarray) -> np.ndarray:
    """canonical -> x * 2^32 mod p (the storage form of p3's BabyBear)."""
    a = np.asarray(a, dtype=np.uint64)
    return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


_R_INV = pow(1 << 32, -1, P)


def from_monty(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.uint64)
    return ((a * np.uint64(_R_INV)) % np.uint64(P)).astype(np.uint32)


def digest_to_int(digest) -> int:
    """Little-endian base-p number of a digest, the way the reference prints big nums
    (/root/reference/src/core/big_num.rs:101-108)."""
    n = 0
    for limb in reversed([int(x) for x in digest]):
        n = n * P + limb
    return n
