"""Deterministic synthetic inputs: lanes uniform in [0, p) drawn
from splitmix64 with seed 0x4C55524B ("LURK") + stream offset."""
from __future__ import annotations

import numpy as np

from .field import P

SEED = 0x4C55524B
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(n: int, seed: int = SEED) -> np.ndarray:
    """n consecutive splitmix64 outputs for the given seed."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def field_elements(shape, seed: int = SEED) -> np.ndarray:
    n = int(np.prod(shape))
    return (splitmix64(n, seed) % np.uint64(P)).astype(np.uint32).reshape(shape)
