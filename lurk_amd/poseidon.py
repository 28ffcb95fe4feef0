"""Host-side mirror of the reference's Poseidon chipset over the HIP kernels.

`PoseidonChipset` keeps the names and argument meaning of
/root/reference/src/core/poseidon.rs:14-94 (`hash`, `execute_simple`,
`populate_witness`, the four size getters); `Hasher` mirrors
/root/reference/src/core/zstore.rs:222-249 (`hash3/4/5`, `hash` dispatching on the
preimage length).  The batch methods are what a GPU-aware caller would use: one
launch for n preimages.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .context import Context, _addr, as_u32

OUTPUT_SIZE = 8  # core/poseidon.rs:14


class PoseidonChipset:
    def __init__(self, ctx: Context, width: int):
        cols = N.lib.lurkhip_poseidon2_num_cols(width)
        if cols < 0:
            raise ValueError(f"unsupported Poseidon2 width {width}")
        self.ctx = ctx
        self.width = width
        self._num_cols = cols

    # --- Chipset sizes (src/lair/chipset.rs:9-17, core/poseidon.rs:44-59)
    def input_size(self) -> int:
        return self.width

    def output_size(self) -> int:
        return OUTPUT_SIZE

    def witness_size(self) -> int:
        return OUTPUT_SIZE + self._num_cols

    def require_size(self) -> int:
        return 0

    def num_cols(self) -> int:
        return self._num_cols

    # --- batched entry points (host buffers)
    def _check(self, x: np.ndarray) -> np.ndarray:
        x = as_u32(x)
        if x.ndim != 2 or x.shape[1] != self.width:
            raise ValueError(f"expected [n, {self.width}] preimages, got {x.shape}")
        return x

    def permute_batch(self, x, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        x = self._check(x)
        out = np.empty_like(x)
        self.ctx.check(N.lib.lurkhip_poseidon2_permute(self.ctx.handle, self.width, x.shape[0], _addr(x), _addr(out), repr))
        return out

    def hash_batch(self, x, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        x = self._check(x)
        out = np.empty((x.shape[0], OUTPUT_SIZE), dtype=np.uint32)
        self.ctx.check(N.lib.lurkhip_poseidon2_hash8(self.ctx.handle, self.width, x.shape[0], _addr(x), _addr(out), repr))
        return out

    def witness_batch(self, x, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        x = self._check(x)
        out = np.empty((x.shape[0], self.witness_size()), dtype=np.uint32)
        self.ctx.check(
            N.lib.lurkhip_poseidon2_wide_witness(self.ctx.handle, self.width, x.shape[0], _addr(x), _addr(out), repr)
        )
        return out

    # --- device-buffer entry points (torch tensors / raw pointers), async on ctx stream
    def permute_dev(self, x, out, n: int, repr: int = N.REPR_CANONICAL):
        self.ctx.check(N.lib.lurkhip_poseidon2_permute_dev(self.ctx.handle, self.width, n, _addr(x), _addr(out), repr))

    def hash_dev(self, x, out, n: int, repr: int = N.REPR_CANONICAL):
        self.ctx.check(N.lib.lurkhip_poseidon2_hash8_dev(self.ctx.handle, self.width, n, _addr(x), _addr(out), repr))

    def witness_dev(self, x, out, n: int, repr: int = N.REPR_CANONICAL):
        self.ctx.check(
            N.lib.lurkhip_poseidon2_wide_witness_dev(self.ctx.handle, self.width, n, _addr(x), _addr(out), repr)
        )

    # --- single-input interface with the reference's names
    def hash(self, preimg) -> list[int]:
        return [int(v) for v in self.hash_batch(np.asarray(preimg, dtype=np.uint32)[None, :])[0]]

    def execute_simple(self, input) -> list[int]:
        return self.hash(input)

    def populate_witness(self, input, witness: np.ndarray) -> list[int]:
        """Fills `witness` (length witness_size()) and returns the *full* permuted state, exactly as
        the reference does (core/poseidon.rs:65-72 returns `result.to_vec()`, all W lanes)."""
        x = np.asarray(input, dtype=np.uint32)[None, :]
        w = self.witness_batch(x)[0]
        witness[: self.witness_size()] = w
        return [int(v) for v in self.permute_batch(x)[0]]


class Poseidon2Chip:
    """The narrow chip of /root/reference/src/poseidon/mod.rs:17-27: one trace row per round.  `generate_trace` keeps the
    reference's meaning (/root/reference/src/poseidon/trace.rs:14-46): a list of W-lane states -> a row-major matrix of
    next_power_of_two(len * (R_F + R_P + 1)) rows, zero rows after the last permutation."""

    def __init__(self, ctx: Context, width: int):
        if N.lib.lurkhip_poseidon2_num_cols(width) < 0:
            raise ValueError(f"unsupported Poseidon2 width {width}")
        self.ctx = ctx
        self.input_width = width

    def shape(self, n: int) -> tuple[int, int]:
        """(height, width) of the trace of n permutations."""
        import ctypes as C

        w, h = C.c_uint32(), C.c_uint64()
        N.check(N.lib.lurkhip_poseidon2_trace_shape(self.input_width, n, C.byref(w), C.byref(h)))
        return int(h.value), int(w.value)

    def width(self) -> int:  # BaseAir::width, poseidon/air.rs:15-19
        return self.shape(0)[1]

    def generate_trace(self, inputs, repr: int = N.REPR_CANONICAL) -> np.ndarray:
        x = as_u32(np.asarray(inputs, dtype=np.uint32).reshape(-1, self.input_width))
        out = np.empty(self.shape(x.shape[0]), dtype=np.uint32)
        self.ctx.check(N.lib.lurkhip_poseidon2_trace(self.ctx.handle, self.input_width, x.shape[0], _addr(x) if x.size else None, _addr(out), repr))
        return out

    def generate_trace_dev(self, x, out, n: int, repr: int = N.REPR_CANONICAL):
        """x [n][W] -> out [shape(n)], device buffers, asynchronous on the context's stream."""
        self.ctx.check(N.lib.lurkhip_poseidon2_trace_dev(self.ctx.handle, self.input_width, n, _addr(x), _addr(out), repr))

    def air(self):
        from .air import ChipAir

        return ChipAir.for_poseidon2(self.input_width)


class Hasher:
    """hash3 / hash4 / hash5 over widths 24 / 32 / 40 (core/chipset.rs:176-182)."""

    def __init__(self, ctx: Context):
        self.chips = {24: PoseidonChipset(ctx, 24), 32: PoseidonChipset(ctx, 32), 40: PoseidonChipset(ctx, 40)}

    def hash(self, preimg) -> list[int]:
        chip = self.chips.get(len(preimg))
        if chip is None:
            raise ValueError("preimage length must be 24, 32 or 40")  # zstore.rs:241-248 `unreachable!()`
        return chip.hash(preimg)

    def hash_many(self, preimgs) -> list[list[int]]:
        """Digests of many preimages at once: one kernel launch per width present (24 / 32 / 40)."""
        out = [None] * len(preimgs)
        for width, chip in self.chips.items():
            idx = [i for i, p in enumerate(preimgs) if len(p) == width]
            if idx:
                digests = chip.hash_batch(np.array([preimgs[i] for i in idx], dtype=np.uint32))
                for i, d in zip(idx, digests):
                    out[i] = [int(v) for v in d]
        if any(o is None for o in out):
            raise ValueError("preimage length must be 24, 32 or 40")
        return out

    def hash3(self, preimg):
        assert len(preimg) == 24
        return self.hash(preimg)

    def hash4(self, preimg):
        assert len(preimg) == 32
        return self.hash(preimg)

    def hash5(self, preimg):
        assert len(preimg) == 40
        return self.hash(preimg)
