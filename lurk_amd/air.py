"""Host-side mirror of a chip's AIR (constraints + lookup interactions) over the lurkhip C ABI.

`ChipAir.for_func / for_mem / for_bytes / for_entrypoint` correspond to the `Air::eval` impls of
/root/reference/src/lair/{air,memory,lair_chip}.rs and /root/reference/src/gadgets/bytes/trace.rs as the
prover's symbolic builder sees them; `check_trace` is the twin of `machine.debug_constraints`
(/root/reference/src/air/debug.rs:161-206), `permutation_trace` of sphinx's generate_permutation_trace.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context, _addr, as_u32
from .lair import LairError


def _new(status, handle):
    if status != N.OK:
        raise LairError(status, N.last_error(None) or N.lib.lurkhip_lair_last_error().decode("utf-8", "replace"))
    return handle


class ChipAir:
    def __init__(self, handle):
        self.handle = handle
        info = np.zeros(16, dtype=np.uint32)
        N.check(N.lib.lurkhip_air_info(handle, _addr(info)))
        (self.width, self.preprocessed_width, self.num_constraints, self.num_sends, self.num_receives, self.max_constraint_degree,
         self.log_quotient_degree, self.permutation_width, self.interaction_words, self.num_public_values, self.constraint_regs,
         self.constraint_instrs, self.interaction_regs, self.interaction_instrs) = [int(x) for x in info[:14]]
        self.name = N.lib.lurkhip_air_name(handle).decode()

    @classmethod
    def for_func(cls, toplevel, func_idx: int) -> "ChipAir":
        h = C.c_void_p()
        return cls(_new(N.lib.lurkhip_air_func(toplevel.handle, func_idx, C.byref(h)), h))

    @classmethod
    def for_mem(cls, mem_len: int) -> "ChipAir":
        h = C.c_void_p()
        return cls(_new(N.lib.lurkhip_air_mem(mem_len, C.byref(h)), h))

    @classmethod
    def for_bytes(cls) -> "ChipAir":
        h = C.c_void_p()
        return cls(_new(N.lib.lurkhip_air_bytes(C.byref(h)), h))

    @classmethod
    def for_entrypoint(cls, func_idx: int, num_public_values: int) -> "ChipAir":
        h = C.c_void_p()
        return cls(_new(N.lib.lurkhip_air_entrypoint(func_idx, num_public_values, C.byref(h)), h))

    @classmethod
    def for_poseidon2(cls, width: int) -> "ChipAir":
        """The narrow (one row per round) Poseidon2 chip, /root/reference/src/poseidon/air.rs:21-165."""
        h = C.c_void_p()
        return cls(_new(N.lib.lurkhip_air_poseidon2(width, C.byref(h)), h))

    def __del__(self):
        if getattr(self, "handle", None) and N is not None and getattr(N, "lib", None) is not None:
            N.lib.lurkhip_air_free(self.handle)
            self.handle = None

    def compile(self, ctx: Context) -> None:
        """Compile this chip's program pieces to straight-line device code (hiprtc); its permutation traces and quotients on
        `ctx`'s device then run the compiled kernels.  Seconds to tens of seconds per chip: for big traces."""
        ctx.check(N.lib.lurkhip_air_compile(ctx.handle, self.handle))

    def interaction_sizes(self) -> list[int]:
        n = self.num_sends + self.num_receives
        out = np.zeros(max(n, 1), dtype=np.uint32)
        got = N.lib.lurkhip_air_interaction_sizes(self.handle, _addr(out), n)
        assert got == n
        return [int(x) for x in out[:n]]

    def eval_rows(self, ctx: Context, local, nxt, prep_local=None, prep_next=None, public=None, selectors=None):
        """(constraints [n][K], interactions [n][T]) on explicit row pairs; canonical values."""
        local, nxt = as_u32(local), as_u32(nxt)
        n = local.shape[0]
        sel = as_u32(selectors if selectors is not None else np.zeros((n, 3)))
        pl = as_u32(prep_local) if prep_local is not None else None
        pn = as_u32(prep_next) if prep_next is not None else None
        pub = as_u32(public) if public is not None and len(public) else None
        cons = np.zeros((n, max(self.num_constraints, 1)), dtype=np.uint32)
        inter = np.zeros((n, max(self.interaction_words, 1)), dtype=np.uint32)
        cons_arg = np.zeros((n, self.num_constraints), dtype=np.uint32) if self.num_constraints else cons
        inter_arg = np.zeros((n, self.interaction_words), dtype=np.uint32) if self.interaction_words else inter
        ctx.check(N.lib.lurkhip_air_eval_rows(ctx.handle, self.handle, n, _addr(local), _addr(nxt), _addr(pl) if pl is not None else None,
                                              _addr(pn) if pn is not None else None, _addr(pub) if pub is not None else None, _addr(sel),
                                              _addr(cons_arg), _addr(inter_arg)))
        # chips without constraints / interactions: the dummy one-word buffers only keep the pointers valid
        return (cons_arg if self.num_constraints else cons[:, :0]), (inter_arg if self.interaction_words else inter[:, :0])

    def check_trace(self, ctx: Context, height: int, main_dev, prep_dev=None, public=None):
        """(-1, -1) when every constraint vanishes on every row, else (row, constraint index) of the first failure.
        main_dev / prep_dev: device buffers, Montgomery, natural row order."""
        row, k = C.c_int64(), C.c_int32()
        pub = as_u32(public) if public is not None and len(public) else None
        ctx.check(N.lib.lurkhip_air_check_trace_dev(ctx.handle, self.handle, height, _addr(main_dev), _addr(prep_dev) if prep_dev is not None else None,
                                                    _addr(pub) if pub is not None else None, C.byref(row), C.byref(k)))
        return row.value, k.value

    def permutation_trace(self, ctx: Context, height: int, main_dev, prep_dev, challenges, out_dev, want_sum: bool = True):
        """Fills out_dev (height x 4*permutation_width, Montgomery); returns the chip's cumulative sum (4 canonical lanes)."""
        ch = as_u32(challenges).reshape(8)
        cs = np.zeros(4, dtype=np.uint32)
        ctx.check(N.lib.lurkhip_permutation_trace_dev(ctx.handle, self.handle, height, _addr(main_dev), _addr(prep_dev) if prep_dev is not None else None,
                                                      _addr(ch), _addr(out_dev), _addr(cs) if want_sum else None))
        return cs

    def quotient(self, ctx: Context, log_n: int, main_lde_dev, prep_lde_dev, perm_lde_dev, perm_challenges, alpha, cumulative_sum, out_dev, public=None):
        """Fills out_dev with 2^log_quotient_degree chunk matrices (2^log_n x 4 Montgomery words each)."""
        ch = as_u32(perm_challenges).reshape(8)
        al = as_u32(alpha).reshape(4)
        cs = as_u32(cumulative_sum).reshape(4)
        pub = as_u32(public) if public is not None and len(public) else None
        ctx.check(N.lib.lurkhip_quotient_dev(ctx.handle, self.handle, log_n, _addr(main_lde_dev), _addr(prep_lde_dev) if prep_lde_dev is not None else None,
                                             _addr(perm_lde_dev), _addr(ch), _addr(al), _addr(cs), _addr(pub) if pub is not None else None, _addr(out_dev)))
