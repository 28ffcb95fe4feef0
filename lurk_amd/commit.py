"""Host-side mirror of the commit stage (p3 `Pcs::commit` as sphinx's prover calls it): coset LDE of each
trace matrix + one mixed-height Poseidon2-16 Merkle tree, all on the device."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context, _addr, as_u32


def coset_lde(ctx: Context, mat: np.ndarray, log_blowup: int = 1, repr: int = N.REPR_CANONICAL) -> np.ndarray:
    mat = as_u32(mat)
    n, w = mat.shape
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("height must be a power of two")
    out = np.empty((n << log_blowup, w), dtype=np.uint32)
    ctx.check(N.lib.lurkhip_coset_lde(ctx.handle, log_n, w, log_blowup, _addr(mat), _addr(out), repr))
    return out


def coset_lde_dev(ctx: Context, log_n: int, width: int, log_blowup: int, src, dst, repr: int = N.REPR_CANONICAL):
    ctx.check(N.lib.lurkhip_coset_lde_dev(ctx.handle, log_n, width, log_blowup, _addr(src), _addr(dst), repr))


class Commitment:
    """Handle to the LDE matrices and Merkle tree kept on the device."""

    def __init__(self, ctx: Context, handle, root, log_heights, widths, log_blowup):
        self.ctx = ctx
        self.handle = handle
        self.root = root
        self.log_heights = [h + log_blowup for h in log_heights]  # of the LDE matrices
        self.widths = list(widths)
        self.log_max = max(self.log_heights)

    def close(self):
        if self.handle:
            N.lib.lurkhip_commitment_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def open(self, index: int, repr: int = N.REPR_CANONICAL):
        rows = np.empty(sum(self.widths), dtype=np.uint32)
        path = np.empty((self.log_max, 8), dtype=np.uint32)
        self.ctx.check(N.lib.lurkhip_commitment_open(self.ctx.handle, self.handle, index, _addr(rows), _addr(path), repr))
        return rows, path

    def matrix_dev(self, index: int):
        p = C.c_void_p()
        lh = C.c_uint32()
        w = C.c_uint32()
        self.ctx.check(N.lib.lurkhip_commitment_matrix_dev(self.ctx.handle, self.handle, index, C.byref(p), C.byref(lh), C.byref(w)))
        return p.value, lh.value, w.value

    def lde_host(self, index: int) -> np.ndarray:
        """Copies the LDE of matrix `index` back (canonical form)."""
        from .field import from_monty

        ptr, lh, w = self.matrix_dev(index)
        out = np.empty((1 << lh, w), dtype=np.uint32)
        self.ctx.d2h(out, ptr)
        return from_monty(out)


def _commit(ctx: Context, fn, mats, log_heights, widths, log_blowup, repr, keep_coeffs) -> Commitment:
    n = len(mats)
    ptrs = (C.c_void_p * n)(*[_addr(m) for m in mats])
    lh = np.asarray(log_heights, dtype=np.uint32)
    ws = np.asarray(widths, dtype=np.uint32)
    handle = C.c_void_p()
    root = np.empty(8, dtype=np.uint32)
    ctx.check(fn(ctx.handle, n, C.cast(ptrs, C.c_void_p), _addr(lh), _addr(ws), log_blowup, repr, int(keep_coeffs), C.byref(handle), _addr(root)))
    return Commitment(ctx, handle, root, [int(x) for x in lh], [int(x) for x in ws], log_blowup)


def commit(ctx: Context, mats, log_blowup: int = 1, repr: int = N.REPR_CANONICAL, keep_coeffs: bool = False) -> Commitment:
    """mats: list of host [2^k, w] uint32 arrays."""
    mats = [as_u32(m) for m in mats]
    lh = []
    for m in mats:
        k = m.shape[0].bit_length() - 1
        if m.ndim != 2 or m.shape[0] != 1 << k:
            raise ValueError("every matrix must be [2^k, w]")
        lh.append(k)
    return _commit(ctx, N.lib.lurkhip_commit, mats, lh, [m.shape[1] for m in mats], log_blowup, repr, keep_coeffs)


def commit_dev(ctx: Context, mats, log_heights, widths, log_blowup: int = 1, repr: int = N.REPR_CANONICAL, keep_coeffs: bool = False) -> Commitment:
    """mats: list of device buffers (torch tensors or raw pointers)."""
    return _commit(ctx, N.lib.lurkhip_commit_dev, mats, log_heights, widths, log_blowup, repr, keep_coeffs)


def commit_cosets_dev(ctx: Context, mats, log_heights, widths, shifts, log_blowup: int = 1, repr: int = N.REPR_MONTY) -> Commitment:
    """As commit_dev for matrices given over cosets: shifts[i] = 31 / (coset shift of matrix i) (canonical)."""
    n = len(mats)
    ptrs = (C.c_void_p * n)(*[_addr(m) for m in mats])
    lh = np.asarray(log_heights, dtype=np.uint32)
    ws = np.asarray(widths, dtype=np.uint32)
    sh = np.asarray(shifts, dtype=np.uint32)
    handle = C.c_void_p()
    root = np.empty(8, dtype=np.uint32)
    ctx.check(N.lib.lurkhip_commit_cosets_dev(ctx.handle, n, C.cast(ptrs, C.c_void_p), _addr(lh), _addr(ws), _addr(sh), log_blowup, repr, C.byref(handle), _addr(root)))
    return Commitment(ctx, handle, root, [int(x) for x in lh], [int(x) for x in ws], log_blowup)
