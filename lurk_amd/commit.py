"""Host-side mirror of the commit stage (p3 `Pcs::commit` as sphinx's prover calls it): coset LDE of each
trace matrix + one mixed-height Poseidon2-16 Merkle tree, all on the device."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

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
        pitch = C.c_uint32()
        self.ctx.check(N.lib.lurkhip_commitment_matrix_pitch(self.ctx.handle, self.handle, index, C.byref(pitch)))
        h, pitch = 1 << lh, pitch.value
        if pitch == w:
            out = np.empty((h, w), dtype=np.uint32)
            self.ctx.d2h(out, ptr)
            return from_monty(out)
        flat = np.zeros(h * pitch, dtype=np.uint32)  # a column range of a wider buffer: rows are `pitch` words apart
        self.ctx.d2h(flat[: (h - 1) * pitch + w], ptr)
        return from_monty(np.ascontiguousarray(flat.reshape(h, pitch)[:, :w]))


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


def mmcs_commit(ctx: Context, mats, repr: int = N.REPR_CANONICAL) -> Commitment:
    """The Merkle commitment of host matrices as given (no LDE): p3 FieldMerkleTreeMmcs::commit."""
    mats = [as_u32(m) for m in mats]
    lh = [m.shape[0].bit_length() - 1 for m in mats]
    if any(m.ndim != 2 or m.shape[0] != 1 << k for m, k in zip(mats, lh)):
        raise ValueError("every matrix must be [2^k, w]")
    n = len(mats)
    ptrs = (C.c_void_p * n)(*[_addr(m) for m in mats])
    lha, ws = np.asarray(lh, dtype=np.uint32), np.asarray([m.shape[1] for m in mats], dtype=np.uint32)
    handle = C.c_void_p()
    root = np.empty(8, dtype=np.uint32)
    ctx.check(N.lib.lurkhip_mmcs_commit(ctx.handle, n, C.cast(ptrs, C.c_void_p), _addr(lha), _addr(ws), repr, C.byref(handle), _addr(root)))
    return Commitment(ctx, handle, root, lh, [int(x) for x in ws], 0)


def commit_dev(ctx: Context, mats, log_heights, widths, log_blowup: int = 1, repr: int = N.REPR_CANONICAL, keep_coeffs: bool = False) -> Commitment:
    """mats: list of device buffers (torch tensors or raw pointers)."""
    return _commit(ctx, N.lib.lurkhip_commit_dev, mats, log_heights, widths, log_blowup, repr, keep_coeffs)


def commit_dev_sparse(ctx: Context, mats, log_heights, widths, log_blowup: int = 1, repr: int = N.REPR_CANONICAL, aligned_groups: bool = False):
    """commit_dev that leaves identically-zero columns out of the LDE (lurkhip_commit_dev_sparse): (Commitment, columns left out)."""
    n = len(mats)
    ptrs = (C.c_void_p * n)(*[_addr(m) for m in mats])
    lh = np.asarray(log_heights, dtype=np.uint32)
    ws = np.asarray(widths, dtype=np.uint32)
    handle = C.c_void_p()
    root = np.empty(8, dtype=np.uint32)
    nz = np.zeros(1, dtype=np.uint32)
    ctx.check(N.lib.lurkhip_commit_dev_sparse(ctx.handle, n, C.cast(ptrs, C.c_void_p), _addr(lh), _addr(ws), log_blowup, repr, 1 if aligned_groups else 0,
                                              C.byref(handle), _addr(root), _addr(nz)))
    return Commitment(ctx, handle, root, [int(x) for x in lh], [int(x) for x in ws], log_blowup), int(nz[0])


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


# ------------------------------------------------------------------ Pcs::open on its own (lurkhip_open)
OPENING_MAGIC = 0x4E504F4C  # "LOPN"


@dataclass
class Opening:
    """Parsed result of `open_rounds`.  Word layout (all canonical):
    [magic "LOPN", n_rounds, log_blowup, num_queries, pow_bits, n_layers, log_max_height]
    per round: n_mats, then per matrix (log_n, width, n_points)
    opened values [round][matrix][point][column] (4 words each)
    n_layers FRI layer roots (8 words each), final polynomial (4), proof-of-work witness (1), num_queries query indices
    per round: record size R, then num_queries records of R words = the LDE rows of every matrix at the query's index
      (concatenated, committed order) followed by the 8-word siblings of its Merkle path, leaf level first
    per FRI layer: record size, then num_queries records = the pair of extension elements (8 words) + the path."""
    log_blowup: int
    num_queries: int
    pow_bits: int
    log_max_height: int
    shapes: list          # per round: [(log_n, width, n_points)]
    opened: list          # [round][matrix][point] -> list of EF tuples
    fri_roots: list
    final_poly: tuple
    pow_witness: int
    query_indices: list
    round_openings: list  # per round: (record_words, [records])
    layer_openings: list
    words: np.ndarray = None


def parse_opening(words) -> Opening:
    w = [int(x) for x in words]
    pos = [0]

    def take(n):
        out = w[pos[0]:pos[0] + n]
        if len(out) != n:
            raise ValueError("truncated opening")
        pos[0] += n
        return out

    magic, n_rounds, log_blowup, nq, pow_bits, n_layers, log_max = take(7)
    if magic != OPENING_MAGIC:
        raise ValueError("not a lurkhip opening (bad magic)")
    shapes = []
    for _ in range(n_rounds):
        (n_mats,) = take(1)
        shapes.append([tuple(take(3)) for _ in range(n_mats)])
    opened = []
    for mats in shapes:
        rnd = []
        for _, width, n_pts in mats:
            pts = []
            for _ in range(n_pts):
                flat = take(4 * width)
                pts.append([tuple(flat[4 * i:4 * i + 4]) for i in range(width)])
            rnd.append(pts)
        opened.append(rnd)
    fri_roots = [take(8) for _ in range(n_layers)]
    final_poly = tuple(take(4))
    (pow_witness,) = take(1)
    indices = take(nq)
    rounds = []
    for _ in range(n_rounds):
        (rw,) = take(1)
        rounds.append((rw, [take(rw) for _ in range(nq)]))
    layers = []
    for _ in range(n_layers):
        (rw,) = take(1)
        layers.append((rw, [take(rw) for _ in range(nq)]))
    if pos[0] != len(w):
        raise ValueError("trailing words in opening")
    return Opening(log_blowup, nq, pow_bits, log_max, shapes, opened, fri_roots, final_poly, pow_witness, indices, rounds, layers,
                   np.asarray(words, dtype=np.uint32))


def open_rounds(ctx: Context, commitments, points, challenger, num_queries: int = 100, pow_bits: int = 16, parse: bool = True):
    """p3 `Pcs::open`: `commitments` = Commitment handles (one per round), `points[r][m]` = the one or two extension-field
    points (4 canonical lanes each) matrix m of round r is opened at; `challenger` = lurk_amd.prover.Challenger in the
    verifier's state.  Returns the opened values + FRI proof (Opening, or the raw words with parse=False)."""
    n_points, flat = [], []
    for c, rp in zip(commitments, points):
        if len(rp) != len(c.widths):
            raise ValueError("one point list per committed matrix")
        for mp in rp:
            n_points.append(len(mp))
            for z in mp:
                flat.extend(int(x) for x in z)
    handles = (C.c_void_p * len(commitments))(*[c.handle for c in commitments])
    npt = np.asarray(n_points, dtype=np.uint32)
    pts = np.asarray(flat, dtype=np.uint32)
    h = C.c_void_p()
    ctx.check(N.lib.lurkhip_open(ctx.handle, len(commitments), C.cast(handles, C.c_void_p), _addr(npt), _addr(pts), challenger.handle,
                                 num_queries, pow_bits, C.byref(h)))
    try:
        n = N.lib.lurkhip_proof_words(h)
        words = np.empty(n, dtype=np.uint32)
        ctx.check(N.lib.lurkhip_proof_read(h, _addr(words), n))
    finally:
        N.lib.lurkhip_proof_free(h)
    return parse_opening(words) if parse else words
