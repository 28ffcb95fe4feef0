"""ORACLE (test infrastructure): ctypes access to oracle/liblurkoracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblurkoracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.or_p2_num_cols.restype = C.c_int
        for name in ("or_p2_permute", "or_p2_hash8", "or_p2_wide_witness"):
            f = getattr(_lib, name)
            f.restype = C.c_int
            f.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        _lib.or_f_inv.restype = C.c_uint32
        _lib.or_f_inv.argtypes = [C.c_uint32]
        _lib.or_f_mul.restype = C.c_uint32
        _lib.or_f_mul.argtypes = [C.c_uint32, C.c_uint32]
    return _lib


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def p2_num_cols(width: int) -> int:
    return lib().or_p2_num_cols(width)


def p2_permute(width: int, x) -> np.ndarray:
    x = _u32(x).reshape(-1, width)
    out = np.empty_like(x)
    assert lib().or_p2_permute(width, x.shape[0], x.ctypes.data, out.ctypes.data) == 0
    return out


def p2_hash8(width: int, x) -> np.ndarray:
    x = _u32(x).reshape(-1, width)
    out = np.empty((x.shape[0], 8), dtype=np.uint32)
    assert lib().or_p2_hash8(width, x.shape[0], x.ctypes.data, out.ctypes.data) == 0
    return out


def p2_wide_witness(width: int, x) -> np.ndarray:
    x = _u32(x).reshape(-1, width)
    out = np.empty((x.shape[0], 8 + p2_num_cols(width)), dtype=np.uint32)
    assert lib().or_p2_wide_witness(width, x.shape[0], x.ctypes.data, out.ctypes.data) == 0
    return out


def p2_narrow_width(width: int) -> int:
    return lib().or_p2_narrow_width(width)


def p2_narrow_trace(width: int, x) -> np.ndarray:
    """Trace of the narrow Poseidon2 chip (one row per round) for the given states, zero-padded to a power of two."""
    x = _u32(x).reshape(-1, width)
    rp = p2_params(width)[0]
    rows = x.shape[0] * (8 + rp + 1)
    height = 1 << max(rows - 1, 0).bit_length()
    out = np.empty((height, p2_narrow_width(width)), dtype=np.uint32)
    L = lib()
    L.or_p2_narrow_trace.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    assert L.or_p2_narrow_trace(width, x.shape[0], x.ctypes.data, height, out.ctypes.data) == 0
    return out


class _P2Params(C.Structure):
    _fields_ = [("width", C.c_int), ("rounds_p", C.c_int), ("diag", C.POINTER(C.c_uint32)), ("ext_rc", C.POINTER(C.c_uint32)),
                ("int_rc", C.POINTER(C.c_uint32))]


def p2_params(width: int):
    """(rounds_p, diag[W], ext_rc[8][W], int_rc[rounds_p]) of the oracle's Poseidon2 tables (canonical ints)."""
    L = lib()
    L.or_p2_lookup.restype = C.c_int
    L.or_p2_lookup.argtypes = [C.c_int, C.POINTER(_P2Params)]
    p = _P2Params()
    assert L.or_p2_lookup(width, C.byref(p)) == 0, f"no Poseidon2 parameters for width {width}"
    rp = p.rounds_p
    diag = [int(p.diag[i]) for i in range(width)]
    ext = [[int(p.ext_rc[r * width + i]) for i in range(width)] for r in range(8)]
    internal = [int(p.int_rc[i]) for i in range(rp)]
    return rp, diag, ext, internal


def f_inv(a: int) -> int:
    return int(lib().or_f_inv(a))


# ---- commit stage (PARITY UNPINNED, see oracle/commit.c) -------------------------------------

def _setup_commit():
    L = lib()
    if getattr(L, "_commit_ready", False):
        return L
    for name in ("or_lde_naive", "or_lde_fft"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.or_merkle_commit.restype = C.c_int
    L.or_merkle_commit.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.or_merkle_verify.restype = C.c_int
    L.or_merkle_verify.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L._commit_ready = True
    return L


def lde(mat, log_blowup=1, naive=False) -> np.ndarray:
    L = _setup_commit()
    mat = _u32(mat)
    n, w = mat.shape
    log_n = n.bit_length() - 1
    out = np.empty((n << log_blowup, w), dtype=np.uint32)
    f = L.or_lde_naive if naive else L.or_lde_fft
    assert f(log_n, w, log_blowup, mat.ctypes.data, out.ctypes.data) == 0
    return out


def merkle_commit(lde_mats):
    """lde_mats: list of [2^k, w] arrays (already extended).  Returns (root, digests)."""
    L = _setup_commit()
    mats = [_u32(m) for m in lde_mats]
    n = len(mats)
    ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in mats])
    lh = np.array([m.shape[0].bit_length() - 1 for m in mats], dtype=np.uint32)
    ws = np.array([m.shape[1] for m in mats], dtype=np.uint32)
    log_max = int(lh.max())
    digests = np.zeros(((2 << log_max) - 1, 8), dtype=np.uint32)
    root = np.zeros(8, dtype=np.uint32)
    assert L.or_merkle_commit(n, C.cast(ptrs, C.c_void_p), lh.ctypes.data, ws.ctypes.data, digests.ctypes.data, root.ctypes.data) == 0
    return root, digests


# ---- the CPU port timed by bench.py's cpu_baseline (oracle/cpu_port.c); checked against the functions above by tests/test_cpu_port.py

def usable_cores() -> int:
    """Cores this process may actually use: its affinity mask capped by the cgroup CPU quota (cpu.max) -- an OpenMP team of
    one thread per host core on a 16-core quota spends its time being throttled."""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_port_set_threads(n: int) -> None:
    lib().cp_set_threads(int(n))


def cpu_port_p2_hash8(width: int, states) -> np.ndarray:
    """PoseidonChipset::hash of every row of `states` ([n][width], canonical) on all host cores, sixteen rows per AVX-512 register set
    (oracle/cpu_port.c: cp_p2_hash8): the CPU leg of BASELINE config 2.  Raises when the CPU has no AVX-512."""
    L = lib()
    x = _u32(states)
    n, w = x.shape
    assert w == width
    out = np.empty((n, 8), dtype=np.uint32)
    L.cp_p2_hash8.restype = C.c_int
    L.cp_p2_hash8.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    if L.cp_p2_hash8(width, n, x.ctypes.data, out.ctypes.data) != 0:
        raise RuntimeError("cp_p2_hash8: unknown width or no AVX-512 on this CPU")
    return out


def cpu_port_lde(mat, log_blowup=1) -> np.ndarray:
    L = lib()
    mat = _u32(mat)
    n, w = mat.shape
    out = np.empty((n << log_blowup, w), dtype=np.uint32)
    L.cp_lde.restype = C.c_int
    L.cp_lde.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    assert L.cp_lde(n.bit_length() - 1, w, log_blowup, mat.ctypes.data, out.ctypes.data, 0) == 0
    return out


def cpu_port_commit_round(mats, log_blowup=1) -> np.ndarray:
    """Root (canonical) of the commitment of the trace matrices `mats` (canonical, 2^k x w): coset LDEs + Merkle tree, all in C."""
    L = lib()
    mats = [_u32(m) for m in mats]
    n = len(mats)
    ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in mats])
    ln = np.array([m.shape[0].bit_length() - 1 for m in mats], dtype=np.uint32)
    ws = np.array([m.shape[1] for m in mats], dtype=np.uint32)
    root = np.zeros(8, dtype=np.uint32)
    L.cp_commit_round.restype = C.c_int
    L.cp_commit_round.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    assert L.cp_commit_round(n, C.cast(ptrs, C.c_void_p), ln.ctypes.data, ws.ctypes.data, log_blowup, root.ctypes.data) == 0
    return root


def merkle_verify(log_heights, widths, index, rows, path, root) -> bool:
    L = _setup_commit()
    lh = np.asarray(log_heights, dtype=np.uint32)
    ws = np.asarray(widths, dtype=np.uint32)
    rows = _u32(rows)
    path = _u32(path)
    root = _u32(root)
    return bool(L.or_merkle_verify(len(lh), lh.ctypes.data, ws.ctypes.data, index, rows.ctypes.data, path.ctypes.data, root.ctypes.data))


def set_p16(rounds_p=None, ext_rc=None, int_rc=None, diag=None, scale=1) -> None:
    """Installs the width-16 permutation tables of a protocol profile in the oracle (Merkle tree, sponge, transcript);
    set_p16() with no arguments restores the default (the reference's BabyBearConfig16)."""
    L = _setup_commit()
    L.or_set_p16.restype = C.c_int
    L.or_set_p16.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    if ext_rc is None:
        assert L.or_set_p16(0, None, None, None, 1) == 0
        return
    e, i, d = _u32(ext_rc).reshape(-1), _u32(int_rc).reshape(-1), _u32(diag).reshape(-1)
    assert e.size == 128 and d.size == 16 and i.size >= rounds_p
    assert L.or_set_p16(rounds_p, e.ctypes.data, i.ctypes.data, d.ctypes.data, scale) == 0


def perm16(state):
    """The width-16 permutation currently installed (canonical ints in, canonical ints out)."""
    L = _setup_commit()
    L.or_perm16.restype = None
    L.or_perm16.argtypes = [C.c_void_p]
    s = _u32(state).copy()
    L.or_perm16(s.ctypes.data)
    return [int(x) for x in s]
