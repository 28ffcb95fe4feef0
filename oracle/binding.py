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


def f_inv(a: int) -> int:
    return int(lib().or_f_inv(a))
