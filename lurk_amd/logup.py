"""Mirror of the reference's in-tree LogUp module (/root/reference/src/logup/, dead code upstream) over the C ABI
(lurk_amd/csrc/logup.hip): `generate_multiplicities_trace`, `generate_permutation_trace` (logup/trace.rs:10-50,53-151) and
`eval_logup_constraints` (logup/air.rs:11-77).  An interaction is (values: [lc], is_real: lc or None), an lc (PairColLC,
src/air/symbolic/virtual_col.rs:8-13) is (terms: [(kind, index, weight)], constant) with kind IDENTITY / PREP / MAIN."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context, _addr, as_u32

IDENTITY, PREP, MAIN = 0, 1, 2


def encode_program(provides, requires) -> np.ndarray:
    w = [len(provides), len(requires)]

    def lc(form):
        terms, const = form
        w.append(len(terms))
        for kind, idx, weight in terms:
            w.extend([kind, idx, weight])
        w.append(const)

    for values, is_real in list(provides) + list(requires):
        w.append(0 if is_real is None else 1)
        if is_real is not None:
            lc(is_real)
        w.append(len(values))
        for v in values:
            lc(v)
    return np.array(w, dtype=np.uint32)


def _ef(x):
    return as_u32(np.array(list(x), dtype=np.uint32))


def multiplicities_trace(ctx: Context, multiplicities, z) -> np.ndarray:
    """multiplicities: [(traces, counts [height][len(traces)])] per provide -> [height][n_provides][4]."""
    height = len(multiplicities[0][1])
    n_traces = np.array([len(t) for t, _ in multiplicities], dtype=np.uint32)
    traces = np.array([x for t, _ in multiplicities for x in t], dtype=np.uint32)
    counts = [as_u32(np.array(c, dtype=np.uint32).reshape(height, -1)) for _, c in multiplicities]
    ptrs = (C.c_void_p * len(counts))(*[_addr(c) for c in counts])
    out = np.zeros((height, len(multiplicities), 4), dtype=np.uint32)
    zz = _ef(z)  # (kept alive across the call: _addr is a bare address)
    ctx.check(N.lib.lurkhip_logup_multiplicities(ctx.handle, height, len(multiplicities), _addr(n_traces), _addr(traces), C.cast(ptrs, C.c_void_p), _addr(zz), _addr(out)))
    return out


def permutation_trace(ctx: Context, identity, prep, main, mult, provides, requires, z, r, gamma, exclusive: bool = False):
    main = as_u32(np.array(main, dtype=np.uint32))
    height, main_w = main.shape
    prep_a = as_u32(np.array(prep, dtype=np.uint32)) if prep is not None else None
    prep_w = prep_a.shape[1] if prep_a is not None else 0
    ident = as_u32(np.array(identity, dtype=np.uint32))
    mult_a = as_u32(np.array(mult, dtype=np.uint32).reshape(height, -1)) if len(provides) else None
    prog = encode_program(provides, requires)
    n_int = len(provides) + len(requires)
    out = np.zeros((height, 1 + n_int, 4), dtype=np.uint32)
    total = np.zeros(4, dtype=np.uint32)
    zz, rr, gg = _ef(z), _ef(r), _ef(gamma)
    ctx.check(N.lib.lurkhip_logup_permutation_trace(ctx.handle, height, prep_w, main_w, _addr(ident), _addr(prep_a) if prep_a is not None else None, _addr(main),
                                                    _addr(mult_a) if mult_a is not None else None, _addr(prog), prog.size, _addr(zz), _addr(rr), _addr(gg),
                                                    int(exclusive), _addr(out), _addr(total)))
    return out, tuple(int(x) for x in total)


def eval_constraints(ctx: Context, perm_local, perm_next, mult, identity, prep, main, provides, requires, z, r, gamma, final_sum, selectors, air_order: bool = True):
    main = as_u32(np.array(main, dtype=np.uint32))
    n, main_w = main.shape
    prep_a = as_u32(np.array(prep, dtype=np.uint32)) if prep is not None else None
    prep_w = prep_a.shape[1] if prep_a is not None else 0
    pl, pn = as_u32(np.array(perm_local, dtype=np.uint32).reshape(n, -1)), as_u32(np.array(perm_next, dtype=np.uint32).reshape(n, -1))
    mult_a = as_u32(np.array(mult, dtype=np.uint32).reshape(n, -1)) if len(provides) else None
    prog = encode_program(provides, requires)
    n_int = len(provides) + len(requires)
    out = np.zeros((n, n_int + 3, 4), dtype=np.uint32)
    zz, rr, gg, fs = _ef(z), _ef(r), _ef(gamma), _ef(final_sum)
    ident = as_u32(np.array(identity, dtype=np.uint32))
    sels = as_u32(np.array(selectors, dtype=np.uint32).reshape(n, 3))
    ctx.check(N.lib.lurkhip_logup_eval_constraints(ctx.handle, n, prep_w, main_w, _addr(pl), _addr(pn), _addr(mult_a) if mult_a is not None else None,
                                                   _addr(ident), _addr(prep_a) if prep_a is not None else None, _addr(main),
                                                   _addr(prog), prog.size, _addr(zz), _addr(rr), _addr(gg), _addr(fs), _addr(sels), int(air_order), _addr(out)))
    return out
