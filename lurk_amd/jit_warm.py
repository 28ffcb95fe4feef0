"""Warms the persistent AIR code-object cache (csrc/jit.cpp) without a device: hiprtc cross-compiles gfx950 anywhere.

A machine's AIR programs are a function of its toplevel alone, so the kernels `Machine.compile_airs` will ask for can be
compiled ahead of time -- by `__graft_entry__.build()` for the bench's default machine, or by a deployment for its own
toplevel.  Chips are compiled in parallel worker processes (a Poseidon2 chip takes about a minute on one core)."""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ProcessPoolExecutor

COMPILE_MIN_LOG_ROWS = 17     # chips at least this tall run compiled kernels ...
COMPILE_MIN_INSTRS = 2000     # ... and so do chips whose constraint program is this long whatever their height (Poseidon2 chips)


def wants_compile(log_rows: int, constraint_instrs: int) -> bool:
    return log_rows >= COMPILE_MIN_LOG_ROWS or constraint_instrs >= COMPILE_MIN_INSTRS


def _compile_one(job):
    source, lurk_chips, kind, arg, n_public = job
    from . import _native as N
    from . import air, lair

    if kind == "trace":  # the function's trace generator (csrc/trace_jit.cpp)
        top = lair.Toplevel(source, lurk_chips=lurk_chips)
        log = C.create_string_buffer(2048)
        r = N.lib.lurkhip_trace_compile_check(top.handle, arg, log, 2048)
        return "trace kernel of function #%d" % arg, int(r), log.value.decode("utf-8", "replace")
    if kind == "func":
        top = lair.Toplevel(source, lurk_chips=lurk_chips)
        a = air.ChipAir.for_func(top, arg)
    elif kind == "mem":
        a = air.ChipAir.for_mem(arg)
    elif kind == "bytes":
        a = air.ChipAir.for_bytes()
    else:
        a = air.ChipAir.for_entrypoint(arg, n_public)
    log = C.create_string_buffer(2048)
    r = N.lib.lurkhip_air_compile_check(a.handle, log, 2048)
    return a.name, int(r), log.value.decode("utf-8", "replace")


def warm(source: str, lurk_chips: bool, rows_by_func: dict, mem_lens=(2, 3, 4, 5, 6, 8), workers: int | None = None, verbose=False,
         min_log_rows: int = COMPILE_MIN_LOG_ROWS, entry: str | None = None, n_public: int = 0):
    """rows_by_func: function name -> expected number of trace rows.  Compiles (or finds in the cache) the kernels of every
    function chip `wants_compile` selects at those heights (min_log_rows = 0: of every function chip, whatever its height -- a
    prover of small programs compiles them all, Machine.compile_airs(min_log_rows=0)), of the given memory tables, of the byte
    chip and, given `entry`, of that function's entrypoint chip."""
    from . import air, lair

    top = lair.Toplevel(source, lurk_chips=lurk_chips)
    jobs = []
    for name, rows in rows_by_func.items():
        idx = top.func_index(name)
        a = air.ChipAir.for_func(top, idx)
        log_rows = max(0, (max(rows, 1) - 1).bit_length())
        if log_rows >= min_log_rows or wants_compile(log_rows, a.constraint_instrs):
            jobs.append((source, lurk_chips, "func", idx, 0))
        if log_rows >= min_log_rows:
            jobs.append((source, lurk_chips, "trace", idx, 0))
    jobs += [(source, lurk_chips, "mem", ml, 0) for ml in mem_lens]
    jobs.append((source, lurk_chips, "bytes", 0, 0))
    if entry is not None:
        jobs.append((source, lurk_chips, "entry", top.func_index(entry), n_public))
    workers = workers or min(len(jobs), os.cpu_count() or 1)
    out = []
    with ProcessPoolExecutor(max_workers=workers) as ex:
        for name, r, log in ex.map(_compile_one, jobs):
            out.append((name, r))
            if verbose:
                print(f"  jit cache: {name}: {'%d bytes' % r if r > 0 else 'FAILED ' + log[:200]}", flush=True)
    return out
