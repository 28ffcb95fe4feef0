"""ORACLE directory (test infrastructure, not product code): driver of the CPU port of trace generation (oracle/cpu_trace.c).

`compile_func` walks a function of the oracle's toplevel (oracle/lair.py bytecode) ONCE into the flat int program cpu_trace.c
interprets -- every static decision taken here: which multiplications / inversions own an aux column (the degree rule of
/root/reference/src/lair/func_chip.rs:202-229 as oracle/lair.py: generate_trace applies it), how many looked-up values each
call / preimage / load / store brings, which calls carry a depth comparison.  `flatten` turns the oracle's query record into
the per-row headers and hint streams (the values the reference's `populate_row` finds in hash maps,
/root/reference/src/lair/trace.rs:328-373), in the order the program consumes them.  `generate_trace` = both + the C loop;
tests/test_cpu_trace.py requires it to equal oracle/lair.py: generate_trace word for word.
Only tests/ and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lair as ol

(OP_ASSERT_NE, OP_CONTAINS, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_INV, OP_NOT, OP_HINT, OP_REQ, OP_DEPTH, OP_EXTERN, OP_RETURN, OP_MATCH) = range(1, 15)
EXTERN_KIND = {"hasher3": 1, "hasher4": 1, "hasher5": 1, "u64_add": 2, "u64_sub": 3, "u64_mul": 4, "u64_divrem": 5, "u64_lessthan": 6,
               "u64_iszero": 7, "big_num_lessthan": 8}
P = ol.P


class FuncProgram:
    def __init__(self, top, name):
        self.top, self.f = top, top.funcs[top.index[name]]
        self.lay = top.layout(self.f)
        self.width = 1 + sum(self.lay[k] for k in ("input", "output", "aux", "sel"))
        self.words = []
        self.max_vars = 0
        self._block(self.f["body"], [1] * self.f["input_size"])
        self.prog = np.array(self.words, dtype=np.int32)

    def _block(self, blk, degs):
        """Appends the block's code; `degs` = degree (0 constant / 1 variable) of every variable so far.  Returns nothing: a
        block ends in a return or a match whose targets are patched in."""
        w = self.words
        degs = list(degs)
        for op in blk["ops"]:
            k = op[0]
            if k == "assert_eq":
                continue
            if k == "assert_ne":
                w += [OP_ASSERT_NE, len(op[1])] + list(op[1]) + list(op[2])
            elif k == "contains":
                w += [OP_CONTAINS, len(op[1]), op[2]] + list(op[1])
            elif k == "const":
                w += [OP_CONST, op[1] % P]
                degs.append(0)
            elif k in ("add", "sub"):
                w += [OP_ADD if k == "add" else OP_SUB, op[1], op[2]]
                degs.append(max(degs[op[1]], degs[op[2]]))
            elif k == "mul":
                d = degs[op[1]] + degs[op[2]]
                w += [OP_MUL, op[1], op[2], 1 if d >= 2 else 0]
                degs.append(d if d < 2 else 1)
            elif k == "inv":
                w += [OP_INV, op[1], 1 if degs[op[1]] else 0]
                degs.append(1 if degs[op[1]] else 0)
            elif k == "not":
                w += [OP_NOT, op[1], 1 if degs[op[1]] else 0]
                degs.append(1 if degs[op[1]] else 0)
            elif k in ("call", "preimg"):
                g = self.top.funcs[op[1]]
                n = g["output_size"] if k == "call" else g["input_size"]
                w += [OP_HINT, n, OP_REQ]
                degs += [1] * n
                if g["partial"]:
                    w.append(OP_DEPTH)
            elif k == "store":
                w += [OP_HINT, 1, OP_REQ]
                degs.append(1)
            elif k == "load":
                w += [OP_HINT, op[1], OP_REQ]
                degs += [1] * op[1]
            elif k == "extern":
                chip = self.top.chips[op[1]]
                w += [OP_EXTERN, EXTERN_KIND[chip.name], len(op[2])] + list(op[2]) + [chip.require_size]
                degs += [1] * chip.ret_size
            elif k == "range_u8":
                w += [OP_REQ] * ((len(op[1]) + 1) // 2)
            elif k in ("emit", "breakpoint", "debug"):
                continue
            else:
                raise AssertionError(k)
        self.max_vars = max(self.max_vars, len(degs))
        c = blk["ctrl"]
        if c[0] == "return":
            w += [OP_RETURN, c[1]]
            return
        _, vs, cases, _, d = c
        keys = sorted(cases)
        w += [OP_MATCH, len(vs)] + list(vs) + [len(keys)]
        table = len(w)
        w += [0] * (len(keys) * (len(vs) + 1) + 1)
        targets = {}
        for i, key in enumerate(keys):
            blk2 = cases[key]
            if id(blk2) not in targets:
                targets[id(blk2)] = len(w)
                self._block(blk2, degs)
            for j, kv in enumerate(key):
                w[table + i * (len(vs) + 1) + j] = kv % P
            w[table + i * (len(vs) + 1) + len(vs)] = targets[id(blk2)]
        if d is not None:
            w[table + len(keys) * (len(vs) + 1)] = len(w)
            self._block(d, degs)
        else:
            w[table + len(keys) * (len(vs) + 1)] = -1


def flatten(top, name, q, shard_index=0, max_shard_size=1 << 22):
    """Per-row headers [args | outputs | provide(2) | depth | 2 depth requires(4)] and the hint streams of the function's rows in
    the shard: a walk of each row like oracle/lair.py: generate_trace, recording what it looks up instead of what it computes."""
    f = top.funcs[top.index[name]]
    items = list(q.func[f["index"]].items())
    start = shard_index * max_shard_size
    end = min((shard_index + 1) * max_shard_size, len(items))
    n = max(end - start, 0)
    partial = bool(f["partial"])
    stride = f["input_size"] + f["output_size"] + 2 + (5 if partial else 0)
    hdr = np.zeros((max(n, 1), stride), dtype=np.uint32)
    hints, offs = [], []
    for i in range(n):
        args, res = items[start + i]
        row = list(args) + list(res.output) + [res.provide[0], res.provide[1]]
        dreqs = iter(res.depth_requires)
        reqs = iter(res.requires)
        if partial:
            row.append(res.depth)
            for _ in range(2):
                r = next(dreqs)
                row += [r[0], r[1]]
        hdr[i] = row
        offs.append(len(hints))
        m = list(args)
        blk = f["body"]
        while True:
            for op in blk["ops"]:
                k = op[0]
                if k == "const":
                    m.append(op[1] % P)
                elif k in ("add", "sub"):
                    m.append((m[op[1]] + m[op[2]]) % P if k == "add" else (m[op[1]] - m[op[2]]) % P)
                elif k == "mul":
                    m.append(m[op[1]] * m[op[2]] % P)
                elif k == "inv":
                    m.append(ol.inv(m[op[1]]))
                elif k == "not":
                    m.append(1 if m[op[1]] == 0 else 0)
                elif k in ("call", "preimg"):
                    g = top.funcs[op[1]]
                    key = tuple(m[v] for v in op[2])
                    if k == "call":
                        r = q.func[op[1]][key]
                        vals = r.output
                    else:
                        inp = q.inv[op[1]][key]
                        r = q.func[op[1]][inp]
                        vals = inp
                    m += list(vals)
                    hints += list(vals)
                    rq = next(reqs)
                    hints += [rq[0], rq[1]]
                    if g["partial"]:
                        hints.append(r.depth)
                        dq = next(dreqs)
                        hints += [dq[0], dq[1]]
                elif k == "store":
                    vals = tuple(m[v] for v in op[1])
                    ptr = ol._nonce_of(q.mem[ol.MEM_TABLE_SIZES.index(len(vals))], vals) + 1
                    m.append(ptr)
                    rq = next(reqs)
                    hints += [ptr, rq[0], rq[1]]
                elif k == "load":
                    mm = q.mem[ol.MEM_TABLE_SIZES.index(op[1])]
                    vals = list(mm.keys())[m[op[2]] - 1]
                    m += list(vals)
                    rq = next(reqs)
                    hints += list(vals) + [rq[0], rq[1]]
                elif k == "extern":
                    chip = top.chips[op[1]]
                    m += _extern_return(chip, [m[v] for v in op[2]])
                    for _ in range(chip.require_size):
                        rq = next(reqs)
                        hints += [rq[0], rq[1]]
                elif k == "range_u8":
                    for _ in range((len(op[1]) + 1) // 2):
                        rq = next(reqs)
                        hints += [rq[0], rq[1]]
            c = blk["ctrl"]
            if c[0] == "return":
                break
            _, vs, cases, _, d = c
            blk = cases.get(tuple(m[v] for v in vs), d)
    offs.append(len(hints))
    return n, hdr, np.array(hints + [0], dtype=np.uint32), np.array(offs, dtype=np.uint64)


def _extern_return(chip, inp):
    """What the chip's populate_witness RETURNS (the variables the walk continues with); the witness itself is the C side's."""
    from . import binding

    name = chip.name
    if name.startswith("hasher"):
        return [int(v) for v in binding.p2_permute(chip.input_size, np.array(inp, dtype=np.uint32))[0]]
    u64 = lambda x: sum((b & 0xFF) << (8 * i) for i, b in enumerate(x))
    le = lambda v, n: [(v >> (8 * i)) & 0xFF for i in range(n)]
    a = u64(inp[:8])
    b = u64(inp[8:16]) if len(inp) >= 16 else 0
    if name == "u64_add":
        return le((a + b) % (1 << 64), 8)
    if name == "u64_sub":
        return le((a - b) % (1 << 64), 8)
    if name == "u64_mul":
        return le((a * b) % (1 << 64), 8)
    if name == "u64_divrem":
        return le(a // b, 8) + le(a % b, 8)
    if name == "u64_lessthan":
        return [1 if a < b else 0]
    if name == "u64_iszero":
        return [1 if a == 0 else 0]
    if name == "big_num_lessthan":
        for i in reversed(range(8)):
            if inp[i] != inp[8 + i]:
                l, r = inp[i], inp[8 + i]
                for j in reversed(range(4)):
                    x, y = (l >> (8 * j)) & 0xFF, (r >> (8 * j)) & 0xFF
                    if x != y:
                        return [1 if x < y else 0]
        return [0]
    raise NotImplementedError(name)


_LIB = None


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    from . import binding

    binding.build()
    L = binding._setup_commit()
    L.cp2_trace_func.restype = C.c_int
    L.cp2_trace_func.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_uint32] * 3 + [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    _LIB = L
    return L


def run(prog: FuncProgram, n, hdr, hints, offs, height, nonce_start=0):
    """The C row loop over flattened inputs -> canonical trace [height][width]."""
    L = _lib()
    f, lay = prog.f, prog.lay
    out = np.empty((height, prog.width), dtype=np.uint32)
    hdr = np.ascontiguousarray(hdr, dtype=np.uint32)
    rc = L.cp2_trace_func(prog.prog.ctypes.data, prog.width, lay["input"], lay["output"], lay["aux"], 1 if f["partial"] else 0, n, height, nonce_start,
                          hdr.ctypes.data, hdr.shape[1], hints.ctypes.data, offs.ctypes.data, prog.max_vars, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"cp2_trace_func failed with {rc}")
    return out


def generate_trace(top, name, q, shard_index=0, max_shard_size=1 << 22):
    prog = FuncProgram(top, name)
    n, hdr, hints, offs = flatten(top, name, q, shard_index, max_shard_size)
    return run(prog, n, hdr, hints, offs, ol.next_pow2(n), nonce_start=shard_index * max_shard_size)
