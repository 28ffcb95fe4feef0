"""ORACLE directory (test infrastructure, not product code): C emitter of the CPU port's chip evaluators.

The oracle's AIR (oracle/air.py) is a *numeric* walk: `air.eval(builder)` asserts values and records interactions of one row
pair.  Here the same walk runs ONCE per chip over symbolic values (a hash-consed expression DAG whose leaves are the cells of
the local / next / preprocessed rows, the public values and the three domain selectors), and the DAG is printed as a
straight-line C function on Montgomery words -- the CPU counterpart of what the reference gets from monomorphising
`Air::eval` over a prover folder (/root/reference/src/lair/air.rs:158-552 etc. through sphinx's ProverConstraintFolder
[UPSTREAM-RECALL]).  Independent of lurk_amd/csrc/lair/air.cpp and jit.cpp: nothing of the product is imported.

Output per chip:  chipK_inter(loc, prep, pub, out)         every interaction of one row: multiplicity, then its tuple
                  chipK_full(loc, nxt, pl, pn, pub, sel, cons, inter)   constraints + interactions of one row pair
plus the `cp_chips[]` table cpu_step.c walks.  Only tests/ and bench.py's cpu_baseline leg use this."""
from __future__ import annotations

P = 2013265921
R_MONT = (1 << 32) % P


class Graph:
    """Hash-consed expression DAG over the base field; nodes are created in topological order."""

    def __init__(self):
        self.nodes = []   # (op, a, b): op in {"const", "in", "add", "sub", "mul"}
        self.index = {}

    def node(self, op, a=0, b=0):
        key = (op, a, b)
        i = self.index.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self.index[key] = i
        return i

    def const(self, v):
        return self.node("const", v % P)

    def is_const(self, i):
        return self.nodes[i][0] == "const"

    def cval(self, i):
        return self.nodes[i][1]

    def add(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) + self.cval(b))
        if self.is_const(a) and self.cval(a) == 0:
            return b
        if self.is_const(b) and self.cval(b) == 0:
            return a
        if a > b:
            a, b = b, a
        return self.node("add", a, b)

    def sub(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) - self.cval(b))
        if self.is_const(b) and self.cval(b) == 0:
            return a
        if a == b:
            return self.const(0)
        return self.node("sub", a, b)

    def mul(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) * self.cval(b))
        for x, y in ((a, b), (b, a)):
            if self.is_const(x):
                if self.cval(x) == 0:
                    return x
                if self.cval(x) == 1:
                    return y
        if a > b:
            a, b = b, a
        return self.node("mul", a, b)


class Sym:
    """A value of the walk: wraps a node; behaves like the ints oracle/air.py computes with."""

    __slots__ = ("g", "i")

    def __init__(self, g, i):
        self.g, self.i = g, i

    def _lift(self, o):
        if isinstance(o, Sym):
            return o.i
        return self.g.const(int(o))

    def __add__(self, o):
        return Sym(self.g, self.g.add(self.i, self._lift(o)))

    __radd__ = __add__

    def __sub__(self, o):
        return Sym(self.g, self.g.sub(self.i, self._lift(o)))

    def __rsub__(self, o):
        return Sym(self.g, self.g.sub(self._lift(o), self.i))

    def __mul__(self, o):
        return Sym(self.g, self.g.mul(self.i, self._lift(o)))

    __rmul__ = __mul__

    def __neg__(self):
        return Sym(self.g, self.g.sub(self.g.const(0), self.i))

    def __mod__(self, _):
        return self


ARRAYS = ("loc", "nxt", "pl", "pn", "pub", "sel")


def walk(air, width, prep_width, n_public):
    """Runs air.eval over symbolic rows; returns (graph, constraint nodes, [(is_send, mult node, [tuple nodes])])."""
    from . import air as oair

    g = Graph()
    arr = lambda k, n: [Sym(g, g.node("in", k, i)) for i in range(n)]
    b = oair.Builder(arr(0, width), arr(1, width), arr(2, prep_width), arr(3, prep_width), arr(4, n_public), tuple(arr(5, 3)))
    lift = lambda v: v.i if isinstance(v, Sym) else g.const(int(v))
    cons, sends, recvs = [], [], []
    b.assert_zero = lambda x, cond=None: cons.append(lift(x if cond is None else cond * x))
    b.send = lambda values, is_real: sends.append((lift(is_real), [lift(v) for v in values]))
    b.receive = lambda values, is_real: recvs.append((lift(is_real), [lift(v) for v in values]))
    air.eval(b)
    inter = [(True, m, v) for m, v in sends] + [(False, m, v) for m, v in recvs]
    return g, cons, inter


def _reachable(g, roots):
    seen, stack = set(), list(roots)
    while stack:
        i = stack.pop()
        if i in seen:
            continue
        seen.add(i)
        op, a, b = g.nodes[i]
        if op in ("add", "sub", "mul"):
            stack += [a, b]
    return seen


def _emit_body(g, roots, lines):
    """SSA statements for every node the roots need, in creation (= topological) order; returns node -> C expression."""
    need = _reachable(g, roots)
    name = {}
    for i in sorted(need):
        op, a, b = g.nodes[i]
        if op == "const":
            name[i] = f"{a * R_MONT % P}u"
        elif op == "in":
            name[i] = f"{ARRAYS[a]}[{b}]"
        else:
            fn = {"add": "madd", "sub": "msub", "mul": "mm"}[op]
            lines.append(f"    const uint32_t t{i} = {fn}({name[a]}, {name[b]});")
            name[i] = f"t{i}"
    return name


def emit_chip(k, air, name, width, prep_width, n_public):
    g, cons, inter = walk(air, width, prep_width, n_public)
    inter_roots = [m for _, m, _ in inter] + [x for _, _, v in inter for x in v]
    # the interactions of a Lair chip read its local row, its preprocessed row and the public values only
    for i in _reachable(g, inter_roots):
        op, a, _ = g.nodes[i]
        assert not (op == "in" and a in (1, 3, 5)), f"{name}: an interaction reads the next row or a selector"
    out = []
    lines = []
    nm = _emit_body(g, inter_roots, lines)
    out.append(f"static void chip{k}_inter(const uint32_t* restrict loc, const uint32_t* restrict pl, const uint32_t* restrict pub, uint32_t* restrict inter) {{")
    out += lines
    at = 0
    for _, m, vals in inter:
        for x in [m] + vals:
            out.append(f"    inter[{at}] = {nm[x]};")
            at += 1
    out.append("    (void)loc; (void)pl; (void)pub;\n}")
    inter_words = at
    lines = []
    nm = _emit_body(g, cons + inter_roots, lines)
    out.append(f"static void chip{k}_full(const uint32_t* restrict loc, const uint32_t* restrict nxt, const uint32_t* restrict pl, const uint32_t* restrict pn, "
               "const uint32_t* restrict pub, const uint32_t* restrict sel, uint32_t* restrict cons, uint32_t* restrict inter) {")
    out += lines
    for j, c in enumerate(cons):
        out.append(f"    cons[{j}] = {nm[c]};")
    at = 0
    for _, m, vals in inter:
        for x in [m] + vals:
            out.append(f"    inter[{at}] = {nm[x]};")
            at += 1
    out.append("    (void)loc; (void)nxt; (void)pl; (void)pn; (void)pub; (void)sel; (void)cons; (void)inter;\n}")
    n_sends = sum(1 for s, _, _ in inter if s)
    lens = ", ".join(str(len(v)) for _, _, v in inter) or "0"
    out.append(f"static const uint32_t chip{k}_tuple_len[] = {{{lens}}};")
    desc = (f'    {{"{name}", {width}, {prep_width}, {len(cons)}, {n_sends}, {len(inter) - n_sends}, {inter_words}, chip{k}_tuple_len, chip{k}_inter, chip{k}_full}},')
    return "\n".join(out), desc, {"n_cons": len(cons), "n_inter": len(inter), "nodes": len(g.nodes)}


HEADER = """/* GENERATED by oracle/cpu_emit.py from the oracle's AIR (oracle/air.py) -- test infrastructure, not product code. */
#include <stdint.h>
#include "cpu_step.h"
"""


def emit_machine(chips, n_public):
    """chips: [(air, name, width, prep_width)] by machine index.  Returns ([C source per translation unit], per-chip stats):
    one unit per chip (compiled in parallel) and a last one with the cp_chips[] table."""
    units, descs, stats, protos = [], [], [], []
    for k, (air, name, width, prep_width) in enumerate(chips):
        src, desc, st = emit_chip(k, air, name, width, prep_width, n_public)
        # the evaluators and the tuple-length table are referenced from the table unit: external linkage
        src = src.replace(f"static void chip{k}_inter(", f"void chip{k}_inter(").replace(f"static void chip{k}_full(", f"void chip{k}_full(")
        src = src.replace(f"static const uint32_t chip{k}_tuple_len[]", f"const uint32_t chip{k}_tuple_len[]")
        units.append(HEADER + src + "\n")
        descs.append(desc)
        stats.append(st)
        protos.append(f"void chip{k}_inter(const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*);\n"
                      f"void chip{k}_full(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);\n"
                      f"extern const uint32_t chip{k}_tuple_len[];")
    table = HEADER + "\n".join(protos) + "\nconst cp_chip cp_chips[] = {\n" + "\n".join(descs) + "\n};\n" + f"const int cp_n_chips = {len(chips)};\n"
    units.append(table)
    return units, stats
