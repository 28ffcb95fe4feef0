"""ORACLE (test infrastructure, not product code): pure-Python restatement of Lair.

Independent of lurk_amd/csrc/lair/*: own parser, own compiler, own interpreter, and a trace generator
that works the way the reference does -- by looking values up in the query record at trace time -- not
from the product's recorded hint stream.  Pinned by the reference's literal golden traces
(tests/golden/lair_traces.json).  Small cases only (pure-Python loops).

Follows:
  parse      /root/reference/src/lair/macros.rs (func! surface syntax)
  expand     /root/reference/src/lair/toplevel.rs:423-527,739-753
  compile    /root/reference/src/lair/toplevel.rs:255-283,303-322,529-574,755-879
  layout     /root/reference/src/lair/func_chip.rs:90-276
  execute    /root/reference/src/lair/execute.rs:436-784 (recursive here; fine for small cases)
  trace      /root/reference/src/lair/trace.rs:72-135,145-418; /root/reference/src/lair/memory.rs:30-69
  records    /root/reference/src/air/builder.rs:135-214; bytes /root/reference/src/gadgets/bytes/record.rs

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.
"""
from __future__ import annotations

import re
import sys
from dataclasses import dataclass, field

P = 2013265921
MEM_TABLE_SIZES = [2, 3, 4, 5, 6, 8]
DEPTH_W = 4
DEPTH_LESS_THAN_SIZE = 6


def inv(a):
    assert a % P != 0, "inverse of zero"
    return pow(a, P - 2, P)


# ------------------------------------------------------------------ parser
TOKEN = re.compile(r"\s*(?://[^\n]*\n\s*)*(=>|[A-Za-z_][A-Za-z0-9_]*!?|-?\d+|\S)")


def tokenize(src):
    out, i = [], 0
    while True:
        m = TOKEN.match(src, i)
        if not m:
            break
        out.append(m.group(1))
        i = m.end()
    return out


class Parser:
    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else None

    def eat(self, tok=None):
        v = self.t[self.i]
        if tok is not None and v != tok:
            raise SyntaxError(f"expected {tok!r}, got {v!r} at token {self.i}")
        self.i += 1
        return v

    def size_annot(self):
        self.eat("[")
        n = int(self.eat())
        self.eat("]")
        return n

    def args(self, env):
        out = []
        while self.peek() != ")":
            n = self.eat()
            out.append((n, env[n]))
            if self.peek() == ",":
                self.eat()
        self.eat(")")
        return out

    def const(self):
        return int(self.eat()) % P

    def let(self, env, ops):
        tg, paren = [], False
        if self.peek() == "(":
            paren = True
            self.eat()
            while self.peek() != ")":
                n, s = self.eat(), 1
                if self.peek() == ":":
                    self.eat()
                    s = self.size_annot()
                tg.append((n, s))
                if self.peek() == ",":
                    self.eat()
            self.eat(")")
        else:
            n, s = self.eat(), 1
            if self.peek() == ":":
                self.eat()
                s = self.size_annot()
            tg.append((n, s))
        self.eat("=")
        tok = self.peek()
        if re.fullmatch(r"-?\d+", tok):
            op = ("const", [(tg[0][0], 1)], [], [self.const()])
        elif tok == "[":
            self.eat()
            first = self.const()
            if self.peek() == ";":
                self.eat()
                arr = [first] * int(self.eat())
            else:
                arr = [first]
                while self.peek() == ",":
                    self.eat()
                    arr.append(self.const())
            self.eat("]")
            op = ("array", [(tg[0][0], len(arr))], [], arr)
        elif tok == "(":
            self.eat()
            op = ("slice", tg, self.args(env), None)
        else:
            fn = self.eat()
            if paren and self.peek() != "(":
                op = ("slice", tg, [(fn, env[fn])], None)
            else:
                self.eat("(")
                if fn in ("add", "sub", "mul", "div", "eq"):
                    a = self.args(env)
                    op = (fn, [(tg[0][0], 1 if fn == "eq" else a[0][1])], a, None)
                elif fn in ("inv", "not"):
                    a = self.args(env)
                    op = (fn, [(tg[0][0], a[0][1] if fn == "inv" else 1)], a, None)
                elif fn == "store":
                    op = ("store", [(tg[0][0], 1)], self.args(env), None)
                elif fn == "load":
                    op = ("load", tg, self.args(env), None)
                elif fn in ("call", "extern_call", "preimg"):
                    name = self.eat()
                    if self.peek() == ",":
                        self.eat()
                    op = (fn, tg, self.args(env), name)
                else:
                    raise SyntaxError(f"unknown op {fn}")
        ops.append(op)
        for n, s in op[1]:
            env[n] = s
        self.eat(";")

    def block(self, env):
        env = dict(env)
        constrained = True
        if self.peek() == "#":
            self.eat("#"), self.eat("["), self.eat("unconstrained"), self.eat("]")
            constrained = False
        ops = []
        while True:
            tok = self.peek()
            if tok == "let":
                self.eat()
                self.let(env, ops)
            elif tok == "return":
                self.eat()
                if self.peek() == "(":
                    self.eat()
                    ret = self.args(env)
                else:
                    n = self.eat()
                    ret = [(n, env[n])]
                if self.peek() == ";":
                    self.eat()
                self.eat("}")
                return {"ops": ops, "ctrl": ("return", ret), "constrained": constrained}
            elif tok == "if":
                self.eat()
                neg = False
                if self.peek() == "!":
                    self.eat()
                    neg = True
                n = self.eat()
                self.eat("{")
                inner = self.block(env)
                rest = self.block(env)
                t, f = (rest, inner) if neg else (inner, rest)
                return {"ops": ops, "ctrl": ("if", (n, env[n]), t, f), "constrained": constrained}
            elif tok == "match":
                self.eat()
                n = self.eat()
                self.eat("{")
                many, branches = False, []
                while self.peek() != "}":
                    pats = []
                    while True:
                        if self.peek() == "[":
                            many = True
                            self.eat()
                            arr = []
                            while self.peek() != "]":
                                arr.append(self.const())
                                if self.peek() == ",":
                                    self.eat()
                            self.eat("]")
                            pats.append(arr)
                        else:
                            pats.append([self.const()])
                        if self.peek() == ",":
                            self.eat()
                            continue
                        break
                    self.eat("=>")
                    self.eat("{")
                    b = self.block(env)
                    if many:
                        branches += [(p, b) for p in pats]
                    else:
                        branches.append(([p[0] for p in pats], b))
                self.eat("}")
                default = None
                if self.peek() == ";":
                    self.eat()
                    default = self.block(env)
                else:
                    self.eat("}")
                return {"ops": ops, "ctrl": ("matchmany" if many else "match", (n, env[n]), branches, default), "constrained": constrained}
            else:
                fn = self.eat()
                if fn == "breakpoint":
                    pass
                else:
                    self.eat("(")
                    kind = {"assert_eq!": "assert_eq", "assert_ne!": "assert_ne", "contains!": "contains", "range_u8!": "range_u8", "emit": "emit"}[fn]
                    ops.append((kind, [], self.args(env), None))
                self.eat(";")

    def func(self):
        f = {"partial": False, "invertible": False}
        while self.peek() in ("partial", "invertible"):
            f[self.eat()] = True
        self.eat("fn")
        f["name"] = self.eat()
        self.eat("(")
        env, params = {}, []
        while self.peek() != ")":
            n, s = self.eat(), 1
            if self.peek() == ":":
                self.eat()
                s = self.size_annot()
            params.append((n, s))
            env[n] = s
            if self.peek() == ",":
                self.eat()
        self.eat(")")
        self.eat(":")
        f["output_size"] = self.size_annot()
        self.eat("{")
        f["params"] = params
        f["body"] = self.block(env)
        return f

    def funcs(self):
        out = []
        while self.peek() is not None:
            out.append(self.func())
        return out


# ------------------------------------------------------------------ expand + compile
class Chip:
    """Extern chips usable from the oracle: Poseidon hashers and u64 add/sub/lessthan."""

    def __init__(self, name, input_size, output_size, witness_size, require_size, ret_size):
        self.name, self.input_size, self.output_size = name, input_size, output_size
        self.witness_size, self.require_size, self.ret_size = witness_size, require_size, ret_size


def lurk_chips():
    def h(n, w, rp):
        return Chip(n, w, 8, 8 + 16 * w + w + (rp - 1) + rp, 0, w)

    return [
        h("hasher3", 24, 21), h("hasher4", 32, 30), h("hasher5", 40, 38),
        Chip("u64_add", 16, 8, 8, 4, 8), Chip("u64_sub", 16, 8, 8, 4, 8), Chip("u64_mul", 16, 8, 16, 12, 8),
        Chip("u64_divrem", 16, 16, 62, 22, 16), Chip("u64_lessthan", 16, 1, 12, 1, 1), Chip("u64_iszero", 8, 1, 9, 0, 1),
        Chip("big_num_lessthan", 16, 1, 28, 7, 1),
    ]


class Toplevel:
    def __init__(self, src, chips=()):
        self.funcs_e = Parser(src).funcs()
        self.index = {f["name"]: i for i, f in enumerate(self.funcs_e)}
        self.chips = list(chips)
        self.chip_index = {c.name: i for i, c in enumerate(self.chips)}
        self.funcs = [self._compile(i, f) for i, f in enumerate(self.funcs_e)]

    # --- expand
    def _expand_block(self, b, ops, ctx):
        ops = list(ops)
        for op in b["ops"]:
            kind, out, inp, extra = op
            if kind == "div":
                t = (f"${ctx[0]}", inp[1][1])
                ctx[0] += 1
                ops.append(("inv", [t], [inp[1]], None))
                ops.append(("mul", out, [inp[0], t], None))
            elif kind == "eq":
                # /root/reference/src/lair/toplevel.rs:622-629: eq (like not) is defined on single elements only
                if inp[0][1] != 1 or inp[1][1] != 1 or out[0][1] != 1:
                    raise ValueError("eq needs size-1 operands")
                t = (f"${ctx[0]}", inp[0][1])
                ctx[0] += 1
                ops.append(("sub", [t], [inp[0], inp[1]], None))
                ops.append(("not", out, [t], None))
            else:
                ops.append(op)
        c = b["ctrl"]
        if c[0] == "return":
            ctrl = c
        elif c[0] == "if":
            _, x, t, f = c
            zero = (f"${ctx[0]}", x[1])
            ctx[0] += 1
            arr = [0] * x[1]
            tb = self._expand_block(t, [("array", [zero], [], arr), ("assert_ne", [], [x, zero], None)], ctx)
            fb = self._expand_block(f, [("array", [zero], [], arr), ("assert_eq", [], [x, zero], None)], ctx)
            ctrl = ("choose" if x[1] == 1 else "choosemany", x, [(arr, fb)], tb)
        elif c[0] == "match":
            _, v, branches, default = c
            nb = []
            for keys, blk in branches:
                pro = []
                if blk["constrained"]:
                    arr = (f"${ctx[0]}", len(keys))
                    ctx[0] += 1
                    pro = [("array", [arr], [], list(keys)), ("contains", [], [arr, v], None)]
                nb.append((list(keys), self._expand_block(blk, pro, ctx)))
            nd = None
            if default is not None:
                pro = []
                if default["constrained"]:
                    for keys, _ in branches:
                        for k in keys:
                            fv = (f"${ctx[0]}", 1)
                            ctx[0] += 1
                            pro += [("const", [fv], [], [k]), ("assert_ne", [], [v, fv], None)]
                nd = self._expand_block(default, pro, ctx)
            ctrl = ("choose", v, nb, nd)
        elif c[0] == "matchmany":
            _, v, branches, default = c
            nb = []
            for keys, blk in branches:
                pro = []
                if blk["constrained"]:
                    arr = (f"${ctx[0]}", len(keys))
                    ctx[0] += 1
                    pro = [("array", [arr], [], list(keys)), ("assert_eq", [], [v, arr], None)]
                nb.append((list(keys), self._expand_block(blk, pro, ctx)))
            nd = None
            if default is not None:
                pro = []
                if default["constrained"]:
                    for keys, _ in branches:
                        arr = (f"${ctx[0]}", len(keys))
                        ctx[0] += 1
                        pro += [("array", [arr], [], list(keys)), ("assert_ne", [], [v, arr], None)]
                nd = self._expand_block(default, pro, ctx)
            ctrl = ("choosemany", v, nb, nd)
        return {"ops": ops, "ctrl": ctrl}

    # --- compile
    def _compile(self, idx, fe):
        body = self._expand_block(fe["body"], [], [0])
        st = {"var": 0, "ret": 0, "link": {}, "partial": fe["partial"]}
        for p in fe["params"]:
            st["link"][p] = list(range(st["var"], st["var"] + p[1]))
            st["var"] += p[1]
        blk = self._compile_block(body, st)
        return {"name": fe["name"], "index": idx, "partial": fe["partial"], "invertible": fe["invertible"],
                "input_size": sum(s for _, s in fe["params"]), "output_size": fe["output_size"], "body": blk}

    def _new(self, v, st):
        st["link"][v] = list(range(st["var"], st["var"] + v[1]))
        st["var"] += v[1]

    def _flat(self, vs, st):
        return [i for v in vs for i in st["link"][v]]

    def _compile_block(self, b, st):
        ops = []
        for kind, out, inp, extra in b["ops"]:
            if kind in ("assert_ne", "assert_eq"):
                ops.append((kind, st["link"][inp[0]], st["link"][inp[1]]))
            elif kind == "contains":
                ops.append(("contains", st["link"][inp[0]], st["link"][inp[1]][0]))
            elif kind == "const":
                ops.append(("const", extra[0]))
                self._new(out[0], st)
            elif kind == "array":
                ops += [("const", f) for f in extra]
                self._new(out[0], st)
            elif kind in ("add", "sub", "mul"):
                a, bb = st["link"][inp[0]], st["link"][inp[1]]
                ops += [(kind, x, y) for x, y in zip(a, bb)]
                self._new(out[0], st)
            elif kind == "inv":
                ops += [("inv", x) for x in st["link"][inp[0]]]
                self._new(out[0], st)
            elif kind == "not":
                if len(st["link"][inp[0]]) != 1:  # toplevel.rs:616-621
                    raise ValueError("not needs a size-1 operand")
                ops.append(("not", st["link"][inp[0]][0]))
                self._new(out[0], st)
            elif kind in ("call", "preimg"):
                if self.funcs_e[self.index[extra]]["partial"] and not st["partial"]:  # toplevel.rs:640-642,668-670: assert!(ctx.partial)
                    raise ValueError(f"a total function may not call the partial function {extra}")
                ops.append((kind, self.index[extra], self._flat(inp, st)))
                for t in out:
                    self._new(t, st)
            elif kind == "store":
                ops.append(("store", self._flat(inp, st)))
                self._new(out[0], st)
            elif kind == "load":
                ops.append(("load", sum(s for _, s in out), st["link"][inp[0]][0]))
                for t in out:
                    self._new(t, st)
            elif kind == "slice":
                args = self._flat(inp, st)
                i = 0
                for pat in out:
                    st["link"][pat] = args[i:i + pat[1]]
                    i += pat[1]
            elif kind == "extern_call":
                ops.append(("extern", self.chip_index[extra], self._flat(inp, st)))
                for t in out:
                    self._new(t, st)
            elif kind == "emit":
                ops.append(("emit", self._flat(inp, st)))
            elif kind == "range_u8":
                ops.append(("range_u8", self._flat(inp, st)))
        c = b["ctrl"]
        if c[0] == "return":
            ctrl = ("return", st["ret"], self._flat(c[1], st))
            st["ret"] += 1
        else:
            kind, v, branches, default = c
            # the scrutinee is resolved HERE: the default block is compiled in the running scope (toplevel.rs:541-556) and may
            # rebind the same name (the real `eval` does: `let (head_tag, head) = call(eval, head_tag, ..)` inside a default)
            scrutinee = list(st["link"][v])
            cases, uniq = [], []
            for keys, blk in branches:
                saved = (st["var"], dict(st["link"]))
                cb = self._compile_block(blk, st)
                st["var"], st["link"] = saved
                uniq.append(cb)
                if kind == "choose":
                    cases += [((k,), cb) for k in keys]
                else:
                    cases.append((tuple(keys), cb))
            d = self._compile_block(default, st) if default is not None else None
            ctrl = (kind, scrutinee, dict(cases), uniq, d)
        return {"ops": ops, "ctrl": ctrl}

    # --- layout (func_chip.rs)
    def layout(self, f):
        aux = [2 + ((DEPTH_W + 3 * 2) if f["partial"] else 0)]
        sel = [0]

        def block(b, deg, aux_in):
            a = aux_in
            deg = list(deg)
            for op in b["ops"]:
                k = op[0]
                if k == "assert_ne":
                    a += len(op[1])
                elif k == "contains":
                    a += len(op[1]) - 1
                elif k == "const":
                    deg.append(0)
                elif k in ("add", "sub"):
                    deg.append(max(deg[op[1]], deg[op[2]]))
                elif k == "mul":
                    d = deg[op[1]] + deg[op[2]]
                    if d < 2:
                        deg.append(d)
                    else:
                        deg.append(1)
                        a += 1
                elif k == "inv":
                    if deg[op[1]] == 0:
                        deg.append(0)
                    else:
                        deg.append(1)
                        a += 1
                elif k == "not":
                    if deg[op[1]] == 0:
                        deg.append(0)
                    else:
                        deg.append(1)
                        a += 2
                elif k in ("call", "preimg"):
                    g = self.funcs_e[op[1]]
                    n = g["output_size"] if k == "call" else sum(s for _, s in g["params"])
                    a += n + 3
                    if g["partial"]:
                        a += DEPTH_W + DEPTH_LESS_THAN_SIZE + 3
                    deg += [1] * n
                elif k == "store":
                    a += 4
                    deg.append(1)
                elif k == "load":
                    a += op[1] + 3
                    deg += [1] * op[1]
                elif k == "extern":
                    c = self.chips[op[1]]
                    n = c.witness_size + 3 * c.require_size
                    a += n
                    deg += [1] * n
                elif k == "range_u8":
                    a += 3 * ((len(op[1]) + 1) // 2)
            c = b["ctrl"]
            if c[0] == "return":
                sel[0] += 1
                return a
            _, _, cases, uniq, d = c
            blocks = list(uniq) + ([d] if d is not None else [])
            return max([a] + [block(x, deg, a) for x in blocks])

        total_aux = block(f["body"], [1] * f["input_size"], aux[0])
        return {"nonce": 1, "input": f["input_size"], "output": f["output_size"], "aux": total_aux, "sel": sel[0]}


# ------------------------------------------------------------------ execution
@dataclass
class Result:
    output: list | None = None
    provide: list = field(default_factory=lambda: [0, 0])  # nonce, count
    requires: list = field(default_factory=list)
    depth: int = 0
    depth_requires: list = field(default_factory=list)


def new_lookup(rec, nonce):
    old = list(rec)
    rec[0] = nonce
    rec[1] += 1
    return old


class QueryRecord:
    def __init__(self, top: Toplevel):
        self.top = top
        self.func = [dict() for _ in top.funcs]        # args tuple -> Result (insertion ordered)
        self.inv = [dict() if f["invertible"] else None for f in top.funcs]
        self.mem = [dict() for _ in MEM_TABLE_SIZES]
        self.bytes = {}                                 # u16 -> 6 records
        self.public_values = None

    def byte_rec(self, i1, i2):
        return self.bytes.setdefault(i1 | (i2 << 8), [[0, 0] for _ in range(6)])

    def range_u8_iter(self, bs, nonce, reqs):
        bs = list(bs)
        for i in range(0, len(bs), 2):
            r = self.byte_rec(bs[i], bs[i + 1] if i + 1 < len(bs) else 0)
            reqs.append(new_lookup(r[0], nonce))

    def less_than(self, a, b, nonce, reqs):
        reqs.append(new_lookup(self.byte_rec(a, b)[2], nonce))
        return a < b

    def range_u16(self, v, nonce, reqs):  # gadgets/bytes/record.rs: the row of the table whose (i1, i2) spell v
        reqs.append(new_lookup(self.byte_rec(v & 0xFF, v >> 8)[1], nonce))


def le_bytes(v, n):
    return [(v >> (8 * i)) & 0xFF for i in range(n)]


def execute(top: Toplevel, name: str, args, q: QueryRecord, poseidon=None):
    sys.setrecursionlimit(20000)
    f = top.funcs[top.index[name]]
    res = Result()
    res.provide = [0, 1]
    q.func[f["index"]][tuple(args)] = res
    out, depth = _run(top, f, tuple(args), q, poseidon)
    q.public_values = list(args) + list(out) + (le_bytes(depth, 4) if f["partial"] else [])
    return out


def _nonce_of(qmap, key):
    return list(qmap.keys()).index(key)


def _run(top, f, args, q, poseidon):
    fi = f["index"]
    nonce = _nonce_of(q.func[fi], args)
    m = list(args)
    reqs, depths, dreqs = [], [], []
    blk = f["body"]
    while True:
        for op in blk["ops"]:
            k = op[0]
            if k == "assert_eq":
                assert all(m[a] == m[b] for a, b in zip(op[1], op[2]))
            elif k == "assert_ne":
                assert any(m[a] != m[b] for a, b in zip(op[1], op[2]))
            elif k == "contains":
                assert m[op[2]] in [m[a] for a in op[1]]
            elif k == "const":
                m.append(op[1])
            elif k == "add":
                m.append((m[op[1]] + m[op[2]]) % P)
            elif k == "sub":
                m.append((m[op[1]] - m[op[2]]) % P)
            elif k == "mul":
                m.append(m[op[1]] * m[op[2]] % P)
            elif k == "inv":
                m.append(inv(m[op[1]]))
            elif k == "not":
                m.append(1 if m[op[1]] == 0 else 0)
            elif k in ("call", "preimg"):
                g = top.funcs[op[1]]
                key = tuple(m[v] for v in op[2])
                inp = q.inv[op[1]][key] if k == "preimg" else key
                r = q.func[op[1]].get(inp)
                if r is None:
                    q.func[op[1]][inp] = Result()
                    _run(top, g, inp, q, poseidon)
                    r = q.func[op[1]][inp]
                elif r.output is None:
                    raise RuntimeError("Loop detected")
                m += list(inp if k == "preimg" else r.output)
                reqs.append(new_lookup(r.provide, nonce))
                if f["partial"] and g["partial"]:
                    depths.append(r.depth)
            elif k == "store":
                vals = tuple(m[v] for v in op[1])
                mm = q.mem[MEM_TABLE_SIZES.index(len(vals))]
                if vals not in mm:
                    mm[vals] = Result()
                m.append(_nonce_of(mm, vals) + 1)
                reqs.append(new_lookup(mm[vals].provide, nonce))
            elif k == "load":
                mm = q.mem[MEM_TABLE_SIZES.index(op[1])]
                vals = list(mm.keys())[m[op[2]] - 1]
                m += list(vals)
                reqs.append(new_lookup(mm[vals].provide, nonce))
            elif k == "extern":
                m += chip_execute(top.chips[op[1]], [m[v] for v in op[2]], nonce, q, reqs, poseidon)
            elif k == "range_u8":
                q.range_u8_iter([m[v] for v in op[1]], nonce, reqs)
        c = blk["ctrl"]
        if c[0] == "return":
            out = [m[v] for v in c[2]]
            r = q.func[fi][args]
            assert r.output is None
            if q.inv[fi] is not None:
                q.inv[fi][tuple(out)] = args
            depth = max([d + 1 for d in depths], default=0)
            if f["partial"]:
                q.range_u8_iter(le_bytes(depth, 4), nonce, dreqs)
                for d in depths:
                    lb, rb = le_bytes(d, 4), le_bytes(depth, 4)
                    for i in reversed(range(4)):
                        if lb[i] != rb[i]:
                            q.less_than(lb[i], rb[i], nonce, dreqs)
                            break
                r.depth = depth
            r.output, r.requires, r.depth_requires = out, reqs, dreqs
            return out, depth
        _, vs, cases, _, d = c
        blk = cases.get(tuple(m[v] for v in vs), d)
        assert blk is not None, "No match"


def chip_execute(chip, inp, nonce, q, reqs, poseidon):
    def u64(x):
        return sum(b << (8 * i) for i, b in enumerate(x))

    if chip.name.startswith("hasher"):
        return list(poseidon(chip.input_size, inp))[:8]
    a, b = u64(inp[:8]), u64(inp[8:16]) if len(inp) >= 16 else 0
    if chip.name in ("u64_add", "u64_sub"):
        r = (a + b if chip.name == "u64_add" else a - b) % (1 << 64)
        q.range_u8_iter(le_bytes(r, 8), nonce, reqs)
        return le_bytes(r, 8)
    if chip.name == "u64_lessthan":
        la, lb = le_bytes(a, 8), le_bytes(b, 8)
        for i in reversed(range(8)):
            if la[i] != lb[i]:
                return [1 if q.less_than(la[i], lb[i], nonce, reqs) else 0]
        q.less_than(0, 0, nonce, reqs)
        return [0]
    if chip.name == "u64_iszero":
        return [1 if a == 0 else 0]
    if chip.name == "u64_mul":  # gadgets/unsigned/mul.rs:24-64,125-133
        la, lb = le_bytes(a, 8), le_bytes(b, 8)
        res, carry = [], 0
        for k in range(8):
            o = sum(la[i] * lb[k - i] for i in range(k + 1)) + carry
            res.append(o & 0xFF)
            carry = (o >> 8) & 0xFFFF
            q.range_u16(carry, nonce, reqs)
        q.range_u8_iter(res, nonce, reqs)
        return res
    if chip.name == "u64_divrem":  # gadgets/unsigned/div_rem.rs:33-62
        assert b != 0, "expected input to be non-zero"
        qv, r = divmod(a, b)
        qb = qv * b
        q.range_u8_iter(le_bytes(qv, 8), nonce, reqs)
        lq, lb = le_bytes(qv, 8), le_bytes(b, 8)
        res, carry = [], 0
        for k in range(8):
            o = sum(lq[i] * lb[k - i] for i in range(k + 1)) + carry
            res.append(o & 0xFF)
            carry = (o >> 8) & 0xFFFF
            q.range_u16(carry, nonce, reqs)
        q.range_u8_iter(res, nonce, reqs)
        q.range_u8_iter(le_bytes(r, 8), nonce, reqs)

        def msb(l, rr, strict):
            ll, lr = le_bytes(l, 8), le_bytes(rr, 8)
            for i in reversed(range(8)):
                if ll[i] != lr[i]:
                    q.less_than(ll[i], lr[i], nonce, reqs)
                    return
            assert not strict
            q.less_than(0, 0, nonce, reqs)

        msb(r, b, True)
        msb(qb, a, False)
        return le_bytes(qv, 8) + le_bytes(r, 8)
    if chip.name == "big_num_lessthan":  # gadgets/big_num/cmp.rs:24-49
        l = r = 0
        for i in reversed(range(8)):
            if inp[i] != inp[8 + i]:
                l, r = inp[i], inp[8 + i]
                break
        for v in (l, r):
            by = le_bytes(v, 4)
            q.less_than(by[3], 0x78, nonce, reqs)
            q.range_u8_iter(by, nonce, reqs)
        bl, br = le_bytes(l, 4), le_bytes(r, 4)
        for i in reversed(range(4)):
            if bl[i] != br[i]:
                return [1 if q.less_than(bl[i], br[i], nonce, reqs) else 0]
        q.less_than(0, 0, nonce, reqs)
        return [0]
    raise NotImplementedError(chip.name)


# ------------------------------------------------------------------ trace generation (trace.rs)
def require_cols(rec):
    return [rec[0], rec[1], inv(rec[1] + 1)]


def next_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


def generate_trace(top: Toplevel, name: str, q: QueryRecord, shard_index=0, max_shard_size=1 << 22, witness=None):
    f = top.funcs[top.index[name]]
    lay = top.layout(f)
    width = 1 + lay["input"] + lay["output"] + lay["aux"] + lay["sel"]
    items = list(q.func[f["index"]].items())
    start = shard_index * max_shard_size
    end = min((shard_index + 1) * max_shard_size, len(items))
    n = max(end - start, 0)
    height = next_pow2(n)
    rows = [[0] * width for _ in range(height)]
    for i in range(height):
        rows[i][0] = (start + i) % P
    for i in range(n):
        args, res = items[start + i]
        row = rows[i]
        aux0 = 1 + lay["input"] + lay["output"]
        aux = []
        for k, o in enumerate(res.output):
            row[1 + lay["input"] + k] = o
        aux += [res.provide[0], res.provide[1]]
        dreqs = iter(res.depth_requires)
        reqs = iter(res.requires)
        if f["partial"]:
            aux += le_bytes(res.depth, 4)
            for _ in range(2):
                aux += require_cols(next(dreqs))
        for k, a in enumerate(args):
            row[1 + k] = a
        m = [(a, 1) for a in args]
        blk = f["body"]
        while True:
            for op in blk["ops"]:
                k = op[0]
                if k == "assert_ne":
                    found = False
                    for a, b in zip(op[1], op[2]):
                        d = (m[a][0] - m[b][0]) % P
                        if not found and d != 0:
                            aux.append(inv(d))
                            found = True
                        else:
                            aux.append(0)
                    assert found
                elif k == "contains":
                    b = m[op[2]][0]
                    acc = None
                    for a in op[1]:
                        d = (m[a][0] - b) % P
                        if acc is None:
                            acc = d
                        else:
                            acc = acc * d % P
                            aux.append(acc)
                elif k == "const":
                    m.append((op[1], 0))
                elif k in ("add", "sub"):
                    (a, da), (b, db) = m[op[1]], m[op[2]]
                    m.append(((a + b) % P if k == "add" else (a - b) % P, max(da, db)))
                elif k == "mul":
                    (a, da), (b, db) = m[op[1]], m[op[2]]
                    v = a * b % P
                    if da + db < 2:
                        m.append((v, da + db))
                    else:
                        m.append((v, 1))
                        aux.append(v)
                elif k == "inv":
                    a, da = m[op[1]]
                    v = inv(a)
                    if da == 0:
                        m.append((v, 0))
                    else:
                        m.append((v, 1))
                        aux.append(v)
                elif k == "not":
                    a, da = m[op[1]]
                    d = 0 if a == 0 else inv(a)
                    v = 1 if a == 0 else 0
                    if da == 0:
                        m.append((v, 0))
                    else:
                        m.append((v, 1))
                        aux += [d, v]
                elif k in ("call", "preimg"):
                    g = top.funcs[op[1]]
                    key = tuple(m[v][0] for v in op[2])
                    if k == "call":
                        r = q.func[op[1]][key]
                        vals = r.output
                    else:
                        inp = q.inv[op[1]][key]
                        r = q.func[op[1]][inp]
                        vals = inp
                    for v in vals:
                        m.append((v, 1))
                        aux.append(v)
                    aux += require_cols(next(reqs))
                    if g["partial"]:
                        aux += le_bytes(r.depth, 4)
                        lb, rb = le_bytes(r.depth, 4), le_bytes(res.depth, 4)
                        wit = [0] * 6
                        for j in reversed(range(4)):
                            if lb[j] != rb[j]:
                                wit[j], wit[4], wit[5] = 1, lb[j], rb[j]
                                break
                        aux += wit
                        aux += require_cols(next(dreqs))
                elif k == "store":
                    vals = tuple(m[v][0] for v in op[1])
                    mm = q.mem[MEM_TABLE_SIZES.index(len(vals))]
                    ptr = _nonce_of(mm, vals) + 1
                    m.append((ptr, 1))
                    aux.append(ptr)
                    aux += require_cols(next(reqs))
                elif k == "load":
                    mm = q.mem[MEM_TABLE_SIZES.index(op[1])]
                    vals = list(mm.keys())[m[op[2]][0] - 1]
                    for v in vals:
                        m.append((v, 1))
                        aux.append(v)
                    aux += require_cols(next(reqs))
                elif k == "extern":
                    chip = top.chips[op[1]]
                    wit, ret = witness(chip, [m[v][0] for v in op[2]])
                    assert len(wit) == chip.witness_size
                    m += [(v, 1) for v in ret]
                    aux += list(wit)
                    for _ in range(chip.require_size):
                        aux += require_cols(next(reqs))
                elif k == "range_u8":
                    for _ in range((len(op[1]) + 1) // 2):
                        aux += require_cols(next(reqs))
            c = blk["ctrl"]
            if c[0] == "return":
                assert next(reqs, None) is None and next(dreqs, None) is None
                row[1 + lay["input"] + lay["output"] + lay["aux"] + c[1]] = 1
                break
            _, vs, cases, _, d = c
            blk = cases.get(tuple(m[v][0] for v in vs), d)
        assert len(aux) <= lay["aux"], (len(aux), lay["aux"])
        for k, v in enumerate(aux):
            row[aux0 + k] = v % P
    return rows, width


def mem_trace(q: QueryRecord, mem_len: int):
    mm = q.mem[MEM_TABLE_SIZES.index(mem_len)]
    height = max(4, next_pow2(len(mm)))
    rows = [[0] * (4 + mem_len) for _ in range(height)]
    for i, (vals, r) in enumerate(mm.items()):
        rows[i] = [1, i + 1, r.provide[0], r.provide[1]] + list(vals)
    return rows


def bytes_trace(q: QueryRecord, shard_index=0):
    rows = [[0] * 13 for _ in range(1 << 16)]
    if shard_index == 0 and q.bytes:
        for i in range(1 << 16):
            rows[i][0] = 1
        for key, recs in q.bytes.items():
            for k, r in enumerate(recs):
                rows[key][1 + 2 * k] = r[0]
                rows[key][2 + 2 * k] = r[1]
    return rows


# ------------------------------------------------------------------ compiled-toplevel exchange ("LBC1")
# The oracle compiler's bytecode in the flat u32 format of lurk_amd/csrc/lair/bytecode_io.cpp (its header comment is the
# grammar; the types are /root/reference/src/lair/bytecode.rs:12-146).  Written from the oracle's own structures, so a
# word-for-word match with the product's export means two independent compilers agree on every index.
_OP_TAGS = {"assert_eq": 0, "assert_ne": 1, "contains": 2, "const": 3, "add": 4, "sub": 5, "mul": 6, "inv": 7, "not": 8,
            "call": 9, "preimg": 10, "store": 11, "load": 12, "extern": 13, "emit": 14, "range_u8": 15}


def to_bytecode(top: Toplevel):
    w = []

    def s(text):
        b = text.encode()
        w.append(len(b))
        for i in range(0, len(b), 4):
            w.append(int.from_bytes(b[i:i + 4].ljust(4, b"\0"), "little"))

    def lst(xs):
        w.append(len(xs))
        w.extend(int(x) for x in xs)

    def block(b):
        """writes the block, returns its return idents"""
        w.append(len(b["ops"]))
        for op in b["ops"]:
            k = op[0]
            w.append(_OP_TAGS[k])
            if k in ("assert_eq", "assert_ne"):
                lst(op[1]); lst(op[2])
            elif k == "contains":
                lst(op[1]); w.append(op[2])
            elif k == "const":
                w.append(op[1] % P)
            elif k in ("add", "sub", "mul"):
                w.extend([op[1], op[2]])
            elif k in ("inv", "not"):
                w.append(op[1])
            elif k in ("call", "preimg", "extern"):
                w.append(op[1]); lst(op[2])
            elif k in ("store", "emit", "range_u8"):
                lst(op[1])
            elif k == "load":
                w.extend([op[1], op[2]])
        c = b["ctrl"]
        idents = []
        if c[0] == "return":
            w.append(0)
            w.append(c[1])
            lst(c[2])
            idents = [c[1]]
        else:
            kind, v, cases, uniq, d = c
            if kind == "choose":
                w.extend([1, v[0], len(uniq)])
                for u in uniq:
                    idents += block(u)
                keys = sorted(cases)
                w.append(len(keys))
                for key in keys:
                    w.append(key[0] % P)
                    w.append(next(i for i, u in enumerate(uniq) if u is cases[key]))
            else:
                w.append(2)
                lst(v)
                keys = sorted(cases)
                w.append(len(keys))
                by_arm = {}
                for key in keys:
                    lst([x % P for x in key])
                    by_arm[id(cases[key])] = block(cases[key])
                # `return_idents` is the list the compiler collected, i.e. in SOURCE order of the arms
                # (/root/reference/src/lair/toplevel.rs:557-570: compiled in source order, then sorted by key for storage)
                for u in uniq:
                    idents += by_arm[id(u)]
            w.append(1 if d is not None else 0)
            if d is not None:
                idents += block(d)
        lst(idents)
        return idents

    w += [0x3143424C, 1, len(top.chips), len(top.funcs)]
    for c in top.chips:
        s(c.name)
    for f in top.funcs:
        s(f["name"])
        w += [(1 if f["invertible"] else 0) | (2 if f["partial"] else 0), f["input_size"], f["output_size"]]
        block(f["body"])
    return w
