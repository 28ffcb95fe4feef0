"""Random Lair programs in the `func!` surface syntax, for differential tests of the product's host Lair (parser, compiler,
layout, interpreter, trace kernels) against the oracle's independent Python one.  Every program terminates: a function's first
parameter is a counter, the first statement returns when it is zero, and every call passes the counter's predecessor.

Covered on purpose: nested `match` (single keys, several keys per arm, with and without fall-through), `if` / `if !`, array
matches, multi-output calls, memoised repeated calls, stores of every table width with loads, `div` (multiply by an inverse),
`eq` / `not`, assert_eq! / assert_ne! / contains! / range_u8!, array literals, invertible functions with `preimg`, partial
functions (depth columns)."""
import random

MEM_WIDTHS = (2, 3, 4, 5, 6, 8)


class _Fn:
    def __init__(self, name, n_extra, out, invertible=False):
        self.name, self.n_extra, self.out, self.invertible = name, n_extra, out, invertible


class Gen:
    def __init__(self, seed, n_funcs=4, partial=None, const_times_var=False):
        # const_times_var: allow mul(constant, variable).  The reference's AIR gives that product an aux column
        # (/root/reference/src/lair/air.rs:345-358) while its layout and trace generator do not (func_chip.rs:202-211,
        # trace.rs:291-302): such a program has a trace but no valid proof upstream, so only trace-level tests turn it on.
        self.const_times_var = const_times_var
        self.consts = set()   # names of variables that are compile-time constants (degree 0)
        self.r = random.Random(seed)
        self.partial = self.r.random() < 0.4 if partial is None else partial
        self.uid = 0
        self.funcs = [_Fn(f"f{i}", self.r.randint(1, 3), self.r.randint(1, 3)) for i in range(n_funcs)]
        self.use_preimg = not self.partial and self.r.random() < 0.5

    def v(self, p="v"):
        self.uid += 1
        return f"{p}{self.uid}"

    def tup(self, names):
        return names[0] if len(names) == 1 else "(" + ", ".join(names) + ")"

    # ---- statements; `sc` = scalar variables in scope, `ptrs` = (pointer var, width) in scope
    def stmts(self, fn, sc, ptrs, lines, ind, depth, budget):
        r = self.r
        pad = "    " * ind
        for _ in range(r.randint(1, budget)):
            k = r.random()
            if k < 0.22:
                a, b, x = r.choice(sc), r.choice(sc), self.v()
                fnn = r.choice(['add', 'sub', 'mul'])
                if fnn == "mul" and not self.const_times_var and (a in self.consts) != (b in self.consts):
                    b = a  # both constants or both variables
                lines.append(f"{pad}let {x} = {fnn}({a}, {b});")
                if a in self.consts and b in self.consts:
                    self.consts.add(x)
                sc.append(x)
            elif k < 0.30:
                x = self.v("c")
                lines.append(f"{pad}let {x} = {r.choice([0, 1, 2, 3, 7, 255, 256, 2013265920, 1006632961])};")
                self.consts.add(x)
                sc.append(x)
            elif k < 0.37:
                a, b, x = r.choice(sc), r.choice(sc), self.v("e")
                if r.random() < 0.6:
                    lines.append(f"{pad}let {x} = eq({a}, {b});")
                    if a in self.consts and b in self.consts:
                        self.consts.add(x)
                else:
                    lines.append(f"{pad}let {x} = not({a});")
                    if a in self.consts:
                        self.consts.add(x)
                sc.append(x)
            elif k < 0.42:
                # a / d with d a non-zero VARIABLE (1 or 2): div expands to inv + mul, and a variable times a constant inverse
                # would be the const x var product above
                e, d, q = self.v("e"), self.v("d"), self.v("q")
                a = r.choice([x for x in sc if x not in self.consts])
                lines.append(f"{pad}let {e} = not({a});")
                lines.append(f"{pad}let {d} = add({e}, {fn['one']});")
                lines.append(f"{pad}let {q} = div({r.choice(sc) if self.const_times_var else a}, {d});")
                sc += [e, d, q]
            elif k < 0.56:
                w = r.choice(MEM_WIDTHS)
                p = self.v("p")
                lines.append(f"{pad}let {p} = store({', '.join(r.choice(sc) for _ in range(w))});")
                ptrs.append((p, w))
                sc.append(p)
            elif k < 0.66 and ptrs:
                p, w = r.choice(ptrs)
                xs = [self.v("l") for _ in range(w)]
                lines.append(f"{pad}let {self.tup(xs)} = load({p});")
                sc += xs
            elif k < 0.80:
                g = r.choice(self.funcs)
                args = [fn["pred"]] + [r.choice(sc) for _ in range(g.n_extra)]
                xs = [self.v("r") for _ in range(g.out)]
                lines.append(f"{pad}let {self.tup(xs)} = call({g.name}, {', '.join(args)});")
                sc += xs
                if r.random() < 0.3:  # the same query again: a memoised hit
                    ys = [self.v("r") for _ in range(g.out)]
                    lines.append(f"{pad}let {self.tup(ys)} = call({g.name}, {', '.join(args)});")
                    lines.append(f"{pad}assert_eq!({xs[0]}, {ys[0]});")
                    sc += ys
            elif k < 0.84 and self.use_preimg:
                a, b = r.choice(sc), r.choice(sc)
                h, s, t = self.v("h"), self.v("i"), self.v("i")
                lines.append(f"{pad}let {h}: [2] = call(pair, {a}, {b});")
                lines.append(f"{pad}let ({s}, {t}) = preimg(pair, {h});")
                sc += [s, t]
            elif k < 0.88:
                a, one, b = r.choice(sc), self.v("c"), self.v()
                lines.append(f"{pad}let {one} = 1;")
                lines.append(f"{pad}let {b} = add({a}, {one});")
                lines.append(f"{pad}assert_ne!({a}, {b});")
                self.consts.add(one)
                if a in self.consts:
                    self.consts.add(b)
                sc += [one, b]
            elif k < 0.91:
                arr, c = self.v("arr"), self.v("c")
                vals = [r.randint(0, 9) for _ in range(r.randint(2, 4))]
                lines.append(f"{pad}let {arr} = [{', '.join(map(str, vals))}];")
                lines.append(f"{pad}let {c} = {r.choice(vals)};")
                lines.append(f"{pad}contains!({arr}, {c});")
                self.consts.add(c)
                sc.append(c)
            elif k < 0.94:
                lines.append(f"{pad}range_u8!({fn['n']}, {fn['pred']});")
            elif depth > 0:
                self.branch(fn, sc, ptrs, lines, ind, depth)

    def ret(self, fn, sc, lines, ind):
        lines.append("    " * ind + "return " + self.tup([self.r.choice(sc) for _ in range(fn["out"])]))

    def block(self, fn, sc, ptrs, lines, ind, depth):
        """A case / if body: statements, then a return (or a nested exhaustive-with-default match)."""
        sc, ptrs = list(sc), list(ptrs)
        self.stmts(fn, sc, ptrs, lines, ind, depth - 1, 3)
        self.ret(fn, sc, lines, ind)

    def branch(self, fn, sc, ptrs, lines, ind, depth):
        r = self.r
        pad = "    " * ind
        k = r.random()
        if k < 0.35:
            x = r.choice(sc)
            lines.append(f"{pad}if {'!' if r.random() < 0.5 else ''}{x} {{")
            self.block(fn, sc, ptrs, lines, ind + 1, depth)
            lines.append(f"{pad}}}")
        elif k < 0.85:
            scrut = r.choice([fn["n"], fn["pred"], r.choice(sc)])
            keys = r.sample(range(0, 7), r.randint(1, 4))
            lines.append(f"{pad}match {scrut} {{")
            while keys:
                take = [keys.pop() for _ in range(min(len(keys), r.choice([1, 1, 2, 3])))]
                lines.append(f"{pad}    {', '.join(map(str, take))} => {{")
                self.block(fn, sc, ptrs, lines, ind + 2, depth)
                lines.append(f"{pad}    }}")
            lines.append(f"{pad}}};")
        else:
            # an array match: the scrutinee is a two-output call's result (arrays are made by literals, calls and loads only)
            g = next((h for h in self.funcs if h.out == 2), None)
            if g is None:
                return
            arr = self.v("arr")
            lines.append(f"{pad}let {arr}: [2] = call({g.name}, {', '.join([fn['pred']] + [r.choice(sc) for _ in range(g.n_extra)])});")
            lines.append(f"{pad}match {arr} {{")
            for key in r.sample([(1, 0), (1, 1), (2, 0), (0, 1), (3, 0), (0, 0)], r.randint(1, 3)):
                lines.append(f"{pad}    [{key[0]}, {key[1]}] => {{")
                self.block(fn, sc, ptrs, lines, ind + 2, depth)
                lines.append(f"{pad}    }}")
            lines.append(f"{pad}}};")

    def func(self, g):
        params = ["n"] + [f"a{i}" for i in range(g.n_extra)]
        head = ("partial " if self.partial else "") + f"fn {g.name}({', '.join(params)}): [{g.out}] {{"
        lines = [head, "    let one = 1;", "    match n {", "        0 => {"]
        fn = {"n": "n", "pred": "n", "out": g.out, "one": "one"}
        self.consts.add("one")
        base_sc = list(params) + ["one"]
        self.ret(fn, base_sc, lines, 3)
        lines += ["        }", "    };", "    let pred = sub(n, one);"]
        fn["pred"] = "pred"
        sc, ptrs = base_sc + ["pred"], []
        self.stmts(fn, sc, ptrs, lines, 1, 2, 7)
        self.ret(fn, sc, lines, 1)
        lines.append("}")
        return "\n".join(lines)

    def source(self):
        out = [self.func(g) for g in self.funcs]
        if self.use_preimg:
            # injective, as an invertible function must be: the inverse map keeps ONE preimage per output, and the reference looks
            # it up again at trace time (/root/reference/src/lair/trace.rs:328-373)
            out.append("invertible fn pair(a, b): [2] {\n    let s = add(a, b);\n    let m = mul(a, a);\n    return (a, s)\n}")
        return "\n".join(out) + "\n"

    def calls(self, depth=5):
        r = self.r
        g = self.funcs[0]
        return [[g.name, [depth] + [r.choice([0, 1, 2, 5, 77, 2013265920]) for _ in range(g.n_extra)]],
                [self.funcs[-1].name, [depth - 1] + [r.randint(0, 300) for _ in range(self.funcs[-1].n_extra)]]]


def program(seed, const_times_var=False):
    g = Gen(seed, const_times_var=const_times_var)
    return g.source(), g.calls(), g.partial
