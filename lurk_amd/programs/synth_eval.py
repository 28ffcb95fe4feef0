"""Synthetic stand-in for the Lurk `eval` chip used by bench.py and the full-size tests.

The reference's evaluator (/root/reference/src/core/eval_direct.rs, 39 Lair functions) is host-side
program text and out of scope for this build; what the proving hot path sees of it is
a chip of width 78 (`eval`, eval_direct.rs:2028) whose rows mix calls, memory loads/stores, tag matches
and inequality witnesses.  `synth_eval` below is a Lair function with exactly that width (78 columns)
and the same kinds of rows, recursive in its second argument so that `synth_eval(1, n, 0)` produces n + 1
distinct queries: one trace row each.
"""

SOURCE = """
fn synth_eval(tag, x, env): [2] {
    let zero = 0;
    let one = 1;
    if !x {
        return (zero, env)
    }
    let nx = sub(x, one);
    match tag {
        1 => {
            // cons-like: allocate cells, evaluate the tail, rebuild
            let p = store(tag, x, env);
            let t = 2;
            let (rt, r) = call(synth_eval, t, nx, p);
            let (a, b, c) = load(p);
            let s = add(r, a);
            let m = mul(s, b);
            let e = eq(m, c);
            let o = add(e, m);
            let q = store(o, rt, a, b);
            let (u, v, w, z) = load(q);
            let k = mul(u, v);
            let l = mul(k, w);
            let (at, av) = call(aux, l, z);
            let f = store(at, av);
            let (g, h) = load(f);
            let y = mul(g, h);
            let res = add(y, o);
            let (bt, bv) = call(aux, res, y);
            let cell5 = store(bt, bv, g, h, y);
            let (c1, c2, c3, c4, c5) = load(cell5);
            let c12 = mul(c1, c2);
            let c34 = add(c3, c4);
            let c345 = add(c34, c5);
            let fin = add(c12, c345);
            return (rt, fin)
        }
        2 => {
            // lookup-like: walk the environment cell, compare symbols
            let (a, b, c) = load(env);
            let t = 3;
            let (rt, r) = call(synth_eval, t, nx, c);
            let q = store(r, a, b, rt);
            let (u, v, w, z) = load(q);
            let m = mul(u, v);
            let k = mul(m, w);
            let e1 = eq(a, x);
            let e2 = eq(b, r);
            let both = mul(e1, e2);
            let (st, sv) = call(aux, k, both);
            let o = add(sv, z);
            let o2 = mul(o, st);
            return (t, o2)
        }
        3 => {
            // arithmetic-like: dependent calls, a division, comparisons
            let t = 1;
            let (rt, r) = call(synth_eval, t, nx, env);
            let five = 5;
            let d = add(r, five);
            let n = not(d);
            let g = add(d, n);
            let i = div(x, g);
            let (st, s) = call(aux, i, rt);
            let o = mul(s, st);
            let (st2, s2) = call(aux, o, s);
            let ne = eq(st2, s2);
            let h = add(ne, o);
            let j = mul(h, i);
            let cell = store(j, h);
            let (ca, cb) = load(cell);
            let fin = mul(ca, cb);
            return (rt, fin)
        }
    };
    let t = 1;
    let (rt, r) = call(synth_eval, t, nx, env);
    return (rt, r)
}
fn aux(a, b): [2] {
    let c = mul(a, b);
    let d = add(c, a);
    return (d, c)
}
"""

FUNC = "synth_eval"
WIDTH = 78


def args_for_rows(n_rows: int):
    """Arguments that make `synth_eval` produce exactly n_rows queries."""
    return [1, n_rows - 1, 0]
