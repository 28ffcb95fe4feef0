import json
import os

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "lair_traces.json")


def load_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


# extra programs used by the oracle-vs-GPU tests (not from the reference)
PARTIAL_SRC = """
partial fn pfib(n): [1] {
    let one = 1;
    match n {
        0 => {
            let zero = 0;
            return zero
        }
        1 => {
            return one
        }
    };
    let n_1 = sub(n, one);
    let a = call(pfib, n_1);
    let n_2 = sub(n_1, one);
    let b = call(pfib, n_2);
    let res = add(a, b);
    return res
}
partial fn top(n): [2] {
    let a = call(pfib, n);
    let two = 2;
    let b = call(helper, a, two);
    return (a, b)
}
fn helper(x, y): [1] {
    let p = mul(x, y);
    let q = div(p, y);
    let ptr = store(p, q);
    let (u, v) = load(ptr);
    let r = add(u, v);
    let e = eq(r, p);
    let z = add(e, r);
    return z
}
"""

# a `match` whose default block rebinds the scrutinee's name (the reference's `eval` does, eval_direct.rs:409-423): found by running
# the real evaluator through the oracle in round 5 -- its compiler resolved the scrutinee after compiling the default
SHADOW_SRC = """
fn shadow(t, x): [2] {
    match t {
        3 => {
            return (t, x)
        }
    };
    let (t, x) = call(step, t, x);
    match t {
        5, 6 => {
            let one = 1;
            let x = add(x, one);
            return (t, x)
        }
    };
    return (x, t)
}
fn step(t, x): [2] {
    let one = 1;
    let t = add(t, one);
    let x = mul(x, t);
    return (t, x)
}
"""
SHADOW_CALLS = [(("shadow", [3, 7]), [3, 7]), (("shadow", [4, 2]), [5, 11]), (("shadow", [1, 2]), [4, 2]), (("shadow", [5, 3]), [6, 19])]

U64_SRC = """
fn u64_ops(a: [8], b: [8]): [18] {
    let s: [8] = extern_call(u64_add, a, b);
    let d: [8] = extern_call(u64_sub, a, b);
    let lt = extern_call(u64_lessthan, a, b);
    let z = extern_call(u64_iszero, d);
    range_u8!(s, d);
    return (s, d, lt, z)
}
invertible fn hash3(preimg: [24]): [8] {
    let img: [8] = extern_call(hasher3, preimg);
    return img
}
invertible fn hash4(preimg: [32]): [8] {
    let img: [8] = extern_call(hasher4, preimg);
    return img
}
invertible fn hash5(preimg: [40]): [8] {
    let img: [8] = extern_call(hasher5, preimg);
    return img
}
fn u64_more(a: [8], b: [8]): [24] {
    let p: [8] = extern_call(u64_mul, a, b);
    let (q: [8], r: [8]) = extern_call(u64_divrem, a, b);
    return (p, q, r)
}
fn big_lt(a: [8], b: [8]): [1] {
    let lt = extern_call(big_num_lessthan, a, b);
    return lt
}
fn chain(x: [8]): [8] {
    let z = [0; 8];
    let one = [1; 8];
    let h: [8] = call(hash3, z, one, x);
    let (a: [8], b: [8], c: [8]) = preimg(hash3, h);
    let g: [8] = call(hash4, a, b, c, h);
    return g
}
"""
