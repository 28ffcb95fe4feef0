#!/usr/bin/env python3
"""Writes tests/golden/lair_traces.json: the literal golden matrices of the reference's own Lair tests,
transcribed by hand (data only).  Each case names the reference test it comes from, the function source in
the `func!` surface syntax (test input), the calls made, and the expected row-major u32 trace.
Run: python tests/golden/make_lair_traces.py
"""
import json
import os

DEMO = """
fn factorial(n): [1] {
    let one = 1;
    if n {
        let pred = sub(n, one);
        let m = call(factorial, pred);
        let res = mul(n, m);
        return res
    }
    return one
}
fn fib(n): [1] {
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
    let a = call(fib, n_1);
    let n_2 = sub(n_1, one);
    let b = call(fib, n_2);
    let res = add(a, b);
    return res
}
fn even(n): [1] {
    let one = 1;
    match n {
        0 => {
            return one
        }
    };
    let pred = sub(n, one);
    let res = call(odd, pred);
    return res
}
fn odd(n): [1] {
    let one = 1;
    match n {
        0 => {
            let zero = 0;
            return zero
        }
    };
    let pred = sub(n, one);
    let res = call(even, pred);
    return res
}
"""

CASES = []


def case(name, source, src_ref, func, calls, width, trace, layout=None, lurk_chips=False, mem=None):
    assert len(trace) % width == 0, (name, len(trace), width)
    CASES.append(dict(name=name, reference=src_ref, source=source, lurk_chips=lurk_chips, func=func, calls=calls,
                      width=width, trace=trace, layout=layout, mem=mem))


# src/lair/trace.rs:461-514 (lair_trace_test) + layout src/lair/trace.rs:445-459
case("factorial_5", DEMO, "src/lair/trace.rs:466-488", "factorial", [["factorial", [5]]], 13, [
    0, 5, 120, 0, 1, 1610612737, 24, 0, 0, 1, 120, 0, 1,
    1, 4, 24, 0, 1, 1509949441, 6, 0, 0, 1, 24, 0, 1,
    2, 3, 6, 1, 1, 1342177281, 2, 0, 0, 1, 6, 0, 1,
    3, 2, 2, 2, 1, 1006632961, 1, 0, 0, 1, 2, 0, 1,
    4, 1, 1, 3, 1, 1, 1, 0, 0, 1, 1, 0, 1,
    5, 0, 1, 4, 1, 0, 0, 0, 0, 0, 0, 1, 0,
    6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
], layout=dict(nonce=1, input=1, aux=8, sel=2, output=1))

case("fib_7", DEMO, "src/lair/trace.rs:490-513", "fib", [["fib", [7]]], 18, [
    0, 7, 13, 0, 1, 862828252, 1677721601, 8, 0, 0, 1, 5, 1, 1, 1006632961, 0, 0, 1,
    1, 6, 8, 0, 1, 1677721601, 1610612737, 5, 0, 0, 1, 3, 2, 1, 1006632961, 0, 0, 1,
    2, 5, 5, 0, 2, 1610612737, 1509949441, 3, 0, 0, 1, 2, 3, 1, 1006632961, 0, 0, 1,
    3, 4, 3, 1, 2, 1509949441, 1342177281, 2, 0, 0, 1, 1, 4, 1, 1006632961, 0, 0, 1,
    4, 3, 2, 2, 2, 1342177281, 1006632961, 1, 0, 0, 1, 1, 5, 1, 1006632961, 0, 0, 1,
    5, 2, 1, 3, 2, 1006632961, 1, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1,
    6, 1, 1, 4, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0,
    7, 0, 0, 5, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0,
])

MATCH = """
fn test(n, m): [1] {
    let one = 1;
    match n {
        0 => {
            return one
        }
        1 => {
            return m
        }
        2 => {
            let res = mul(m, m);
            return res
        }
        3 => {
            let res = mul(m, m);
            let res = mul(res, res);
            return res
        }
    };
    let pred = sub(n, one);
    let res = call(test, pred, m);
    return res
}
"""
case("match_5_2", MATCH, "src/lair/trace.rs:517-576", "test", [["test", [5, 2]]], 19, [
    0, 5, 2, 16, 0, 1, 1610612737, 1509949441, 1342177281, 1006632961, 16, 0, 0, 1, 0, 0, 0, 0, 1,
    1, 4, 2, 16, 0, 1, 1509949441, 1342177281, 1006632961, 1, 16, 0, 0, 1, 0, 0, 0, 0, 1,
    2, 3, 2, 16, 1, 1, 4, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0,
    3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
], layout=dict(nonce=1, input=2, aux=10, sel=5, output=1))

INNER = """
fn test(n, m): [1] {
    let zero = 0;
    let one = 1;
    let two = 2;
    let three = 3;
    match n {
        0 => {
            match m {
                0 => {
                    return zero
                }
                1 => {
                    return one
                }
            }
        }
        1 => {
            match m {
                0 => {
                    return two
                }
                1 => {
                    return three
                }
            }
        }
    }
}
"""
case("inner_match", INNER, "src/lair/trace.rs:579-652", "test",
     [["test", [0, 0]], ["test", [0, 1]], ["test", [1, 0]], ["test", [1, 1]]], 10, [
         0, 0, 0, 0, 0, 1, 1, 0, 0, 0,
         1, 0, 1, 1, 0, 1, 0, 1, 0, 0,
         2, 1, 0, 2, 0, 1, 0, 0, 1, 0,
         3, 1, 1, 3, 0, 1, 0, 0, 0, 1,
     ], layout=dict(nonce=1, input=2, aux=2, sel=4, output=1))

NOT_EQ = """
fn eq(a, b): [1] {
    let x = eq(a, b);
    return x
}
fn not(a): [1] {
    let x = not(a);
    return x
}
"""
case("not", NOT_EQ, "src/lair/air.rs:625-674", "not", [["not", [4]], ["not", [8]], ["not", [0]], ["not", [1]]], 8, [
    0, 4, 0, 0, 1, 1509949441, 0, 1,
    1, 8, 0, 0, 1, 1761607681, 0, 1,
    2, 0, 1, 0, 1, 0, 1, 1,
    3, 1, 0, 0, 1, 1, 0, 1,
])
case("eq", NOT_EQ, "src/lair/air.rs:676-709", "eq", [["eq", [4, 2]], ["eq", [4, 4]], ["eq", [0, 3]], ["eq", [0, 0]]], 9, [
    0, 4, 2, 0, 0, 1, 1006632961, 0, 1,
    1, 4, 4, 1, 0, 1, 0, 1, 1,
    2, 0, 3, 0, 0, 1, 671088640, 0, 1,
    3, 0, 0, 1, 0, 1, 0, 1, 1,
])

IF_MANY = """
fn if_many(a: [4]): [1] {
    if a {
        let one = 1;
        return one
    }
    let zero = 0;
    return zero
}
"""
case("if_many", IF_MANY, "src/lair/air.rs:715-768", "if_many",
     [["if_many", [0, 0, 0, 0]], ["if_many", [1, 3, 8, 2]], ["if_many", [0, 0, 4, 1]], ["if_many", [0, 0, 0, 9]]], 14, [
         0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0,
         1, 1, 3, 8, 2, 1, 0, 1, 1, 0, 0, 0, 0, 1,
         2, 0, 0, 4, 1, 1, 0, 1, 0, 0, 1509949441, 0, 0, 1,
         3, 0, 0, 0, 9, 1, 0, 1, 0, 0, 0, 447392427, 0, 1,
     ])

MATCH_MANY = """
fn match_many(a: [2]): [2] {
    match a {
        [0, 0] => {
            let res = [1, 0];
            return res
        }
        [0, 1] => {
            let res = [1, 1];
            return res
        }
        [1, 0] => {
            let res = [1, 2];
            return res
        }
        [1, 1] => {
            let res = [1, 3];
            return res
        }
    };
    let fail = [0, 0];
    return fail
}
"""
case("match_many", MATCH_MANY, "src/lair/air.rs:770-846", "match_many",
     [["match_many", [0, 0]], ["match_many", [0, 1]], ["match_many", [1, 0]], ["match_many", [1, 1]], ["match_many", [0, 8]]], 20, [
         0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0,
         1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0,
         2, 1, 0, 1, 2, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0,
         3, 1, 1, 1, 3, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0,
         4, 0, 8, 0, 0, 0, 1, 0, 1761607681, 0, 862828252, 2013265920, 0, 2013265920, 0, 0, 0, 0, 0, 1,
         5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
         6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
         7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
     ])

ASSERT = """
fn assert(a: [4]): [4] {
    let arr1 = [2, 4, 5, 8];
    let arr2 = [2, 4, 6, 8];
    assert_ne!(a, arr1);
    let two = 2;
    let four = 4;
    contains!(a, two);
    contains!(a, four);
    assert_eq!(a, arr2);
    return a
}
"""
case("assert", ASSERT, "src/lair/air.rs:848-885", "assert", [["assert", [2, 4, 6, 8]]], 22, [
    0, 2, 4, 6, 8, 2, 4, 6, 8, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1,
])

EQUAL_BRANCH = """
fn test(a): [1] {
    match a {
        2, 3, 4 => {
            let one = 1;
            return one
        }
    };
    return a
}
"""
case("equal_branch", EQUAL_BRANCH, "src/lair/air.rs:887-940", "test",
     [["test", [1]], ["test", [2]], ["test", [3]], ["test", [4]]], 10, [
         0, 1, 1, 0, 1, 2013265920, 1006632960, 671088640, 0, 1,
         1, 2, 1, 0, 1, 0, 0, 0, 1, 0,
         2, 3, 1, 0, 1, 0, 0, 0, 1, 0,
         3, 4, 1, 0, 1, 2, 0, 0, 1, 0,
     ])

RANGE = """
fn range_test(x: [3]): [0] {
    range_u8!(x);
    return ()
}
"""
case("range", RANGE, "src/lair/air.rs:942-966", "range_test", [["range_test", [100, 12, 64]]], 13, [
    0, 100, 12, 64, 0, 1, 0, 0, 1, 0, 0, 1, 1,
])

MEMORY = """
fn test(): [2] {
    let one = 1;
    let two = 2;
    let three = 3;
    let ptr1 = store(one, two, three);
    let ptr2 = store(one, one, one);
    let (_x, y, _z) = load(ptr1);
    return (ptr2, y)
}
"""
case("memory", MEMORY, "src/lair/memory.rs:131-177", "test", [["test", []]], 20, [
    0, 2, 2, 0, 1, 1, 0, 0, 1, 2, 0, 0, 1, 1, 2, 3, 0, 1, 1006632961, 1,
], mem=dict(len=3, width=7, trace=[
    1, 1, 0, 2, 1, 2, 3,
    1, 2, 0, 1, 1, 1, 1,
    0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0,
]))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lair_traces.json")
with open(out, "w") as f:
    json.dump({"_comment": "golden Lair traces transcribed from the reference's tests (data only); see make_lair_traces.py", "cases": CASES}, f, indent=1)
print(out, len(CASES), "cases")
