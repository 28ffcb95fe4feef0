"""Level-order batched ZStore on the device (lurk_amd/csrc/zstore.cpp through the C ABI):
  * the reference's known-answer digests (tests/golden/poseidon_kats.json: zstore.rs:1008-1019, prove.lurk:12-16,
    eval_direct.rs:466-470, verify.lurk:1) come out of the BATCHED path;
  * a forest of syntax trees interned in one pass equals node-by-node interning (zstore.rs:305-349), with one launch per DAG
    height and width and exactly one permutation per distinct preimage;
  * memoize_dag (zstore.rs:569-702) rebuilds the DAG from inverse hash tables: the reference's own test, zstore.rs:934-981;
  * the ZDag export (cli/zdag.rs:16-55) lists every reachable node once, children first."""
import random

import pytest

from kat_helpers import load_kats
from lurk_amd import zstore as zs
from lurk_amd.field import digest_to_int
from lurk_amd.poseidon import Hasher
from lurk_amd.zstore import TAG, BatchedZStore, ZStore

pytestmark = pytest.mark.gpu


def test_known_answers_through_the_batched_path(ctx):
    kats = load_kats()
    st = BatchedZStore(ctx)
    b = st.batch()
    # (commit '(13 . 17)), (commit (lambda (x) x)), the proof key of (cons 1 2) in the empty env, in ONE pass
    c1 = b.commit(b.cons(b.u64(13), b.u64(17)))
    lst = b.list([b.symbol((zs.USER_PACKAGE, "x"))])
    c2 = b.commit(b.fun(lst, lst, b.empty_env()))
    expr = b.syntax(zs.syn_list(zs.syn_builtin("cons"), zs.syn_u64(1), zs.syn_u64(2)))
    b.run()
    assert format(digest_to_int(b[c1].digest), "x") == kats["commit_cons_13_17"]["digest_hex"]
    assert format(digest_to_int(b[c2].digest), "x") == kats["commit_lambda_x_x"]["digest_hex"]
    # the proof key hash3(flatten(expr) | env digest) is not a Lurk datum: plain hasher over the batch-interned expr
    ref = ZStore(Hasher(ctx))
    assert format(digest_to_int(ref.hash(b[expr].flatten() + [0] * 8)), "x") == kats["proof_key_cons_1_2"]["digest_hex"]
    assert b[expr] == ref.intern_list([ref.builtin_sym("cons"), ref.u64(1), ref.u64(2)])
    st.close()


def random_syntax(rng, depth):
    if depth == 0 or rng.random() < 0.25:
        k = rng.randrange(6)
        if k == 0:
            return zs.syn_num(rng.randrange(1 << 30))
        if k == 1:
            return zs.syn_u64(rng.randrange(1 << 40))
        if k == 2:
            return zs.syn_char(rng.choice("abcxyz"))
        if k == 3:
            return zs.syn_str(rng.choice(["", "hi", "lurk", "hello world"]))
        if k == 4:
            return zs.syn_user(rng.choice(["x", "y", "foo"]))
        return zs.syn_builtin(rng.choice(["cons", "car", "lambda"]))
    k = rng.randrange(3)
    xs = [random_syntax(rng, depth - 1) for _ in range(rng.randrange(1, 4))]
    if k == 0:
        return zs.syn_list(*xs)
    if k == 1:
        return zs.syn_improper(xs, random_syntax(rng, depth - 1))
    return zs.syn_quote(xs[0])


def sequential(z: ZStore, syn):
    k = syn[0]
    if k == "num":
        return z.num(syn[1])
    if k == "u64":
        return z.u64(syn[1])
    if k == "char":
        return z.char(syn[1])
    if k == "str":
        return z.intern_string(syn[1])
    if k == "sym":
        return z.intern_symbol(list(syn[1]), builtin=syn[2] == "builtin")
    if k == "list":
        return z.intern_list([sequential(z, x) for x in syn[1]])
    if k == "improper":
        return z.intern_list([sequential(z, x) for x in syn[1]], sequential(z, syn[2]))
    quote = z.intern_symbol([zs.LURK_PACKAGE, zs.BUILTIN_PACKAGE, "quote"], builtin=True)
    return z.intern_list([quote, sequential(z, syn[1])])


def test_forest_equals_node_by_node_interning(ctx):
    rng = random.Random(11)
    forest = [random_syntax(rng, 5) for _ in range(40)]
    st = BatchedZStore(ctx)
    before = st.stats()
    got = st.intern_syntax_many(forest)
    after = st.stats()
    ref = ZStore(Hasher(ctx))
    assert got == [sequential(ref, s) for s in forest]
    # one permutation per distinct preimage (the sequential store's memo holds exactly those), few launches
    assert after["hash4"] == len([k for k in ref.hashes if len(k) == 32])
    launches = after["launches"] - before["launches"]
    assert launches <= 60, launches            # DAG height (strings of <= 11 chars + list spines), not the ~2000 nodes
    assert after["hash4"] > 10 * launches
    # a second pass over the same forest hashes nothing
    st.intern_syntax_many(forest)
    assert st.stats()["launches"] == after["launches"]
    st.close()


def test_memoize_dag_like_the_reference(ctx):
    """zstore.rs:934-981 without the evaluator: the datum ("hi" . <closure (x) -> x>) is built in one store, its hash tables are
    inverted (what `record.get_inv_queries("hash4" / "hash5")` gives the reference), and a FRESH store recovers the DAG from the
    tag and digest alone."""
    src = ZStore(Hasher(ctx))
    x = src.user_sym("x")
    list_x = src.intern_list([x])
    fun = src.intern_fun(list_x, list_x, src.intern_empty_env())
    datum = src.intern_cons(src.intern_string("hi"), fun)
    inv4 = {d: k for k, d in src.hashes.items() if len(k) == 32}
    inv5 = {d: k for k, d in src.hashes.items() if len(k) == 40}
    st = BatchedZStore(ctx)
    st.set_inverse_tables(inv4, inv5)
    with pytest.raises(KeyError):
        st.fetch_tuple11(datum)
    st.memoize_dag(datum.tag, datum.digest)
    car, cdr = st.fetch_tuple11(datum)
    args, body, env = st.fetch_tuple110(cdr)
    assert car == src.intern_string("hi")
    assert args == list_x and body == list_x and env == src.intern_empty_env()
    # the string was unrolled down to the null string, heads tagged Char
    h, tail = st.fetch_tuple11(car)
    assert h == src.char("h") and st.fetch_tuple11(tail)[1] == src.null(TAG["Str"])
    # a digest without a preimage is the reference's `expect("Hash4 preimg not found")`
    with pytest.raises(RuntimeError, match="preimg not found"):
        st.memoize_dag(TAG["Cons"], (1, 2, 3, 4, 5, 6, 7, 8))
    st.close()


def test_dag_export_is_children_first_and_complete(ctx):
    st = BatchedZStore(ctx)
    b = st.batch()
    lst = b.list([b.symbol((zs.USER_PACKAGE, "x"))])
    fun = b.fun(lst, lst, b.empty_env())
    root = b.cons(b.string("hi"), fun)
    b.run()
    entries = st.dag_export([b[root], b[fun]])
    seen = set()
    for z, kind, kids in entries:
        assert z not in seen
        assert all(k in seen for k in kids), "children first"
        assert len(kids) == {0: 0, 1: 2, 2: 3}[kind]
        seen.add(z)
    assert b[root] in seen and b[fun] in seen and st.nil in seen
    # every node of the batch that is reachable from the root is there: "hi" (2 chars + null), x symbol chain, nil, env, ...
    assert ZStore.char("h") in seen and ZStore.null(TAG["Env"]) in seen
    with pytest.raises(KeyError):
        st.dag_export([ZStore.num(77)])  # never interned
    st.close()
