"""Commitment digests the reference's own REPL scripts hold, reproduced by the reference's evaluator on the product's host side
(build container only: the scripts and the evaluator's functions are read from /root/reference at run time, nothing is stored).

The reference runs `src/core/cli/tests/{first,second}.lurk` and the demos in its test suite (/root/reference/src/core/cli/tests/mod.rs:11-55);
the literals `#0x..` / `#c0x..` those scripts pass to `!(call ..)`, `!(chain ..)` and `!(open ..)` are digests an earlier
`!(commit x)` / `!(hide s x)` of the same session printed (/root/reference/src/core/cli/meta.rs:403-470: `commit` = `hide` with a zero
secret; the commitment is hash3(secret || tag, 0^7 || digest of the reduced payload)), or the head of the chain a `!(chain ..)` moved to
(meta.rs:546-587).  Each is recomputed here by evaluating `(commit x)` / `(hide s x)` / `(cdr (comm args))` with the session's
definitions folded around it (tools/lurk_reference.py: fold_repl_script) -- through the real `eval` functions, the product's compiler
and interpreter, the native hash3 / hash4 / hash5 chips and the ZStore mirror that interns the program.  Pins P2/P3 (Poseidon2 widths
24 / 32 / 40 through closures, environments, big nums) and T4 (the interpreter on letrec, closures, commitments opened by application)
beyond the four known answers of tests/golden/poseidon_kats.json."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import lurk_reference as lr  # noqa: E402

pytestmark = pytest.mark.skipif(not lr.available(), reason="/root/reference is not on this box")
P = 2013265921


@pytest.fixture(scope="module")
def real():
    import measure_lurk_shape as ms

    return ms.RealLurk()


def digest_of(literal: str):
    v = int(literal.split("0x", 1)[1], 16)
    out = []
    for _ in range(8):
        out.append(v % P)
        v //= P
    assert v == 0
    return out


def literals(text: str):
    """the digest literals of a script in order of first appearance (comments included: protocol.lurk quotes one there)"""
    seen = []
    for m in re.finditer(r"#c?0x[0-9a-f]+", text):
        d = m.group(0).replace("#c0x", "#0x")
        if d not in seen:
            seen.append(d)
    return seen


def hexkey(digest):
    return format(sum(int(d) * P**i for i, d in enumerate(digest)), "x")


def proof_key(real, expr_text: str, env_digest):
    """`prove_last_reduction`'s key (/root/reference/src/core/cli/repl.rs:164-173): hash3 of the first 24 public values = the flattened
    expression and the environment's digest, printed like a big num."""
    import measure_lurk_shape as ms
    from lurk_amd import zstore as zs

    z = zs.ZStore(real.hasher)
    zp = ms.intern_syntax(z, lr.read_lurk(expr_text))
    return hexkey(real.hasher.hash(list(zp.flatten()) + list(env_digest)))


def definitions(forms, upto, real=None, keys=None):
    """The session's definitions before form `upto`, after the last `!(clear)`: what `fold_repl_script` can wrap around an expression.
    A `!(defq name !(prove))` binds the proof key of the session's last reduction -- a `!(call c args..)` (the call expression under
    the session's environment, /root/reference/src/core/cli/meta.rs:529-545) or a `!(chain c args..)` (under the empty one,
    meta.rs:564-587), arguments evaluated then quoted (meta.rs:484-497: the scripts' arguments are literals) -- which is part of every
    closure defined afterwards: its environment holds the string.  `keys` collects the keys computed on the way."""
    keep, last = [], None
    for head, args, text in forms[:upto]:
        if head == "clear":
            keep, last = [], None
        elif head in ("def", "defrec") or (head == "defq" and "!(" not in args[1]):
            keep.append(text)
        elif head in ("call", "chain"):
            quoted = " ".join(a if a.startswith("'") else "'" + a for a in args[1:])
            last = (f"({args[0]} {quoted})", head == "call", "\n".join(keep))
        elif real is not None and last is not None and (head == "prove" or (head == "defq" and args[1].replace(" ", "") == "!(prove)")):
            expr, session_env, defs = last
            env_digest = [0] * 8
            if session_env and defs:
                tag, env_digest = evaluate(real, defs, "(current-env)")
                assert tag == lr.enums()["Tag"]["Env"]
            key = proof_key(real, expr, env_digest)
            if keys is not None:
                keys.append(key)
            if head == "defq":
                keep.append(f'!(def {args[0]} "{key}")')
    return "\n".join(keep)


def evaluate(real, defs: str, expr: str):
    out, _, _ = real.run(lr.fold_repl_script(defs, tail=expr))
    return int(out[0]), [int(x) for x in out[8:16]]


def commit_forms(forms):
    return [(i, head, args) for i, (head, args, _) in enumerate(forms) if head in ("commit", "hide")]


def as_expr(head, args):
    return "(commit " + args[0] + ")" if head == "commit" else "(hide " + args[0] + " " + args[1] + ")"


COMM = None


def comm_tag():
    global COMM
    if COMM is None:
        COMM = lr.enums()["Tag"]["Comm"]
    return COMM


def test_functional_commitment_demo(real):
    text = lr.demo_script("functional-commitment.lurk")
    forms = lr.repl_forms(text)
    (i, head, args), = commit_forms(forms)
    tag, digest = evaluate(real, definitions(forms, i), as_expr(head, args))
    assert tag == comm_tag() and digest == digest_of(literals(text)[0])
    # `!(call #0x.. 5)` under the session's environment (f is bound), then `!(prove)`: the key `!(verify "..")` names
    keys = []
    definitions(forms, len(forms), real, keys)
    assert keys == [a[0].strip('"') for h, a, _ in forms if h == "verify"] and len(keys) == 1


def test_chained_functional_commitment_demo(real):
    """`!(commit f0)` is the first literal; `!(chain c k)` applies the committed function and moves the head to the commitment in the
    result's cdr: the second and third literals."""
    text = lr.demo_script("chained-functional-commitment.lurk")
    forms = lr.repl_forms(text)
    lits = literals(text)
    (i, head, args), = commit_forms(forms)
    c0 = as_expr(head, args)
    tag, digest = evaluate(real, "", c0)
    assert tag == comm_tag() and digest == digest_of(lits[0])
    chains = [a for h, a, _ in forms if h == "chain"]
    assert [a[0].replace("#c0x", "#0x") for a in chains] == lits[:3]
    state = c0
    for k, a in enumerate(chains[:2]):
        state = f"(cdr ({state} {' '.join(a[1:])}))"
        tag, digest = evaluate(real, "", state)
        assert tag == comm_tag() and digest == digest_of(lits[k + 1]), k
    # the counter the last chain reaches (the script's own comment: 21 + 14 = 35)
    out, _, _ = real.run(f"(car ({state} {' '.join(chains[2][1:])}))")
    assert int(out[8]) == 35
    # the proof keys the script verifies: one per chain transition, the call expression under the empty environment
    keys = []
    definitions(forms, len(forms), real, keys)
    assert keys == [a[0].strip('"') for h, a, _ in forms if h == "verify"] and len(keys) == 3


def test_bank_demo(real):
    """demo/bank.lurk: the committed transfer function, the committed chain over the ledger (hidden behind the secret #0x999), and
    the two heads its transfers move to."""
    text = lr.demo_script("bank.lurk")
    forms = lr.repl_forms(text)
    lits = literals(text)
    commits = commit_forms(forms)
    assert len(commits) == 2
    calls = [(h, a) for h, a, _ in forms if h in ("call", "chain")]
    # literal order in the file: the call's, then the three chain heads (the secret #0x999 is a literal too: not a digest we produce)
    produced = [l for l in lits if l != "#0x999"]
    (i0, h0, a0), (i1, h1, a1) = commits
    tag, digest = evaluate(real, definitions(forms, i0), as_expr(h0, a0))
    assert tag == comm_tag() and digest == digest_of(produced[0])
    assert calls[0][1][0].replace("#c0x", "#0x") == produced[0]
    defs = definitions(forms, i1, real)  # (the session has bound `proof-key` by now: the chained function's closure holds it)
    state = as_expr(h1, a1)
    tag, digest = evaluate(real, defs, state)
    assert tag == comm_tag() and digest == digest_of(produced[1])
    chains = [a for h, a in calls if h == "chain"]
    assert [a[0].replace("#c0x", "#0x") for a in chains] == produced[1:4]
    for k, a in enumerate(chains[:2]):
        state = f"(cdr ({state} {' '.join(a[1:])}))"
        tag, digest = evaluate(real, defs, state)
        assert tag == comm_tag() and digest == digest_of(produced[k + 2]), k


def test_cli_test_scripts(real):
    """src/core/cli/tests/first.lurk commits; second.lurk (a later session of the same test) opens and calls what it printed."""
    first = lr._read("src/core/cli/tests/first.lurk")
    second = lr._read("src/core/cli/tests/second.lurk")
    forms = lr.repl_forms(first)
    commits = commit_forms(forms)
    l1, l2 = literals(first), literals(second)
    # in first.lurk's order: hide .. 42, commit 42, commit (lambda (x) x), commit (letrec add), [clear] commit (letrec add, big nums)
    assert len(commits) == 5
    want = [l2[0], l2[1], l1[0], l1[1], l1[2]]
    assert l2[2:] == [l1[0], l1[1]]  # (second.lurk calls and chains the two functional commitments again)
    for (i, head, args), lit in zip(commits, want):
        tag, digest = evaluate(real, definitions(forms, i), as_expr(head, args))
        assert tag == comm_tag() and digest == digest_of(lit), (head, args, lit)
    # ... and what second.lurk asserts of them: both open to 42; the chain's first transition gives 1
    for (i, head, args) in commits[:2]:
        out, _, _ = real.run(f"(open {as_expr(head, args)})")
        assert int(out[8]) == 42
    (i, head, args) = commits[3]
    out, _, _ = real.run(f"(car ({as_expr(head, args)} 1))")
    assert int(out[8]) == 1


def test_protocol_demo_commitment(real):
    text = lr.demo_script("protocol.lurk")
    lit = literals(text)[0]
    tag, digest = evaluate(real, "", "(commit '(13 . 17))")
    assert tag == comm_tag() and digest == digest_of(lit)
