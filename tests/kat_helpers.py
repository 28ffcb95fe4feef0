"""Shared KAT construction: builds the four known-answer digests of tests/golden/poseidon_kats.json
through lurk_amd.zstore with whichever hasher (oracle or HIP) the test passes in."""
import json
import os

from lurk_amd.field import digest_to_int
from lurk_amd.zstore import ZStore

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "poseidon_kats.json")


def load_kats():
    with open(GOLDEN) as f:
        return json.load(f)


def compute_kats(hasher) -> dict:
    """name -> hex string (no 0x) for each KAT."""
    kats = load_kats()
    z = ZStore(hasher)
    out = {}
    out["hash3_num123"] = z.hash(kats["hash3_num123"]["preimage"])
    # (commit '(13 . 17)): payload = cons(u64 13, u64 17)
    out["commit_cons_13_17"] = z.commit(z.intern_cons(z.u64(13), z.u64(17))).digest
    # (commit (lambda (x) x)): Fun(args=(x), body=(x), env=empty)  (eval_direct.rs:474-482)
    x = z.user_sym("x")
    lst = z.intern_list([x])
    out["commit_lambda_x_x"] = z.commit(z.intern_fun(lst, lst, z.intern_empty_env())).digest
    # proof key = hash3(flatten(expr) || env digest) for (cons 1 2) in the empty env (repl.rs:170-173)
    expr = z.intern_list([z.builtin_sym("cons"), z.u64(1), z.u64(2)])
    out["proof_key_cons_1_2"] = z.hash(expr.flatten() + [0] * 8)
    return {k: format(digest_to_int(v), "x") for k, v in out.items()}
