"""ORACLE (test infrastructure, not product code): the STARK prover stages around the AIRs and the verifier.

PARITY UNPINNED for this file.  Everything here restates algorithms of third-party crates that are absent
from /root/reference -- sphinx-core 1.0.0 @ 8a39b951 (permutation trace, quotient, shard prover/verifier; an
SP1 v1 fork) and Plonky3 @ a0b92870 (BinomialExtensionField<BabyBear,4>, DuplexChallenger, TwoAdicFriPcs, FRI,
FieldMerkleTreeMmcs), Cargo.lock:1626-1852,2528-2592 -- from the published sources, from memory
[UPSTREAM-RECALL]; the reference holds no vectors for any of it (only prove -> verify round trips,
/root/reference/src/lair/lair_chip.rs:246-276).  Call site modelled: machine.prove::<LocalProver>(..) followed by
machine.verify(..), /root/reference/benches/fib.rs:114-160.  What is checked with it: the GPU stages bit for
bit against these restatements, and whole proofs produced by the GPU prover against `verify` below, which
recomputes every transcript challenge, the constraint identity at zeta from the opened values (through the
oracle's own numeric AIR, oracle/air.py) and every FRI query.

Pure Python on small sizes; LDE / Merkle / Poseidon2 come from oracle/liblurkoracle.so.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.
"""
from __future__ import annotations

from . import air as oair
from .lair import P

W = 11  # F[x]/(x^4 - 11)
GEN = 31
ROOT27 = 0x1A427A41


# ------------------------------------------------------------------ base field helpers
def finv(a):
    a %= P
    assert a, "inverse of zero"
    return pow(a, P - 2, P)


def two_adic_generator(bits):
    r = ROOT27
    for _ in range(bits, 27):
        r = r * r % P
    return r


def bitrev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


# ------------------------------------------------------------------ extension field (tuples of 4 ints)
ZERO, ONE = (0, 0, 0, 0), (1, 0, 0, 0)


def ef(x):
    return (x % P, 0, 0, 0)


def ef_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def ef_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def ef_neg(a):
    return tuple((-x) % P for x in a)


def ef_scale(a, s):
    return tuple(x * s % P for x in a)


def ef_mul(a, b):
    t = [0] * 7
    for i in range(4):
        for j in range(4):
            t[i + j] += a[i] * b[j]
    return ((t[0] + W * t[4]) % P, (t[1] + W * t[5]) % P, (t[2] + W * t[6]) % P, t[3] % P)


def ef_inv(a):
    assert any(x % P for x in a), "inverse of zero"
    a1 = (a[0], (-a[1]) % P, a[2], (-a[3]) % P)
    b = ef_mul(a, a1)  # in span{1, x^2}
    b1 = (b[0], 0, (-b[2]) % P, 0)
    n = ef_mul(b, b1)  # in F
    return ef_scale(ef_mul(a1, b1), finv(n[0]))


def ef_pow(a, e):
    r = ONE
    while e:
        if e & 1:
            r = ef_mul(r, a)
        a = ef_mul(a, a)
        e >>= 1
    return r


def ef_sum(xs):
    r = ZERO
    for x in xs:
        r = ef_add(r, x)
    return r


# ------------------------------------------------------------------ LogUp permutation trace
def interactions_of_row(air, main, prep, r, public=()):
    """(multiplicity, tuple, is_send) of every interaction of row r, sends first (sphinx Chip::{sends, receives})."""
    h = len(main)
    n = (r + 1) % h
    b = oair.Builder(main[r], main[n], prep[r] if prep is not None else (), prep[n] if prep is not None else (), public)
    air.eval(b)
    return [(m, v, True) for m, v in b.sends] + [(m, v, False) for m, v in b.receives]


def log_quotient_degree(air, sample_row_width=None):
    """sphinx Chip::new: every Lair chip has degree-3 constraints and interactions -> log2_ceil(3 - 1) = 1."""
    return 1


def fingerprint(alpha, beta, vals, kind=oair.INTERACTION_KIND_MEMORY):
    d = ef_add(alpha, ef(kind))  # alpha + beta^0 * argument_index
    bp = beta
    for v in vals:
        d = ef_add(d, ef_scale(bp, v))
        bp = ef_mul(bp, beta)
    return d


def permutation_trace(air, main, prep, alpha, beta, batch, public=()):
    """sphinx generate_permutation_trace: rows of extension-field tuples; last column = running sum."""
    out, run = [], ZERO
    for r in range(len(main)):
        its = interactions_of_row(air, main, prep, r, public)
        cols = []
        for c in range(0, len(its), batch):
            acc = ZERO
            for m, vals, is_send in its[c:c + batch]:
                mm = m if is_send else (-m) % P
                acc = ef_add(acc, ef_scale(ef_inv(fingerprint(alpha, beta, vals)), mm))
            cols.append(acc)
        run = ef_add(run, ef_sum(cols))
        out.append(cols + [run])
    return out
