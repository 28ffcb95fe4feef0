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


# ------------------------------------------------------------------ domains, LDE (pure Python, small sizes)
def ntt(vals, root):
    """Evaluations of the polynomial with coefficient list `vals` at root^0 .. root^(n-1) (natural order)."""
    n = len(vals)
    if n == 1:
        return list(vals)
    even = ntt(vals[0::2], root * root % P)
    odd = ntt(vals[1::2], root * root % P)
    out = [0] * n
    w = 1
    for i in range(n // 2):
        t = w * odd[i] % P
        out[i] = (even[i] + t) % P
        out[i + n // 2] = (even[i] - t) % P
        w = w * root % P
    return out


def interpolate(evals, log_n):
    """Coefficients of the polynomial with the given evaluations over H = <w_N> (natural order)."""
    n = 1 << log_n
    root_inv = finv(two_adic_generator(log_n))
    n_inv = finv(n)
    return [c * n_inv % P for c in ntt(list(evals), root_inv)]


def coset_lde_column(evals, log_n, log_blowup, shift=GEN):
    """p3 coset_lde_batch(.., shift) on one column: evaluations on shift * <w_{N << b}>, NATURAL order."""
    coef = interpolate(evals, log_n)
    m = 1 << (log_n + log_blowup)
    sp, scaled = 1, []
    for c in coef:
        scaled.append(c * sp % P)
        sp = sp * shift % P
    return ntt(scaled + [0] * (m - len(scaled)), two_adic_generator(log_n + log_blowup))


def coset_lde(rows, log_blowup=1, shift=GEN):
    """Row-major matrix -> LDE rows in natural order."""
    n = len(rows)
    log_n = n.bit_length() - 1
    cols = [coset_lde_column([r[c] for r in rows], log_n, log_blowup, shift) for c in range(len(rows[0]))]
    return [[col[i] for col in cols] for i in range(n << log_blowup)]


def bit_reverse_rows(rows):
    bits = len(rows).bit_length() - 1
    return [rows[bitrev(i, bits)] for i in range(len(rows))]


def flatten_ef_rows(rows):
    return [[c for e in r for c in e] for r in rows]


def selectors_at(x, log_n):
    """(is_first_row, is_last_row, is_transition, inv_zeroifier) of the trace domain H of size 2^log_n at a point x
    outside H (p3 TwoAdicMultiplicativeCoset::selectors_on_coset / selectors_at_point: unnormalised)."""
    n = 1 << log_n
    zh = (pow(x, n, P) - 1) % P
    w_inv = finv(two_adic_generator(log_n))
    return (zh * finv(x - 1) % P, zh * finv(x - w_inv) % P, (x - w_inv) % P, finv(zh))


# ------------------------------------------------------------------ quotient
def fold_constraints(b: "oair.Builder", perm_local, perm_next, perm_alpha, perm_beta, batch, alpha, cumulative_sum, sels):
    """sphinx Chip::eval with the folding builder: the chip's constraints, then eval_permutation_constraints;
    accumulator = accumulator * alpha + constraint.  perm rows are lists of EF tuples.  `sels` may hold base or
    extension values (the verifier evaluates at an extension point); everything is lifted to EF."""

    def lift(v):
        return v if isinstance(v, tuple) else ef(v)

    is_first, is_last, is_trans = (lift(s) for s in sels[:3])
    acc = ZERO
    for c in b.constraints:
        acc = ef_add(ef_mul(acc, alpha), lift(c))
    its = [(m, v, True) for m, v in b.sends] + [(m, v, False) for m, v in b.receives]
    n_cols = len(perm_local)
    for col, c0 in enumerate(range(0, len(its), batch)):
        chunk = its[c0:c0 + batch]
        rlcs = [fingerprint_ext(perm_alpha, perm_beta, [lift(x) for x in vals]) for _, vals, _ in chunk]
        mults = [lift(m) if s else ef_neg(lift(m)) for m, _, s in chunk]
        product, numerator = ONE, ZERO
        for i, (m, rlc) in enumerate(zip(mults, rlcs)):
            product = ef_mul(product, rlc)
            others = ONE
            for j, o in enumerate(rlcs):
                if j != i:
                    others = ef_mul(others, o)
            numerator = ef_add(numerator, ef_mul(m, others))
        acc = ef_add(ef_mul(acc, alpha), ef_sub(ef_mul(product, perm_local[col]), numerator))
    sum_local, sum_next = ef_sum(perm_local[: n_cols - 1]), ef_sum(perm_next[: n_cols - 1])
    phi_local, phi_next = perm_local[-1], perm_next[-1]
    for c in (ef_mul(ef_sub(phi_local, sum_local), is_first), ef_mul(ef_sub(ef_sub(phi_next, phi_local), sum_next), is_trans),
              ef_mul(ef_sub(phi_local, cumulative_sum), is_last)):
        acc = ef_add(ef_mul(acc, alpha), c)
    return acc


def fingerprint_ext(alpha, beta, vals, kind=oair.INTERACTION_KIND_MEMORY):
    d = ef_add(alpha, ef(kind))
    bp = beta
    for v in vals:
        d = ef_add(d, ef_mul(bp, v))
        bp = ef_mul(bp, beta)
    return d


def quotient_chunks(air, log_n, main_lde, prep_lde, perm_lde, perm_alpha, perm_beta, alpha, cumulative_sum, public=(), lqd=1):
    """sphinx quotient_values + split_evals.  *_lde: NATURAL-order rows over the quotient domain 31 * <w_Q>
    (perm_lde rows are lists of EF tuples).  Returns 2^lqd chunk matrices of EF tuples (chunk c row r = value at
    31 * w_Q^(r * 2^lqd + c))."""
    q, qd = 1 << (log_n + lqd), 1 << lqd
    wq = two_adic_generator(log_n + lqd)
    vals = []
    for i in range(q):
        x = GEN * pow(wq, i, P) % P
        is_first, is_last, is_trans, inv_zh = selectors_at(x, log_n)
        nx = (i + qd) % q
        b = oair.Builder(main_lde[i], main_lde[nx], prep_lde[i] if prep_lde is not None else (), prep_lde[nx] if prep_lde is not None else (),
                         public, (is_first, is_last, is_trans))
        air.eval(b)
        folded = fold_constraints(b, perm_lde[i], perm_lde[nx], perm_alpha, perm_beta, qd, alpha, cumulative_sum, (is_first, is_last, is_trans))
        vals.append(ef_scale(folded, inv_zh))
    return [[vals[r * qd + c] for r in range(1 << log_n)] for c in range(qd)]
