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


# ------------------------------------------------------------------ protocol profile
class Profile:
    """Mirror of include/lurkhip.h: lurkhip_protocol_profile -- every recalled choice of the commit / transcript / FRI layer,
    field by field.  `install()` hands the permutation tables to the oracle's C side (Merkle tree, sponge, transcript)."""

    FIELDS = {"p16_rounds_p": 13, "p16_ext_rc": None, "p16_int_rc": None, "p16_diag": None, "p16_internal_scale": 1,
              "challenger_squeeze": 8, "challenger_pop_front": 0, "observe_openings": 0, "observe_chip_meta": 0,
              "constraint_alpha_ascending": 0, "fri_alpha_global": 0, "fri_log_arity": 1, "fri_log_blowup": 1, "fri_num_queries": 100,
              "fri_pow_bits": 16, "serialize_montgomery": 0}

    def __init__(self, **kw):
        for k, v in self.FIELDS.items():
            setattr(self, k, v)
        for k, v in kw.items():
            assert k in self.FIELDS, f"unknown profile field {k}"
            setattr(self, k, v)

    @classmethod
    def from_dict(cls, d):
        return cls(**{k: v for k, v in d.items() if k in cls.FIELDS})

    def install(self):
        from . import binding

        assert self.fri_log_arity == 1, "only FRI folding by 2 is restated"
        if self.p16_ext_rc is None:
            assert self.p16_internal_scale == 1
            binding.set_p16()
        else:
            flat = [x for row in self.p16_ext_rc for x in (row if isinstance(row, (list, tuple)) else [row])]
            binding.set_p16(self.p16_rounds_p, flat, list(self.p16_int_rc) + [0] * (32 - len(self.p16_int_rc)), self.p16_diag, self.p16_internal_scale)
        return self


DEFAULT_PROFILE = Profile()


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
def fold_constraints(b: "oair.Builder", perm_local, perm_next, perm_alpha, perm_beta, batch, alpha, cumulative_sum, sels, ascending=False):
    """sphinx Chip::eval with the folding builder: the chip's constraints, then eval_permutation_constraints;
    accumulator = accumulator * alpha + constraint.  perm rows are lists of EF tuples.  `sels` may hold base or
    extension values (the verifier evaluates at an extension point); everything is lifted to EF."""

    def lift(v):
        return v if isinstance(v, tuple) else ef(v)

    is_first, is_last, is_trans = (lift(s) for s in sels[:3])
    folded_terms = []  # every constraint in folding order; combined at the end (Horner, or ascending powers)
    acc = ZERO
    for c in b.constraints:
        folded_terms.append(lift(c))
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
        folded_terms.append(ef_sub(ef_mul(product, perm_local[col]), numerator))
    sum_local, sum_next = ef_sum(perm_local[: n_cols - 1]), ef_sum(perm_next[: n_cols - 1])
    phi_local, phi_next = perm_local[-1], perm_next[-1]
    for c in (ef_mul(ef_sub(phi_local, sum_local), is_first), ef_mul(ef_sub(ef_sub(phi_next, phi_local), sum_next), is_trans),
              ef_mul(ef_sub(phi_local, cumulative_sum), is_last)):
        folded_terms.append(c)
    if ascending:  # constraint k weighs alpha^k
        pw = ONE
        for c in folded_terms:
            acc = ef_add(acc, ef_mul(pw, c))
            pw = ef_mul(pw, alpha)
    else:  # sphinx's folders: accumulator = accumulator * alpha + constraint
        for c in folded_terms:
            acc = ef_add(ef_mul(acc, alpha), c)
    return acc


def fingerprint_ext(alpha, beta, vals, kind=oair.INTERACTION_KIND_MEMORY):
    d = ef_add(alpha, ef(kind))
    bp = beta
    for v in vals:
        d = ef_add(d, ef_mul(bp, v))
        bp = ef_mul(bp, beta)
    return d


def quotient_chunks(air, log_n, main_lde, prep_lde, perm_lde, perm_alpha, perm_beta, alpha, cumulative_sum, public=(), lqd=1, ascending=False):
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
        folded = fold_constraints(b, perm_lde[i], perm_lde[nx], perm_alpha, perm_beta, qd, alpha, cumulative_sum, (is_first, is_last, is_trans), ascending)
        vals.append(ef_scale(folded, inv_zh))
    return [[vals[r * qd + c] for r in range(1 << log_n)] for c in range(qd)]


# ------------------------------------------------------------------ extension values with operators (verifier-side AIR evaluation)
class EFv:
    """Extension-field value with int-compatible operators so that oracle/air.py's numeric walk can run at an
    out-of-domain extension point (what sphinx's VerifierConstraintFolder does)."""

    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v if isinstance(v, tuple) else ef(v)

    @staticmethod
    def lift(x):
        return x if isinstance(x, EFv) else EFv(ef(x))

    def __add__(self, o):
        return EFv(ef_add(self.v, EFv.lift(o).v))

    __radd__ = __add__

    def __sub__(self, o):
        return EFv(ef_sub(self.v, EFv.lift(o).v))

    def __rsub__(self, o):
        return EFv(ef_sub(EFv.lift(o).v, self.v))

    def __mul__(self, o):
        return EFv(ef_mul(self.v, EFv.lift(o).v))

    __rmul__ = __mul__

    def __neg__(self):
        return EFv(ef_neg(self.v))

    def __mod__(self, _):
        return self

    def __eq__(self, o):
        return self.v == EFv.lift(o).v

    def __hash__(self):
        return hash(self.v)


def _raw(x):
    return x.v if isinstance(x, EFv) else ef(x)


# ------------------------------------------------------------------ transcript (p3 DuplexChallenger<_, Perm16, 16, 8>)
class Challenger:
    def __init__(self, permute16, profile=None):
        self.perm = permute16  # list of 16 canonical ints -> list of 16
        self.profile = profile or DEFAULT_PROFILE
        self.state = [0] * 16
        self.input = []
        self.output = []

    def clone(self):
        c = Challenger(self.perm, self.profile)
        c.state, c.input, c.output = list(self.state), list(self.input), list(self.output)
        return c

    def _duplexing(self):
        for i, v in enumerate(self.input):
            self.state[i] = v
        self.input = []
        self.state = self.perm(self.state)
        self.output = list(self.state[: self.profile.challenger_squeeze])

    def observe(self, v):
        if isinstance(v, (list, tuple)):
            for x in v:
                self.observe(x)
            return
        self.output = []
        self.input.append(int(v) % P)
        if len(self.input) == 8:
            self._duplexing()

    def sample(self):
        if self.input or not self.output:
            self._duplexing()
        return self.output.pop(0) if self.profile.challenger_pop_front else self.output.pop()

    def sample_ext(self):
        return tuple(self.sample() for _ in range(4))

    def sample_bits(self, bits):
        return self.sample() & ((1 << bits) - 1)

    def check_witness(self, bits, witness):
        self.observe(witness)
        return self.sample_bits(bits) == 0


def default_permute16():
    """The width-16 permutation installed in the oracle's C side (Profile.install; default: the reference's BabyBearConfig16)."""
    from . import binding

    return binding.perm16


# ------------------------------------------------------------------ verifier
class VerifyError(AssertionError):
    pass


def _need(cond, msg):
    if not cond:
        raise VerifyError(msg)


def _unflatten(vals):
    """4 opened base-column values (each EF) -> one EF: sum_e basis_e * v_e (basis_e = x^e)."""
    out = []
    for j in range(0, len(vals), 4):
        acc = ZERO
        for e in range(4):
            mono = tuple(1 if k == e else 0 for k in range(4))
            acc = ef_add(acc, ef_mul(mono, vals[j + e]))
        out.append(acc)
    return out


def eval_constraints_at(air, chip, sels, alpha, perm_alpha, perm_beta, public, ascending=False):
    """sphinx Verifier::eval_constraints: the chip's AIR on the opened values with a folding builder."""
    o = chip.opened
    prep_l, prep_n = o.get("prep", ([], []))
    b = oair.Builder([EFv(v) for v in o["main"][0]], [EFv(v) for v in o["main"][1]], [EFv(v) for v in prep_l], [EFv(v) for v in prep_n],
                     list(public), tuple(EFv(s) for s in sels[:3]))
    # the numeric builder reduces with `% P`; EFv ignores it
    b.assert_zero = lambda x, cond=None: b.constraints.append(x if cond is None else cond * x)
    b.receive = lambda values, is_real: b.receives.append((is_real, list(values)))
    b.send = lambda values, is_real: b.sends.append((is_real, list(values)))
    air.eval(b)
    b.constraints = [_raw(c) for c in b.constraints]
    b.sends = [(_raw(m), [_raw(v) for v in vals]) for m, vals in b.sends]
    b.receives = [(_raw(m), [_raw(v) for v in vals]) for m, vals in b.receives]
    perm_local, perm_next = _unflatten(o["perm"][0]), _unflatten(o["perm"][1])
    return fold_constraints(b, perm_local, perm_next, perm_alpha, perm_beta, chip.quotient_degree, alpha, chip.cumulative_sum, sels, ascending)


def recompute_quotient(chip, zeta):
    """sphinx Verifier::recompute_quotient: chunk c lives on the coset 31 * w_Q^c * H."""
    log_n, qd = chip.log_n, chip.quotient_degree
    lqd = qd.bit_length() - 1
    n = 1 << log_n
    wq = two_adic_generator(log_n + lqd)
    shifts = [GEN * pow(wq, c, P) % P for c in range(qd)]

    def zp_at(shift, x):  # (x / shift)^n - 1, x an EF tuple
        return ef_sub(ef_scale(ef_pow(x, n), finv(pow(shift, n, P))), ONE)

    total = ZERO
    for i in range(qd):
        zps = ONE
        for j in range(qd):
            if j != i:
                num = zp_at(shifts[j], zeta)
                den = zp_at(shifts[j], ef(shifts[i]))
                zps = ef_mul(zps, ef_mul(num, ef_inv(den)))
        chunk = chip.opened["quotient"][i]
        acc = ZERO
        for e in range(4):
            mono = tuple(1 if k == e else 0 for k in range(4))
            acc = ef_add(acc, ef_mul(mono, chunk[e]))
        total = ef_add(total, ef_mul(zps, acc))
    return total


def selectors_at_point(zeta, log_n):
    n = 1 << log_n
    zh = ef_sub(ef_pow(zeta, n), ONE)
    w_inv = finv(two_adic_generator(log_n))
    return (ef_mul(zh, ef_inv(ef_sub(zeta, ONE))), ef_mul(zh, ef_inv(ef_sub(zeta, ef(w_inv)))), ef_sub(zeta, ef(w_inv)), ef_inv(zh))


def pcs_verify(rounds, proof, log_blowup, challenger, merkle_verify):
    """p3 TwoAdicFriPcs::verify + p3_fri::verifier.  rounds = [(root, [(log_n, width, [(point, values)])])] in the prover's
    order; `proof` carries fri_roots, final_poly, pow_bits, pow_witness, log_max_height, num_queries, query_indices,
    round_openings, layer_openings (a ShardProof or a standalone opening).  The challenger is advanced like the prover's."""
    prof = challenger.profile
    if prof.observe_openings:
        for _, mats in rounds:
            for _, _, pts in mats:
                for _, values in pts:
                    for v in values:
                        challenger.observe(list(v))
    alpha_fri = challenger.sample_ext()
    betas = []
    for root in proof.fri_roots:
        challenger.observe(root)
        betas.append(challenger.sample_ext())
    challenger.observe(list(proof.final_poly))
    _need(challenger.check_witness(proof.pow_bits, proof.pow_witness), "invalid proof-of-work witness")
    log_max = len(proof.fri_roots) + log_blowup
    _need(log_max == proof.log_max_height and log_max <= 27, "log_max_height")
    indices = [challenger.sample_bits(log_max) for _ in range(proof.num_queries)]
    if proof.query_indices is not None:  # (the upstream proof format does not carry them: oracle/wire.py)
        _need(indices == proof.query_indices, "query indices differ from the transcript's")

    for qi, index in enumerate(indices):
        ro = {}
        alpha_pow = {}
        for (root, mats), (rw, records) in zip(rounds, proof.round_openings):
            rec = records[qi]
            log_hs = [lg + log_blowup for lg, _, _ in mats]
            widths = [w for _, w, _ in mats]
            log_batch_max = max(log_hs)
            _need(rw == sum(widths) + 8 * log_batch_max, "round record size")
            reduced_index = index >> (log_max - log_batch_max)
            rows, path = rec[: sum(widths)], rec[sum(widths):]
            _need(merkle_verify(log_hs, widths, reduced_index, rows, path, root), f"Merkle opening of query {qi} fails")
            off = 0
            for (log_n, w, pts), log_h in zip(mats, log_hs):
                row = rows[off:off + w]
                off += w
                rev = bitrev(index >> (log_max - log_h), log_h)
                x = GEN * pow(two_adic_generator(log_h), rev, P) % P
                for z, ps_at_z in pts:
                    _need(len(ps_at_z) == w, "opened values shape")
                    inv_d = ef_inv(ef_sub(ef(x), z))
                    for p_at_x, p_at_z in zip(row, ps_at_z):
                        quotient = ef_mul(ef_sub(ef(p_at_x), p_at_z), inv_d)
                        key = 0 if prof.fri_alpha_global else log_h  # p3: one power offset per LDE height
                        ap = alpha_pow.get(key, ONE)
                        ro[log_h] = ef_add(ro.get(log_h, ZERO), ef_mul(ap, quotient))
                        alpha_pow[key] = ef_mul(ap, alpha_fri)
        # heights of 2^log_blowup rows (one-row traces) never enter the fold below: their reduced openings must be zero, or the
        # opened values of such a chip are bound to nothing (the later p3 fix of verify_query)
        for log_h, acc in ro.items():
            _need(log_h > log_blowup or acc == ZERO, f"query {qi}: the reduced opening at height 2^{log_h} is not zero")
        # ---- fri verify_query
        folded = ZERO
        idx = index
        x = pow(two_adic_generator(log_max), bitrev(index, log_max), P)
        for li, (log_folded, root, beta) in enumerate(zip(range(log_max - 1, -1, -1), proof.fri_roots, betas)):
            folded = ef_add(folded, ro.get(log_folded + 1, ZERO))
            rw, records = proof.layer_openings[li]
            rec = records[qi]
            if getattr(proof, "sibling_only", False):
                # p3's CommitPhaseProofStep carries the sibling only: the queried element of the pair IS the running fold, and the
                # Merkle opening of the pair binds it
                _need(rw == 4 + 8 * log_folded, "layer record size")
                evals = [None, None]
                evals[idx % 2], evals[(idx ^ 1) % 2] = folded, tuple(rec[:4])
                pair, path = list(evals[0]) + list(evals[1]), rec[4:]
            else:
                _need(rw == 8 + 8 * log_folded, "layer record size")
                pair, path = rec[:8], rec[8:]
                evals = [tuple(pair[0:4]), tuple(pair[4:8])]
                _need(evals[idx % 2] == folded, f"query {qi}: layer {li} does not continue the fold")
            _need(merkle_verify([log_folded], [8], idx >> 1, pair, path, root), f"query {qi}: FRI layer {li} opening fails")
            xs = [x, x]
            xs[(idx ^ 1) % 2] = xs[(idx ^ 1) % 2] * (P - 1) % P  # times the generator of the order-2 subgroup
            # interpolate through (xs[0], evals[0]), (xs[1], evals[1]) and evaluate at beta
            slope = ef_scale(ef_sub(evals[1], evals[0]), finv(xs[1] - xs[0]))
            folded = ef_add(evals[0], ef_mul(ef_sub(beta, ef(xs[0])), slope))
            idx >>= 1
            x = x * x % P
        _need(idx < (1 << log_blowup), "final index")
        _need(folded == proof.final_poly, f"query {qi}: final polynomial mismatch")



def verify_shard(airs_by_machine_index, vk_root, prep_log_heights, prep_widths, proof, challenger, merkle_verify):
    """sphinx Verifier::verify_shard + p3 TwoAdicFriPcs::verify + p3_fri::verifier.  `challenger` must be in the state
    the prover's was when prove_shard started.  Returns the chips' cumulative sums."""
    chips = proof.chips
    log_blowup = proof.log_blowup
    prof = challenger.profile
    if prof.observe_chip_meta:
        for c in chips:
            challenger.observe([c.log_n, c.width, c.prep_index + 1])
    perm_alpha, perm_beta = challenger.sample_ext(), challenger.sample_ext()
    challenger.observe(proof.perm_root)
    if prof.observe_chip_meta:
        for c in chips:
            challenger.observe(list(c.cumulative_sum))
    alpha = challenger.sample_ext()
    challenger.observe(proof.quot_root)
    zeta = challenger.sample_ext()

    # ---- rounds: (root, [(log_n, width, [(point, values)])]) in the prover's order
    def next_point(log_n):
        return ef_scale(zeta, two_adic_generator(log_n))

    rounds = []
    if proof.n_preprocessed:
        by_idx = {c.prep_index: c for c in chips if c.prep_index >= 0}
        mats = []
        for m in range(proof.n_preprocessed):
            c = by_idx[m]
            _need(c.log_n == prep_log_heights[m] and c.prep_width == prep_widths[m], "preprocessed shape")
            mats.append((c.log_n, c.prep_width, [(zeta, c.opened["prep"][0]), (next_point(c.log_n), c.opened["prep"][1])]))
        rounds.append((vk_root, mats))
    rounds.append((proof.main_root, [(c.log_n, c.width, [(zeta, c.opened["main"][0]), (next_point(c.log_n), c.opened["main"][1])]) for c in chips]))
    rounds.append((proof.perm_root, [(c.log_n, c.perm_width, [(zeta, c.opened["perm"][0]), (next_point(c.log_n), c.opened["perm"][1])]) for c in chips]))
    rounds.append((proof.quot_root, [(c.log_n, 4, [(zeta, chunk)]) for c in chips for chunk in c.opened["quotient"]]))

    # ---- pcs.verify
    pcs_verify(rounds, proof, log_blowup, challenger, merkle_verify)

    # ---- constraints at zeta
    for c in chips:
        air = airs_by_machine_index[c.machine_index]
        _need(air.width == c.width, "chip width")
        sels = selectors_at_point(zeta, c.log_n)
        folded = eval_constraints_at(air, c, sels, alpha, perm_alpha, perm_beta, proof.public_values, bool(prof.constraint_alpha_ascending))
        quotient = recompute_quotient(c, zeta)
        _need(ef_mul(folded, sels[3]) == quotient, f"constraints of chip {c.machine_index} do not match the quotient at zeta")
    return [c.cumulative_sum for c in chips]


def verify_machine(airs_by_machine_index, vk_root, prep_log_heights, prep_widths, proofs, merkle_verify, permute16=None, profile=None):
    """sphinx StarkMachine::verify: rebuild the transcript, verify every shard, and check that the cumulative
    sums of all chips of all shards cancel.  `profile` (default DEFAULT_PROFILE) must have been install()ed when it carries
    its own permutation tables."""
    ch = Challenger(permute16 or default_permute16(), profile)
    ch.observe(vk_root)
    ch.observe(0)
    for pr in proofs:
        ch.observe(pr.main_root)
        ch.observe(pr.public_values)
    total = ZERO
    for pr in proofs:
        for cs in verify_shard(airs_by_machine_index, vk_root, prep_log_heights, prep_widths, pr, ch.clone(), merkle_verify):
            total = ef_add(total, cs)
    _need(total == ZERO, "cumulative sums do not cancel")
    return True
