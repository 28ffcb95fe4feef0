/* ORACLE directory (test infrastructure, not product code): the CPU PORT of the WHOLE proving step, timed by bench.py's
 * `cpu_baseline` leg next to the GPU step.
 *
 * cpu_port.c (included below) is the commitment round: coset LDE + Poseidon2-16 Merkle tree.  This file adds every other
 * stage of sphinx's prove_shard [UPSTREAM-RECALL] the way oracle/stark.py states it -- and is checked against it word for
 * word by tests/test_cpu_step.py (the Python there is the definition, this is the fast restatement):
 *   cp2_perm_trace     LogUp permutation trace of one chip + running sum          (stark.py: permutation_trace)
 *   cp2_quotient       quotient values on 31 <w_2N>, split into chunks            (stark.py: quotient_chunks / fold_constraints)
 *   cp2_open           opened values at zeta, zeta w by barycentric sums          (p3 interpolate_coset; verifier: pcs_verify)
 *   cp2_inv_denoms / cp2_reduce   reduced openings per LDE height                 (stark.py: pcs_verify, the prover's side)
 *   cp2_fri_fold       one FRI fold                                               (stark.py: pcs_verify's fold, the prover's side)
 *   cp2_pow_grind      proof-of-work search                                       (stark.py: Challenger.check_witness)
 *   cp2_commit / cp2_tree_open    Merkle tree that keeps its levels, and openings (commit.c: or_merkle_commit / or_merkle_verify)
 * The chips' constraint / interaction evaluators are C generated from the oracle's AIR by oracle/cpu_emit.py (cp_chips[]).
 * oracle/cpu_prover.py drives these stages with the oracle's transcript; its proofs are accepted by the oracle's verifier.
 * Everything is Montgomery form inside; OpenMP over rows.  Only tests/ and bench.py's cpu_baseline leg call this. */
#include "cpu_port.c"
#define CPS_HAVE_FIELD
#include "cpu_step.h"

#include <omp.h>

/* ------------------------------------------------------------------ extension field F[x] / (x^4 - 11), Montgomery lanes */
typedef struct {
    uint32_t c[4];
} ef;
#define W_M 3200u /* placeholder, replaced at init */
static uint32_t g_w_m = 0, g_one_m = 0;
static void ef_init(void) {
    if (!g_one_m) {
        g_w_m = to_m(11u);
        g_one_m = to_m(1u);
    }
}
static inline uint32_t red64(uint64_t x) { /* x 2^-32 mod p for any 64-bit x, in [0, p) */
    const uint32_t m = (uint32_t)x * CP_MU;
    const uint64_t u = ((uint64_t)m * OR_P) >> 32;
    int64_t r = (int64_t)(x >> 32) - (int64_t)u;
    if (r < 0) r += OR_P;
    if (r >= (int64_t)OR_P) r -= OR_P;
    if (r >= (int64_t)OR_P) r -= OR_P;
    return (uint32_t)r;
}
static inline ef ef_zero(void) { return (ef){{0, 0, 0, 0}}; }
static inline ef ef_from_base(uint32_t a_m) { return (ef){{a_m, 0, 0, 0}}; }
static inline ef ef_add(ef a, ef b) { return (ef){{madd(a.c[0], b.c[0]), madd(a.c[1], b.c[1]), madd(a.c[2], b.c[2]), madd(a.c[3], b.c[3])}}; }
static inline ef ef_sub(ef a, ef b) { return (ef){{msub(a.c[0], b.c[0]), msub(a.c[1], b.c[1]), msub(a.c[2], b.c[2]), msub(a.c[3], b.c[3])}}; }
static inline ef ef_neg(ef a) { return ef_sub(ef_zero(), a); }
static inline ef ef_scale(ef a, uint32_t s) { return (ef){{mm(a.c[0], s), mm(a.c[1], s), mm(a.c[2], s), mm(a.c[3], s)}}; }
static inline ef ef_mul(ef a, ef b) {
#define PR(i, j) ((uint64_t)a.c[i] * b.c[j])
    const uint32_t h0 = red64(PR(1, 3) + PR(2, 2) + PR(3, 1)), h1 = red64(PR(2, 3) + PR(3, 2)), h2 = red64(PR(3, 3));
    ef r;
    r.c[0] = madd(red64(PR(0, 0)), mm(h0, g_w_m));
    r.c[1] = madd(red64(PR(0, 1) + PR(1, 0)), mm(h1, g_w_m));
    r.c[2] = madd(red64(PR(0, 2) + PR(1, 1) + PR(2, 0)), mm(h2, g_w_m));
    r.c[3] = red64(PR(0, 3) + PR(1, 2) + PR(2, 1) + PR(3, 0));
#undef PR
    return r;
}
static inline uint32_t minv(uint32_t a_m) { return mpow(a_m, OR_P - 2); }
static ef ef_inv(ef a) { /* tower F[y]/(y^2 - 11), y = x^2: a = A + x B */
    const uint32_t W = g_w_m, a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
    const uint32_t n0 = msub(madd(mm(a0, a0), mm(W, mm(a2, a2))), mm(W, madd(mm(a1, a3), mm(a1, a3))));
    const uint32_t n1 = msub(msub(madd(mm(a0, a2), mm(a0, a2)), mm(a1, a1)), mm(W, mm(a3, a3)));
    const uint32_t d = minv(msub(mm(n0, n0), mm(W, mm(n1, n1))));
    const uint32_t m0 = mm(d, n0), m1 = msub(0, mm(d, n1));
    ef r;
    r.c[0] = madd(mm(a0, m0), mm(W, mm(a2, m1)));
    r.c[2] = madd(mm(a0, m1), mm(a2, m0));
    r.c[1] = msub(0, madd(mm(a1, m0), mm(W, mm(a3, m1))));
    r.c[3] = msub(0, madd(mm(a1, m1), mm(a3, m0)));
    return r;
}
/* v[i] <- 1 / v[i] for i < n (no zeros), one inversion */
static void ef_batch_inv(ef* v, size_t n, ef* scratch) {
    if (!n) return;
    ef acc = v[0];
    scratch[0] = acc;
    for (size_t i = 1; i < n; i++) {
        acc = ef_mul(acc, v[i]);
        scratch[i] = acc;
    }
    ef inv = ef_inv(acc);
    for (size_t i = n - 1; i > 0; i--) {
        const ef t = ef_mul(inv, scratch[i - 1]);
        inv = ef_mul(inv, v[i]);
        v[i] = t;
    }
    v[0] = inv;
}
static void base_batch_inv(uint32_t* v, size_t n, uint32_t* scratch) {
    if (!n) return;
    uint32_t acc = v[0];
    scratch[0] = acc;
    for (size_t i = 1; i < n; i++) {
        acc = mm(acc, v[i]);
        scratch[i] = acc;
    }
    uint32_t inv = minv(acc);
    for (size_t i = n - 1; i > 0; i--) {
        const uint32_t t = mm(inv, scratch[i - 1]);
        inv = mm(inv, v[i]);
        v[i] = t;
    }
    v[0] = inv;
}

void cp2_to_monty(const uint32_t* in, uint32_t* out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) out[i] = to_m(in[i]);
}
void cp2_from_monty(const uint32_t* in, uint32_t* out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) out[i] = from_m(in[i]);
}

/* ------------------------------------------------------------------ coset LDE with a shift, Montgomery in and out
 * in: n x w over the subgroup (natural order); out row bitrev(j) = the columns' polynomials at shift * w_{n << b}^j */
int cp2_lde(int log_n, int w, int log_blowup, const uint32_t* in_m, uint32_t* out_m, uint32_t shift_canonical) {
    const size_t n = (size_t)1 << log_n, m = n << log_blowup, ww = (size_t)w;
    const int log_m = log_n + log_blowup;
    uint32_t* coef = malloc(n * ww * sizeof(uint32_t));
    uint32_t* tw_inv = powers_m(mpow(root_of_unity_m(log_n), OR_P - 2), n / 2 ? n / 2 : 1);
    uint32_t* tw_fwd = powers_m(root_of_unity_m(log_m), m / 2 ? m / 2 : 1);
    uint32_t* shift = powers_m(to_m(shift_canonical), n);
    if (!coef || !tw_inv || !tw_fwd || !shift) {
        free(coef), free(tw_inv), free(tw_fwd), free(shift);
        return -1;
    }
    memcpy(coef, in_m, n * ww * sizeof(uint32_t));
    ntt_dif_rows(coef, log_n, ww, tw_inv);
    const uint32_t n_inv = mpow(to_m((uint32_t)(n % OR_P)), OR_P - 2);
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < n; k++) {
        uint32_t* dst = out_m + k * ww;
        memcpy(dst, coef + (size_t)bitrev32((uint32_t)k, log_n) * ww, ww * sizeof(uint32_t));
        scale_row(dst, mm(shift[k], n_inv), ww);
    }
    memset(out_m + n * ww, 0, (m - n) * ww * sizeof(uint32_t));
    ntt_dif_rows(out_m, log_m, ww, tw_fwd);
    free(coef), free(tw_inv), free(tw_fwd), free(shift);
    return 0;
}

/* ------------------------------------------------------------------ Merkle tree that keeps its levels */
typedef struct {
    int n_mats, log_max;
    const uint32_t** mats; /* not owned: Montgomery LDE matrices */
    uint32_t *log_h, *widths;
    uint32_t* digests; /* level l at offset level_off[l] * 8, Montgomery */
    size_t* level_off;
} cp2_tree;

void cp2_tree_free(cp2_tree* t) {
    if (!t) return;
    free(t->mats), free(t->log_h), free(t->widths), free(t->digests), free(t->level_off), free(t);
}

cp2_tree* cp2_commit(int n_mats, const uint32_t* const* mats, const uint32_t* log_h, const uint32_t* widths, uint32_t* root) {
    p16_init();
    cp2_tree* t = calloc(1, sizeof *t);
    t->n_mats = n_mats;
    t->mats = malloc(sizeof(uint32_t*) * (size_t)n_mats);
    t->log_h = malloc(sizeof(uint32_t) * (size_t)n_mats);
    t->widths = malloc(sizeof(uint32_t) * (size_t)n_mats);
    int log_max = 0;
    for (int i = 0; i < n_mats; i++) {
        t->mats[i] = mats[i], t->log_h[i] = log_h[i], t->widths[i] = widths[i];
        if ((int)log_h[i] > log_max) log_max = (int)log_h[i];
    }
    t->log_max = log_max;
    const size_t n_leaves = (size_t)1 << log_max;
    t->level_off = malloc(sizeof(size_t) * (size_t)(log_max + 1));
    size_t total = 0;
    for (int l = 0; l <= log_max; l++) {
        t->level_off[l] = total;
        total += n_leaves >> l;
    }
    t->digests = malloc(total * 8 * sizeof(uint32_t));
    int* which = malloc(sizeof(int) * (size_t)n_mats);
    int nw = 0;
    for (int i = 0; i < n_mats; i++)
        if ((int)log_h[i] == log_max) which[nw++] = i;
    uint32_t* a = t->digests;
    if (n_leaves >= VL) {
#pragma omp parallel for schedule(dynamic, 16)
        for (size_t r = 0; r < n_leaves; r += VL) sponge_v(mats, widths, which, nw, r, a + r * 8);
    } else {
        for (size_t r = 0; r < n_leaves; r++) sponge(mats, widths, which, nw, r, a + r * 8);
    }
    for (int l = 1; l <= log_max; l++) {
        const size_t n_par = n_leaves >> l;
        const uint32_t* prev = t->digests + t->level_off[l - 1] * 8;
        uint32_t* cur = t->digests + t->level_off[l] * 8;
        nw = 0;
        for (int i = 0; i < n_mats; i++)
            if ((int)log_h[i] == log_max - l) which[nw++] = i;
        if (n_par >= VL) {
#pragma omp parallel for schedule(dynamic, 16)
            for (size_t i = 0; i < n_par; i += VL) {
                uint32_t h[VL * 8];
                if (nw) sponge_v(mats, widths, which, nw, i, h);
                compress_v(prev + 2 * i * 8, nw ? h : NULL, cur + i * 8);
            }
        } else {
            for (size_t i = 0; i < n_par; i++) {
                uint32_t d[8];
                compress(prev + 2 * i * 8, prev + (2 * i + 1) * 8, d);
                if (nw) {
                    uint32_t h[8];
                    sponge(mats, widths, which, nw, i, h);
                    compress(d, h, cur + i * 8);
                } else {
                    memcpy(cur + i * 8, d, 32);
                }
            }
        }
    }
    free(which);
    const uint32_t* top = t->digests + t->level_off[log_max] * 8;
    for (int k = 0; k < 8; k++) root[k] = from_m(top[k]);
    return t;
}

/* rows_out: for every matrix in order its row index >> (log_max - log_h) (canonical, concatenated); path_out: 8 * log_max */
void cp2_tree_open(const cp2_tree* t, uint64_t index, uint32_t* rows_out, uint32_t* path_out) {
    size_t at = 0;
    for (int i = 0; i < t->n_mats; i++) {
        const size_t row = (size_t)(index >> (t->log_max - (int)t->log_h[i]));
        const uint32_t* src = t->mats[i] + row * t->widths[i];
        for (uint32_t c = 0; c < t->widths[i]; c++) rows_out[at++] = from_m(src[c]);
    }
    for (int l = 0; l < t->log_max; l++) {
        const uint32_t* sib = t->digests + (t->level_off[l] + ((index >> l) ^ 1)) * 8;
        for (int k = 0; k < 8; k++) path_out[l * 8 + k] = from_m(sib[k]);
    }
}

/* ------------------------------------------------------------------ LogUp permutation trace (stark.py: permutation_trace)
 * out: n x (4 * pw) with pw = ceil(n_inter / batch) + 1; last column = running sum; cumsum = its last entry */
#define CP2_MAX_INTER 512
#define CP2_MAX_TUPLE 64
int cp2_perm_width(int chip, int batch) {
    const cp_chip* c = &cp_chips[chip];
    const int n_int = (int)(c->n_sends + c->n_recvs);
    return (n_int + batch - 1) / batch + 1;
}
static void beta_powers(const uint32_t* beta, ef* bp, int n) { /* bp[t] = beta^t */
    ef b = {{beta[0], beta[1], beta[2], beta[3]}};
    bp[0] = ef_from_base(g_one_m);
    for (int t = 1; t < n; t++) bp[t] = ef_mul(bp[t - 1], b);
}
/* fingerprints d[j] of a row's interactions from its interaction block */
static inline void fingerprints(const cp_chip* c, const uint32_t* inter, ef alpha_plus_kind, const ef* bp, ef* d, uint32_t* mult) {
    const int n_int = (int)(c->n_sends + c->n_recvs);
    size_t at = 0;
    for (int j = 0; j < n_int; j++) {
        mult[j] = inter[at++];
        ef acc = alpha_plus_kind;
        for (uint32_t t = 0; t < c->tuple_len[j]; t++) acc = ef_add(acc, ef_scale(bp[t + 1], inter[at++]));
        d[j] = acc;
    }
}
int cp2_perm_trace(int chip, int log_n, const uint32_t* main, const uint32_t* prep, const uint32_t* pub, const uint32_t* alpha,
                   const uint32_t* beta, int batch, uint32_t* out, uint32_t* cumsum) {
    ef_init();
    const cp_chip* c = &cp_chips[chip];
    const size_t n = (size_t)1 << log_n;
    const int n_int = (int)(c->n_sends + c->n_recvs);
    if (n_int > CP2_MAX_INTER || batch < 1) return -1;
    const int pw = (n_int + batch - 1) / batch + 1;
    ef bp[CP2_MAX_TUPLE + 1];
    beta_powers(beta, bp, CP2_MAX_TUPLE + 1);
    ef apk = {{madd(alpha[0], g_one_m), alpha[1], alpha[2], alpha[3]}}; /* alpha + InteractionKind::Memory (= 1) */
#pragma omp parallel
    {
        uint32_t* inter = malloc(sizeof(uint32_t) * (c->inter_words + 1));
        ef d[CP2_MAX_INTER], scratch[CP2_MAX_INTER];
        uint32_t mult[CP2_MAX_INTER];
#pragma omp for schedule(static)
        for (size_t r = 0; r < n; r++) {
            c->inter(main + r * c->width, prep ? prep + r * c->prep_width : NULL, pub, inter);
            fingerprints(c, inter, apk, bp, d, mult);
            ef_batch_inv(d, (size_t)n_int, scratch);
            ef* row = (ef*)(out + r * (size_t)pw * 4);
            ef sum = ef_zero();
            for (int col = 0, j0 = 0; j0 < n_int; col++, j0 += batch) {
                ef acc = ef_zero();
                for (int j = j0; j < j0 + batch && j < n_int; j++) {
                    const ef term = ef_scale(d[j], mult[j]);
                    acc = j < (int)c->n_sends ? ef_add(acc, term) : ef_sub(acc, term);
                }
                row[col] = acc;
                sum = ef_add(sum, acc);
            }
            row[pw - 1] = sum; /* the row's own sum; the scan below turns it into the running sum */
        }
        free(inter);
    }
    /* running sum: per-thread blocks, then offsets */
    const int nt = omp_get_max_threads();
    ef* tot = calloc((size_t)nt + 1, sizeof(ef));
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        ef run = ef_zero();
        for (size_t r = lo; r < hi; r++) {
            ef* cell = (ef*)(out + (r * (size_t)pw + (size_t)(pw - 1)) * 4);
            run = ef_add(run, *cell);
            *cell = run;
        }
        tot[t + 1] = run;
#pragma omp barrier
#pragma omp single
        for (int k = 1; k <= nt; k++) tot[k] = ef_add(tot[k], tot[k - 1]);
        if (t > 0)
            for (size_t r = lo; r < hi; r++) {
                ef* cell = (ef*)(out + (r * (size_t)pw + (size_t)(pw - 1)) * 4);
                *cell = ef_add(*cell, tot[t]);
            }
    }
    free(tot);
    memcpy(cumsum, out + (n * (size_t)pw - 1) * 4, 16);
    return 0;
}

/* ------------------------------------------------------------------ quotient (stark.py: quotient_chunks, fold_constraints)
 * *_lde: 2N rows in the committed (bit-reversed) order over 31 <w_2N>; out: qd chunk matrices of N x 4 */
int cp2_quotient(int chip, int log_n, const uint32_t* main_lde, const uint32_t* prep_lde, const uint32_t* perm_lde, const uint32_t* pub,
                 const uint32_t* perm_alpha, const uint32_t* perm_beta, const uint32_t* alpha, const uint32_t* cumsum, int log_qd,
                 uint32_t* out) {
    ef_init();
    const cp_chip* c = &cp_chips[chip];
    const int qd = 1 << log_qd, log_q = log_n + log_qd, batch = qd;
    const size_t n = (size_t)1 << log_n, q = (size_t)1 << log_q;
    const int n_int = (int)(c->n_sends + c->n_recvs);
    if (n_int > CP2_MAX_INTER) return -1;
    const int pw = (n_int + batch - 1) / batch + 1;
    const int n_fold = (int)c->n_cons + (pw - 1) + 3;
    ef bp[CP2_MAX_TUPLE + 1];
    beta_powers(perm_beta, bp, CP2_MAX_TUPLE + 1);
    const ef apk = {{madd(perm_alpha[0], g_one_m), perm_alpha[1], perm_alpha[2], perm_alpha[3]}};
    const ef csum = {{cumsum[0], cumsum[1], cumsum[2], cumsum[3]}};
    /* Horner folding == constraint k weighs alpha^(n_fold - 1 - k) */
    ef* apow = malloc(sizeof(ef) * (size_t)n_fold);
    {
        ef a = {{alpha[0], alpha[1], alpha[2], alpha[3]}}, p = ef_from_base(g_one_m);
        for (int k = n_fold - 1; k >= 0; k--) {
            apow[k] = p;
            p = ef_mul(p, a);
        }
    }
    /* domain tables: x_i = 31 wq^i, selectors (unnormalised, p3 selectors_on_coset) */
    const uint32_t wq = root_of_unity_m(log_q), g = to_m(31u);
    const uint32_t w_inv = minv(root_of_unity_m(log_n));
    uint32_t* xs = malloc(q * 4);
    uint32_t* inv1 = malloc(q * 4);
    uint32_t* inv2 = malloc(q * 4);
    uint32_t* scratch = malloc(q * 4);
    xs[0] = g;
    for (size_t i = 1; i < q; i++) xs[i] = mm(xs[i - 1], wq);
    for (size_t i = 0; i < q; i++) {
        inv1[i] = msub(xs[i], g_one_m);
        inv2[i] = msub(xs[i], w_inv);
    }
    base_batch_inv(inv1, q, scratch);
    base_batch_inv(inv2, q, scratch);
    free(scratch);
    /* zh(x_i) = x_i^n - 1 takes qd distinct values (x^n = g^n (wq^n)^i, wq^n of order qd) */
    uint32_t zh[16], zh_inv[16];
    {
        const uint32_t gn = mpow(g, n), wn = mpow(wq, n);
        uint32_t cur = gn;
        for (int k = 0; k < qd; k++) {
            zh[k] = msub(cur, g_one_m);
            zh_inv[k] = minv(zh[k]);
            cur = mm(cur, wn);
        }
    }
    const size_t w = c->width, pwid = c->prep_width, permw = (size_t)pw * 4;
#pragma omp parallel
    {
        uint32_t* cons = malloc(sizeof(uint32_t) * (c->n_cons + 1));
        uint32_t* inter = malloc(sizeof(uint32_t) * (c->inter_words + 1));
        ef d[CP2_MAX_INTER];
        uint32_t mult[CP2_MAX_INTER];
#pragma omp for schedule(static)
        for (size_t i = 0; i < q; i++) {
            const size_t s = bitrev32((uint32_t)i, log_q), sn = bitrev32((uint32_t)((i + (size_t)qd) & (q - 1)), log_q);
            const int k = (int)(i & (size_t)(qd - 1));
            uint32_t sel[3] = {mm(zh[k], inv1[i]), mm(zh[k], inv2[i]), msub(xs[i], w_inv)};
            c->full(main_lde + s * w, main_lde + sn * w, prep_lde ? prep_lde + s * pwid : NULL, prep_lde ? prep_lde + sn * pwid : NULL, pub, sel,
                    cons, inter);
            ef acc = ef_zero();
            int kf = 0;
            for (uint32_t j = 0; j < c->n_cons; j++, kf++) acc = ef_add(acc, ef_scale(apow[kf], cons[j]));
            fingerprints(c, inter, apk, bp, d, mult);
            const ef* pl = (const ef*)(perm_lde + s * permw);
            const ef* pn = (const ef*)(perm_lde + sn * permw);
            for (int col = 0, j0 = 0; j0 < n_int; col++, j0 += batch, kf++) {
                const int j1 = j0 + batch < n_int ? j0 + batch : n_int;
                ef product = ef_from_base(g_one_m), numerator = ef_zero();
                for (int j = j0; j < j1; j++) {
                    product = ef_mul(product, d[j]);
                    ef others = ef_from_base(g_one_m);
                    for (int o = j0; o < j1; o++)
                        if (o != j) others = ef_mul(others, d[o]);
                    const ef term = ef_scale(others, mult[j]);
                    numerator = j < (int)c->n_sends ? ef_add(numerator, term) : ef_sub(numerator, term);
                }
                acc = ef_add(acc, ef_mul(apow[kf], ef_sub(ef_mul(product, pl[col]), numerator)));
            }
            ef sum_l = ef_zero(), sum_n = ef_zero();
            for (int col = 0; col < pw - 1; col++) {
                sum_l = ef_add(sum_l, pl[col]);
                sum_n = ef_add(sum_n, pn[col]);
            }
            const ef phi_l = pl[pw - 1], phi_n = pn[pw - 1];
            acc = ef_add(acc, ef_mul(apow[kf++], ef_scale(ef_sub(phi_l, sum_l), sel[0])));
            acc = ef_add(acc, ef_mul(apow[kf++], ef_scale(ef_sub(ef_sub(phi_n, phi_l), sum_n), sel[2])));
            acc = ef_add(acc, ef_mul(apow[kf++], ef_scale(ef_sub(phi_l, csum), sel[1])));
            const ef v = ef_scale(acc, zh_inv[k]);
            memcpy(out + ((size_t)k * n + (i >> log_qd)) * 4, v.c, 16);
        }
        free(cons), free(inter);
    }
    free(apow), free(xs), free(inv1), free(inv2);
    return 0;
}

/* ------------------------------------------------------------------ opened values: barycentric over the low coset 31 H
 * (the first N rows of the committed LDE, row bitrev(i) = value at 31 w^i); out[p][c] = f_c(z_p) */
int cp2_open(int log_n, int w, const uint32_t* lde, int n_pts, const uint32_t* zs, uint32_t* out) {
    ef_init();
    const size_t n = (size_t)1 << log_n;
    if (n_pts < 1 || n_pts > 2) return -1;
    const uint32_t g_inv = minv(to_m(31u)), wn = root_of_unity_m(log_n);
    ef* qv[2] = {NULL, NULL};
    ef factor[2];
    uint32_t* wp = powers_m(wn, n);
    for (int p = 0; p < n_pts; p++) {
        const ef u = ef_scale((ef){{zs[4 * p], zs[4 * p + 1], zs[4 * p + 2], zs[4 * p + 3]}}, g_inv);
        ef* den = malloc(sizeof(ef) * n);
        ef* scratch = malloc(sizeof(ef) * n);
        for (size_t i = 0; i < n; i++) den[i] = ef_sub(u, ef_from_base(wp[i]));
        ef_batch_inv(den, n, scratch);
        free(scratch);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) den[i] = ef_scale(den[i], wp[i]);
        qv[p] = den;
        ef un = ef_from_base(g_one_m), b = u; /* u^n */
        for (size_t e = n; e; e >>= 1) {
            if (e & 1) un = ef_mul(un, b);
            b = ef_mul(b, b);
        }
        factor[p] = ef_scale(ef_sub(un, ef_from_base(g_one_m)), minv(to_m((uint32_t)(n % OR_P))));
    }
    const int nt = omp_get_max_threads();
    ef* part = calloc((size_t)nt * (size_t)n_pts * (size_t)w, sizeof(ef));
#pragma omp parallel num_threads(nt)
    {
        ef* mine = part + (size_t)omp_get_thread_num() * (size_t)n_pts * (size_t)w;
#pragma omp for schedule(static)
        for (size_t i = 0; i < n; i++) {
            const uint32_t* row = lde + (size_t)bitrev32((uint32_t)i, log_n) * (size_t)w;
            for (int p = 0; p < n_pts; p++) {
                const ef qi = qv[p][i];
                ef* acc = mine + (size_t)p * (size_t)w;
                for (int c = 0; c < w; c++) acc[c] = ef_add(acc[c], ef_scale(qi, row[c]));
            }
        }
    }
    for (int p = 0; p < n_pts; p++)
        for (int c = 0; c < w; c++) {
            ef s = ef_zero();
            for (int t = 0; t < nt; t++) s = ef_add(s, part[((size_t)t * (size_t)n_pts + (size_t)p) * (size_t)w + (size_t)c]);
            s = ef_mul(s, factor[p]);
            memcpy(out + ((size_t)p * (size_t)w + (size_t)c) * 4, s.c, 16);
        }
    free(part), free(wp), free(qv[0]), free(qv[1]);
    return 0;
}

/* out[r] = 1 / (x_r - z), x_r = 31 w_h^bitrev(r): one table per (LDE height, point) */
int cp2_inv_denoms(int log_h, const uint32_t* z, uint32_t* out) {
    ef_init();
    const size_t h = (size_t)1 << log_h;
    uint32_t* wp = powers_m(root_of_unity_m(log_h), h);
    const uint32_t g = to_m(31u);
    const ef zz = {{z[0], z[1], z[2], z[3]}};
    ef* den = (ef*)out;
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < h; r++) den[r] = ef_sub(ef_from_base(mm(g, wp[bitrev32((uint32_t)r, log_h)])), zz);
    /* chunked batch inversion */
    const size_t chunk = 4096;
#pragma omp parallel
    {
        ef* scratch = malloc(sizeof(ef) * chunk);
#pragma omp for schedule(static)
        for (size_t at = 0; at < h; at += chunk) ef_batch_inv(den + at, h - at < chunk ? h - at : chunk, scratch);
        free(scratch);
    }
    free(wp);
    return 0;
}

/* ro[r] += sum_p apow0[p] * (sum_c alpha^c lde[r][c] - sum_c alpha^c ys[p][c]) * invd[p][r] */
int cp2_reduce(int log_h, int w, const uint32_t* lde, int n_pts, const uint32_t* const* invd, const uint32_t* ys, const uint32_t* alpha,
               const uint32_t* apow0, uint32_t* ro) {
    ef_init();
    const size_t h = (size_t)1 << log_h;
    ef* ap = malloc(sizeof(ef) * (size_t)(w + 1));
    ap[0] = ef_from_base(g_one_m);
    const ef a = {{alpha[0], alpha[1], alpha[2], alpha[3]}};
    for (int c = 1; c <= w; c++) ap[c] = ef_mul(ap[c - 1], a);
    ef Y[2], A0[2];
    for (int p = 0; p < n_pts; p++) {
        ef s = ef_zero();
        for (int c = 0; c < w; c++) s = ef_add(s, ef_mul(ap[c], *(const ef*)(ys + ((size_t)p * (size_t)w + (size_t)c) * 4)));
        Y[p] = s;
        A0[p] = *(const ef*)(apow0 + 4 * p);
    }
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < h; r++) {
        const uint32_t* row = lde + r * (size_t)w;
        ef t = ef_zero();
        for (int c = 0; c < w; c++) t = ef_add(t, ef_scale(ap[c], row[c]));
        ef acc = *(ef*)(ro + r * 4);
        for (int p = 0; p < n_pts; p++) acc = ef_add(acc, ef_mul(A0[p], ef_mul(ef_sub(t, Y[p]), *(const ef*)(invd[p] + r * 4))));
        memcpy(ro + r * 4, acc.c, 16);
    }
    free(ap);
    return 0;
}

/* out[i] = fold of the pair (cur[2i], cur[2i+1]) at beta: e0 + (beta - x0)(e1 - e0) / (x1 - x0), x0 = w^bitrev(i), x1 = -x0 */
int cp2_fri_fold(int log_size, const uint32_t* cur, const uint32_t* beta, uint32_t* out) {
    ef_init();
    const size_t half = (size_t)1 << (log_size - 1);
    const uint32_t w_inv = minv(root_of_unity_m(log_size));
    uint32_t* ip = powers_m(w_inv, half); /* x0^-1 = w^-bitrev(i) */
    const uint32_t half_inv = minv(to_m(2u));
    const ef b = {{beta[0], beta[1], beta[2], beta[3]}};
    uint32_t* fw = powers_m(root_of_unity_m(log_size), half);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < half; i++) {
        const size_t k = log_size > 1 ? bitrev32((uint32_t)i, log_size - 1) : 0;
        const ef e0 = *(const ef*)(cur + 8 * i), e1 = *(const ef*)(cur + 8 * i + 4);
        const uint32_t x0 = fw[k];
        const uint32_t neg_inv_2x0 = msub(0, mm(half_inv, ip[k])); /* 1 / (x1 - x0) = -1 / (2 x0) */
        const ef slope = ef_scale(ef_sub(e1, e0), neg_inv_2x0);
        const ef r = ef_add(e0, ef_mul(ef_sub(b, ef_from_base(x0)), slope));
        memcpy(out + 4 * i, r.c, 16);
    }
    free(ip), free(fw);
    return 0;
}

/* smallest witness w such that, with `pending` + [w] written over the first lanes of `state`, the permuted state's lane `lane`
 * has `bits` zero low bits (canonical value); state / pending canonical */
uint32_t cp2_pow_grind(const uint32_t* state, int n_pending, const uint32_t* pending, int bits, int lane) {
    p16_init();
    uint32_t base[16];
    for (int i = 0; i < 16; i++) base[i] = to_m(state[i]);
    for (int i = 0; i < n_pending; i++) base[i] = to_m(pending[i]);
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t found = 0xFFFFFFFFu;
    for (uint32_t start = 0; found == 0xFFFFFFFFu && start < OR_P; start += 1u << 16) {
#pragma omp parallel for schedule(static)
        for (uint32_t wv = start; wv < start + (1u << 16); wv++) {
            if (wv >= OR_P) continue;
            uint32_t s[16];
            memcpy(s, base, sizeof s);
            s[n_pending] = to_m(wv);
            perm16(s);
            if ((from_m(s[lane]) & mask) == 0) {
#pragma omp critical
                if (wv < found) found = wv;
            }
        }
    }
    return found;
}

int cp2_n_chips(void) { return cp_n_chips; }

/* nanoseconds per width-16 permutation on the calling core (sixteen states at a time: the rate the tree hashing runs at), and
 * whether the AVX-512 routine is the one in use: printed beside the baseline so that its quality can be judged. */
#include <time.h>
double cp2_perm_ns(int iters, int* avx512) {
    p16_init();
    vstate s;
    for (int i = 0; i < 16; i++)
        for (int l = 0; l < VL; l++) s[i][l] = (uint32_t)(i * 131 + l * 7919 + 1);
    struct timespec a, b;
    perm16_v(s);
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int k = 0; k < iters; k++) perm16_v(s);
    clock_gettime(CLOCK_MONOTONIC, &b);
#if defined(__x86_64__)
    if (avx512) *avx512 = g_have_512 > 0;
#else
    if (avx512) *avx512 = 0;
#endif
    volatile uint32_t sink = s[3][5];
    (void)sink;
    return ((double)(b.tv_sec - a.tv_sec) * 1e9 + (double)(b.tv_nsec - a.tv_nsec)) / ((double)iters * VL);
}
const char* cp2_chip_name(int i) { return cp_chips[i].name; }
