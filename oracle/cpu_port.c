/* ORACLE directory (test infrastructure, not product code): the CPU PORT timed by bench.py's `cpu_baseline` leg.
 *
 * A commitment round the way a CPU prover would run it -- p3 TwoAdicFriPcs::commit [UPSTREAM-RECALL, Plonky3 @ a0b92870]:
 * coset LDE of every matrix (blow-up 2^b, shift g = 31, rows bit-reversed) + FieldMerkleTreeMmcs over
 * PaddingFreeSponge<Poseidon2-16, 16, 8, 8> / TruncatedPermutation -- written for speed where commit.c (the checker the
 * GPU results are compared with) is written to be obviously right: Montgomery arithmetic, row-major radix-2 NTTs whose inner
 * loop runs along a row (contiguous, vectorised: AVX-512 / AVX2 clones picked at load time), OpenMP over butterflies and
 * rows.  tests/test_cpu_port.py checks it word for word against commit.c.  It is neither the reference binary (Rust:
 * not buildable here) nor the checker; only bench.py's baseline and that test call it.
 *
 * Poseidon2-16 tables: the oracle's own (or_p2_lookup(16), the reference's BabyBearConfig16,
 * /root/reference/src/poseidon/config.rs:190-199), converted to Montgomery form at first use.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "field.h"

typedef struct { /* poseidon2.c */
    int width;
    int rounds_p;
    const uint32_t* diag;
    const uint32_t* ext_rc;
    const uint32_t* int_rc;
} or_p2_params;
int or_p2_lookup(int width, or_p2_params* out);

#define CP_MU 0x88000001u /* p^-1 mod 2^32 */
#define CP_R2 1172168163u /* 2^64 mod p */

static inline uint32_t mm(uint32_t a, uint32_t b) { /* a b 2^-32 mod p, operands and result in [0, p) */
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * CP_MU;
    const uint32_t u = (uint32_t)(((uint64_t)m * OR_P) >> 32), hi = (uint32_t)(t >> 32);
    const uint32_t r = hi - u;
    return hi < u ? r + OR_P : r;
}
static inline uint32_t madd(uint32_t a, uint32_t b) {
    const uint32_t s = a + b;
    return s >= OR_P ? s - OR_P : s;
}
static inline uint32_t msub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + OR_P - b; }
static inline uint32_t to_m(uint32_t x) { return mm(x, CP_R2); }
static inline uint32_t from_m(uint32_t x) { return mm(x, 1u); }
static uint32_t mpow(uint32_t a_m, uint64_t e) {
    uint32_t r = to_m(1);
    while (e) {
        if (e & 1) r = mm(r, a_m);
        a_m = mm(a_m, a_m);
        e >>= 1;
    }
    return r;
}

/* one DIF butterfly between two rows: x <- x + y, y <- (x - y) t */
__attribute__((target_clones("avx512f", "avx2", "default"))) static void bfly_rows(uint32_t* restrict x, uint32_t* restrict y, uint32_t t,
                                                                                 size_t w) {
    for (size_t c = 0; c < w; c++) {
        const uint32_t u = x[c], v = y[c];
        x[c] = madd(u, v);
        y[c] = mm(msub(u, v), t);
    }
}
__attribute__((target_clones("avx512f", "avx2", "default"))) static void scale_row(uint32_t* restrict x, uint32_t s, size_t w) {
    for (size_t c = 0; c < w; c++) x[c] = mm(x[c], s);
}

static uint32_t root_of_unity_m(int bits) { /* the generator commit.c uses: 0x1a427a41 is a primitive 2^27-th root [UPSTREAM-RECALL] */
    uint32_t r = to_m(0x1a427a41u);
    for (int i = bits; i < 27; i++) r = mm(r, r);
    return r;
}

/* in-place DIF over the rows of an n x w matrix (natural order in, bit-reversed out); tw[k] = root^k, k < n / 2.
 * Cache-blocked (round 5: the step-by-step version passed over the whole matrix once per stage, 20 times for 2^20 rows): the stages
 * are taken in groups of B, B such that 2^B rows fit a core's L2; within a group of stages [lo, hi) the rows that agree outside
 * index bits lo .. hi - 1 form an independent 2^(hi - lo)-point transform, done stage by stage while its rows are hot -- a
 * four-step decomposition in place, two passes over memory for 2^20 rows of a few hundred columns instead of twenty. */
static void ntt_dif_rows(uint32_t* a, int log_n, size_t w, const uint32_t* tw) {
    const size_t n = (size_t)1 << log_n;
    int B = 1;
    while (B < log_n && ((w * sizeof(uint32_t)) << (B + 1)) <= ((size_t)768 << 10)) B++;
    for (int hi = log_n; hi > 0;) {
        const int lo = hi > B ? hi - B : 0, bits = hi - lo;
        const size_t n_low = (size_t)1 << lo, n_groups = n >> bits, pts = (size_t)1 << bits;
#pragma omp parallel for schedule(static)
        for (size_t g = 0; g < n_groups; g++) {
            const size_t blk = g >> lo, j0 = g & (n_low - 1);
            uint32_t* base = a + ((blk << hi) + j0) * w; /* point k of the group is row (blk << hi) + (k << lo) + j0 */
            for (int s = hi - 1; s >= lo; s--) {
                const size_t half_k = (size_t)1 << (s - lo), stride = n >> (s + 1);
                for (size_t b = 0; b < pts / 2; b++) {
                    const size_t kb = b >> (s - lo), kj = b & (half_k - 1);
                    const size_t k = (kb << (s - lo + 1)) + kj;              /* lower point of the pair */
                    const size_t j = (kj << lo) + j0;                        /* its index within the stage's half block */
                    uint32_t* x = base + (k << lo) * w;
                    bfly_rows(x, x + (half_k << lo) * w, tw[j * stride], w);
                }
            }
        }
        hi = lo;
    }
}

static uint32_t* powers_m(uint32_t root_m, size_t count) {
    uint32_t* t = malloc((count ? count : 1) * sizeof(uint32_t));
    if (!t) return NULL;
    /* blocks of 4096 powers, each started from root^(block start): parallel, 12 squarings of set-up per block */
#pragma omp parallel for schedule(static)
    for (size_t b0 = 0; b0 < count; b0 += 4096) {
        uint32_t x = mpow(root_m, b0);
        const size_t e = b0 + 4096 < count ? b0 + 4096 : count;
        for (size_t k = b0; k < e; k++) {
            t[k] = x;
            x = mm(x, root_m);
        }
    }
    return t;
}

static uint32_t bitrev32(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* in: n x w canonical (natural order over the subgroup); out: (n << b) x w, Montgomery when out_monty else canonical, row
 * bitrev(j) = the columns' polynomials at 31 * w_{n << b}^j.  Same definition as commit.c: or_lde_fft. */
int cp_lde(int log_n, int w, int log_blowup, const uint32_t* in, uint32_t* out, int out_monty) {
    const size_t n = (size_t)1 << log_n, m = n << log_blowup, ww = (size_t)w;
    const int log_m = log_n + log_blowup;
    uint32_t* coef = malloc(n * ww * sizeof(uint32_t));
    uint32_t* tw_inv = powers_m(mpow(root_of_unity_m(log_n), OR_P - 2), n / 2);
    uint32_t* tw_fwd = powers_m(root_of_unity_m(log_m), m / 2);
    uint32_t* shift = powers_m(to_m(31u), n);
    if (!coef || !tw_inv || !tw_fwd || !shift) {
        free(coef), free(tw_inv), free(tw_fwd), free(shift);
        return -1;
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n * ww; i++) coef[i] = to_m(in[i]);
    ntt_dif_rows(coef, log_n, ww, tw_inv); /* coefficient k (times n) now sits in row bitrev(k) */
    const uint32_t n_inv = mpow(to_m((uint32_t)(n % OR_P)), OR_P - 2);
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < n; k++) { /* natural order, scaled by shift^k / n, into the first n rows of the padded input */
        uint32_t* dst = out + k * ww;
        memcpy(dst, coef + (size_t)bitrev32((uint32_t)k, log_n) * ww, ww * sizeof(uint32_t));
        scale_row(dst, mm(shift[k], n_inv), ww);
    }
    memset(out + n * ww, 0, (m - n) * ww * sizeof(uint32_t));
    ntt_dif_rows(out, log_m, ww, tw_fwd); /* row r = evaluation index bitrev(r): the committed order */
    if (!out_monty) {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < m * ww; i++) out[i] = from_m(out[i]);
    }
    free(coef), free(tw_inv), free(tw_fwd), free(shift);
    return 0;
}

/* ---- Poseidon2 width 16 on Montgomery words */
static uint32_t g_ext[128], g_int[32], g_diag[16];
static int g_rp = 0;
static void p16_init(void) {
    if (g_rp) return;
    or_p2_params p;
    if (or_p2_lookup(16, &p) != 0) abort();
    for (int i = 0; i < 128; i++) g_ext[i] = to_m(p.ext_rc[i]);
    for (int i = 0; i < p.rounds_p; i++) g_int[i] = to_m(p.int_rc[i]);
    for (int i = 0; i < 16; i++) g_diag[i] = to_m(p.diag[i]);
    g_rp = p.rounds_p;
}
static inline uint32_t pow7(uint32_t x) {
    const uint32_t x2 = mm(x, x), x3 = mm(x2, x), x6 = mm(x3, x3);
    return mm(x6, x);
}
static inline void ext_layer(uint32_t* s) {
    for (int b = 0; b < 16; b += 4) { /* M4 = circ(2, 3, 1, 1) */
        const uint32_t x0 = s[b], x1 = s[b + 1], x2 = s[b + 2], x3 = s[b + 3];
        const uint32_t t01 = madd(x0, x1), t23 = madd(x2, x3), t = madd(t01, t23);
        const uint32_t a = madd(t, x1), c = madd(t, x3);
        s[b] = madd(a, t01);
        s[b + 1] = madd(a, madd(x2, x2));
        s[b + 2] = madd(c, t23);
        s[b + 3] = madd(c, madd(x0, x0));
    }
    uint32_t sums[4];
    for (int k = 0; k < 4; k++) sums[k] = madd(madd(s[k], s[k + 4]), madd(s[k + 8], s[k + 12]));
    for (int i = 0; i < 16; i++) s[i] = madd(s[i], sums[i & 3]);
}
static void perm16(uint32_t* s) {
    ext_layer(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = pow7(madd(s[i], g_ext[r * 16 + i]));
        ext_layer(s);
    }
    for (int r = 0; r < g_rp; r++) {
        s[0] = pow7(madd(s[0], g_int[r]));
        uint32_t sum = 0;
        for (int i = 0; i < 16; i++) sum = madd(sum, s[i]);
        for (int i = 0; i < 16; i++) s[i] = madd(sum, mm(g_diag[i], s[i]));
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 16; i++) s[i] = pow7(madd(s[i], g_ext[r * 16 + i]));
        ext_layer(s);
    }
}

/* ---- the same permutation on VL states at once (structure of arrays: s[i][l] = lane i of state l): every loop over l is a
 * vector loop (the widening multiplies of mm vectorise), which is how a CPU prover's packed field types hash VL rows together */
#define VL 16
typedef uint32_t vstate[16][VL];
#define VLOOP for (int l = 0; l < VL; l++)
__attribute__((target_clones("avx2", "default"))) static void perm16_v_generic(vstate s) {
    uint32_t t01[VL], t23[VL], t[VL], a[VL], c[VL], sums[4][VL];
#define EXT_LAYER_V()                                                          \
    do {                                                                       \
        for (int b = 0; b < 16; b += 4) {                                      \
            VLOOP {                                                            \
                const uint32_t x0 = s[b][l], x1 = s[b + 1][l], x2 = s[b + 2][l], x3 = s[b + 3][l]; \
                t01[l] = madd(x0, x1);                                         \
                t23[l] = madd(x2, x3);                                         \
                t[l] = madd(t01[l], t23[l]);                                   \
                a[l] = madd(t[l], x1);                                         \
                c[l] = madd(t[l], x3);                                         \
                s[b][l] = madd(a[l], t01[l]);                                  \
                s[b + 1][l] = madd(a[l], madd(x2, x2));                        \
                s[b + 2][l] = madd(c[l], t23[l]);                              \
                s[b + 3][l] = madd(c[l], madd(x0, x0));                        \
            }                                                                  \
        }                                                                      \
        for (int k = 0; k < 4; k++) VLOOP sums[k][l] = madd(madd(s[k][l], s[k + 4][l]), madd(s[k + 8][l], s[k + 12][l])); \
        for (int i = 0; i < 16; i++) VLOOP s[i][l] = madd(s[i][l], sums[i & 3][l]); \
    } while (0)
    EXT_LAYER_V();
    for (int r = 0; r < 8; r++) {
        if (r == 4) {
            for (int q = 0; q < g_rp; q++) {
                uint32_t sum[VL];
                const uint32_t rc = g_int[q];
                VLOOP s[0][l] = pow7(madd(s[0][l], rc));
                VLOOP sum[l] = 0;
                for (int i = 0; i < 16; i++) VLOOP sum[l] = madd(sum[l], s[i][l]);
                for (int i = 0; i < 16; i++) {
                    const uint32_t d = g_diag[i];
                    VLOOP s[i][l] = madd(sum[l], mm(d, s[i][l]));
                }
            }
        }
        for (int i = 0; i < 16; i++) {
            const uint32_t rc = g_ext[r * 16 + i];
            VLOOP s[i][l] = pow7(madd(s[i][l], rc));
        }
        EXT_LAYER_V();
    }
#undef EXT_LAYER_V
}

#if defined(__x86_64__)
/* The same permutation with explicit AVX-512: lane i of all sixteen states is one 512-bit register, the Montgomery product is the
 * packed-field routine of the CPU provers (two widening multiplies for the even / odd lanes, the reduction on both, the high
 * halves blended back: p3's monty-31 AVX-512 multiplication [UPSTREAM-RECALL] restated) -- round 5: the auto-vectorised loop above
 * ran at 1.5 us per permutation and core, this one at a tenth of that.  Picked at run time when the CPU has AVX-512F / DQ. */
#include <immintrin.h>
#define T512 __attribute__((target("avx512f,avx512dq")))
typedef __m512i v16;
static inline T512 v16 v_add(v16 a, v16 b) {
    const v16 p = _mm512_set1_epi32((int)OR_P), t = _mm512_add_epi32(a, b);
    return _mm512_min_epu32(t, _mm512_sub_epi32(t, p));
}
static inline T512 v16 v_mul(v16 a, v16 b) { /* a b 2^-32 mod p per lane, operands and result in [0, p) */
    const v16 p = _mm512_set1_epi32((int)OR_P), mu = _mm512_set1_epi32((int)CP_MU);
    const v16 a_odd = _mm512_srli_epi64(a, 32), b_odd = _mm512_srli_epi64(b, 32);
    const v16 t_evn = _mm512_mul_epu32(a, b), t_odd = _mm512_mul_epu32(a_odd, b_odd);            /* 64-bit products */
    const v16 m_evn = _mm512_mul_epu32(t_evn, mu), m_odd = _mm512_mul_epu32(t_odd, mu);            /* low words: t * mu mod 2^32 */
    const v16 u_evn = _mm512_mul_epu32(m_evn, p), u_odd = _mm512_mul_epu32(m_odd, p);              /* m * p: high words are u */
    const v16 hi = _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(t_evn, 32), t_odd);           /* high words of t, lane by lane */
    const v16 u = _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(u_evn, 32), u_odd);
    const v16 r = _mm512_sub_epi32(hi, u);                                                         /* hi - u, + p when it borrowed */
    return _mm512_min_epu32(r, _mm512_add_epi32(r, p));
}
static inline T512 v16 v_pow7(v16 x) {
    const v16 x2 = v_mul(x, x), x3 = v_mul(x2, x), x6 = v_mul(x3, x3);
    return v_mul(x6, x);
}
static T512 void ext_layer_512(v16* s) {
    for (int b = 0; b < 16; b += 4) {
        const v16 x0 = s[b], x1 = s[b + 1], x2 = s[b + 2], x3 = s[b + 3];
        const v16 t01 = v_add(x0, x1), t23 = v_add(x2, x3), t = v_add(t01, t23);
        const v16 a = v_add(t, x1), c = v_add(t, x3);
        s[b] = v_add(a, t01);
        s[b + 1] = v_add(a, v_add(x2, x2));
        s[b + 2] = v_add(c, t23);
        s[b + 3] = v_add(c, v_add(x0, x0));
    }
    v16 sums[4];
    for (int k = 0; k < 4; k++) sums[k] = v_add(v_add(s[k], s[k + 4]), v_add(s[k + 8], s[k + 12]));
    for (int i = 0; i < 16; i++) s[i] = v_add(s[i], sums[i & 3]);
}
static T512 void perm16_v_avx512(vstate st) {
    v16 s[16];
    for (int i = 0; i < 16; i++) s[i] = _mm512_loadu_si512((const void*)st[i]);
    ext_layer_512(s);
    for (int r = 0; r < 8; r++) {
        if (r == 4) {
            for (int q = 0; q < g_rp; q++) {
                s[0] = v_pow7(v_add(s[0], _mm512_set1_epi32((int)g_int[q])));
                v16 sum = s[0];
                for (int i = 1; i < 16; i++) sum = v_add(sum, s[i]);
                for (int i = 0; i < 16; i++) s[i] = v_add(sum, v_mul(_mm512_set1_epi32((int)g_diag[i]), s[i]));
            }
        }
        for (int i = 0; i < 16; i++) s[i] = v_pow7(v_add(s[i], _mm512_set1_epi32((int)g_ext[r * 16 + i])));
        ext_layer_512(s);
    }
    for (int i = 0; i < 16; i++) _mm512_storeu_si512((void*)st[i], s[i]);
}
static int g_have_512 = -1;
static void perm16_v(vstate s) {
    if (g_have_512 < 0) g_have_512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && getenv("LURK_ORACLE_NO_AVX512") == NULL;
    if (g_have_512) perm16_v_avx512(s);
    else perm16_v_generic(s);
}
#else
static void perm16_v(vstate s) { perm16_v_generic(s); }
#endif

/* ---- BASELINE config 2 on the CPU: the batched hash of the Lurk chipset (/root/reference/src/core/poseidon.rs:30-38,61-63;
 * /root/reference/src/core/zstore.rs:241-248) -- digest = the first 8 lanes of the width-W permutation of the W-lane preimage -- for
 * W = 16 .. 48, sixteen states at a time (lane l of the sixteen states = one 512-bit register per state word; round structure of
 * oracle/poseidon2.c: permute_rec, the tables of or_p2_lookup), OpenMP over blocks of sixteen rows.  bench.py --workload poseidon2
 * times it as the "port" baseline; tests/test_cpu_port.py holds it against the scalar oracle.  Canonical words in and out.
 * Returns 0, or -1 for an unknown width / a CPU without AVX-512 (the caller falls back to the scalar oracle). */
#if defined(__x86_64__)
#define P2_MAX_W 48
static T512 void ext_layer_w(v16* s, int w) {
    for (int b = 0; b < w; b += 4) {
        const v16 x0 = s[b], x1 = s[b + 1], x2 = s[b + 2], x3 = s[b + 3];
        const v16 t01 = v_add(x0, x1), t23 = v_add(x2, x3), t = v_add(t01, t23);
        const v16 a = v_add(t, x1), c = v_add(t, x3);
        s[b] = v_add(a, t01);
        s[b + 1] = v_add(a, v_add(x2, x2));
        s[b + 2] = v_add(c, t23);
        s[b + 3] = v_add(c, v_add(x0, x0));
    }
    v16 sums[4] = {s[0], s[1], s[2], s[3]};
    for (int i = 4; i < w; i++) sums[i & 3] = v_add(sums[i & 3], s[i]);
    for (int i = 0; i < w; i++) s[i] = v_add(s[i], sums[i & 3]);
}
static T512 void hash8_block16(int w, int rp, const uint32_t* ext_m, const uint32_t* int_m, const uint32_t* diag_m, const uint32_t* in, uint32_t* out, size_t rows) {
    v16 s[P2_MAX_W];
    const v16 r2 = _mm512_set1_epi32((int)CP_R2), one = _mm512_set1_epi32(1);
    /* row l of the block -> lane l: gather word i of each of the (up to) sixteen rows */
    uint32_t idx[16];
    for (int l = 0; l < 16; l++) idx[l] = (uint32_t)((size_t)l < rows ? l : 0) * (uint32_t)w;
    const v16 vidx = _mm512_loadu_si512((const void*)idx);
    for (int i = 0; i < w; i++) s[i] = v_mul(_mm512_i32gather_epi32(vidx, (const void*)(in + i), 4), r2); /* to Montgomery */
    ext_layer_w(s, w);
    for (int r = 0; r < 8; r++) {
        if (r == 4) {
            for (int q = 0; q < rp; q++) {
                s[0] = v_pow7(v_add(s[0], _mm512_set1_epi32((int)int_m[q])));
                v16 sum = s[0];
                for (int i = 1; i < w; i++) sum = v_add(sum, s[i]);
                for (int i = 0; i < w; i++) s[i] = v_add(sum, v_mul(_mm512_set1_epi32((int)diag_m[i]), s[i]));
            }
        }
        for (int i = 0; i < w; i++) s[i] = v_pow7(v_add(s[i], _mm512_set1_epi32((int)ext_m[r * w + i])));
        ext_layer_w(s, w);
    }
    uint32_t lanes[8][16];
    for (int i = 0; i < 8; i++) _mm512_storeu_si512((void*)lanes[i], v_mul(s[i], one)); /* from Montgomery */
    for (size_t l = 0; l < rows && l < 16; l++)
        for (int i = 0; i < 8; i++) out[l * 8 + (size_t)i] = lanes[i][l];
}
int cp_p2_hash8(int width, size_t n, const uint32_t* in, uint32_t* out) {
    or_p2_params p;
    if (width < 4 || width > P2_MAX_W || (width & 3) || or_p2_lookup(width, &p) != 0) return -1;
    if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"))) return -1;
    uint32_t ext_m[8 * P2_MAX_W], int_m[64], diag_m[P2_MAX_W];
    if (p.rounds_p > 64) return -1;
    for (int i = 0; i < 8 * width; i++) ext_m[i] = to_m(p.ext_rc[i]);
    for (int i = 0; i < p.rounds_p; i++) int_m[i] = to_m(p.int_rc[i]);
    for (int i = 0; i < width; i++) diag_m[i] = to_m(p.diag[i]);
    const long blocks = (long)((n + 15) / 16);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < blocks; b++) {
        const size_t r0 = (size_t)b * 16;
        hash8_block16(width, p.rounds_p, ext_m, int_m, diag_m, in + r0 * (size_t)width, out + r0 * 8, n - r0);
    }
    return 0;
}
#else
int cp_p2_hash8(int width, size_t n, const uint32_t* in, uint32_t* out) {
    (void)width, (void)n, (void)in, (void)out;
    return -1;
}
#endif

/* the sponges of rows row0 .. row0 + VL of the matrices which[0..nw), VL at a time */
static void sponge_v(const uint32_t* const* mats, const uint32_t* widths, const int* which, int nw, size_t row0, uint32_t* out) {
    vstate s;
    memset(s, 0, sizeof s);
    int pos = 0;
    for (int k = 0; k < nw; k++) {
        const int mi = which[k];
        const size_t w = widths[mi];
        const uint32_t* base = mats[mi] + row0 * w;
        for (uint32_t col = 0; col < w; col++) {
            VLOOP s[pos][l] = base[(size_t)l * w + col];
            if (++pos == 8) {
                perm16_v(s);
                pos = 0;
            }
        }
    }
    if (pos) perm16_v(s);
    VLOOP for (int i = 0; i < 8; i++) out[(size_t)l * 8 + i] = s[i][l];
}
/* parents i0 .. i0 + VL: compress(children), then compress(that, injected digest) when inj != NULL */
static void compress_v(const uint32_t* children, const uint32_t* inj, uint32_t* out) {
    vstate s;
    VLOOP for (int i = 0; i < 16; i++) s[i][l] = children[(size_t)l * 16 + i];
    perm16_v(s);
    if (inj) {
        VLOOP for (int i = 0; i < 8; i++) s[8 + i][l] = inj[(size_t)l * 8 + i];
        perm16_v(s);
    }
    VLOOP for (int i = 0; i < 8; i++) out[(size_t)l * 8 + i] = s[i][l];
}

/* overwrite-mode sponge over the concatenated rows `row` of the matrices `which[0..nw)` */
static void sponge(const uint32_t* const* mats, const uint32_t* widths, const int* which, int nw, size_t row, uint32_t* out) {
    uint32_t s[16] = {0};
    int pos = 0;
    for (int k = 0; k < nw; k++) {
        const int mi = which[k];
        const uint32_t* r = mats[mi] + row * widths[mi];
        for (uint32_t c = 0; c < widths[mi]; c++) {
            s[pos++] = r[c];
            if (pos == 8) {
                perm16(s);
                pos = 0;
            }
        }
    }
    if (pos) perm16(s);
    memcpy(out, s, 32);
}
static void compress(const uint32_t* l, const uint32_t* r, uint32_t* out) {
    uint32_t s[16];
    memcpy(s, l, 32);
    memcpy(s + 8, r, 32);
    perm16(s);
    memcpy(out, s, 32);
}

/* Merkle root (canonical) over LDE matrices in Montgomery form, heights 2^log_h[i]; the structure of commit.c: or_merkle_commit */
static int merkle_root(int n_mats, const uint32_t* const* mats, const uint32_t* log_h, const uint32_t* widths, uint32_t* root) {
    p16_init();
    int log_max = 0;
    for (int i = 0; i < n_mats; i++)
        if ((int)log_h[i] > log_max) log_max = (int)log_h[i];
    const size_t n_leaves = (size_t)1 << log_max;
    uint32_t* a = malloc(n_leaves * 8 * sizeof(uint32_t));
    uint32_t* b = malloc((n_leaves / 2 + 1) * 8 * sizeof(uint32_t));
    int* which = malloc(sizeof(int) * (size_t)n_mats);
    if (!a || !b || !which) {
        free(a), free(b), free(which);
        return -1;
    }
    int nw = 0;
    for (int i = 0; i < n_mats; i++)
        if ((int)log_h[i] == log_max) which[nw++] = i;
    if (n_leaves >= VL) {
#pragma omp parallel for schedule(dynamic, 16)
        for (size_t r = 0; r < n_leaves; r += VL) sponge_v(mats, widths, which, nw, r, a + r * 8);
    } else {
        for (size_t r = 0; r < n_leaves; r++) sponge(mats, widths, which, nw, r, a + r * 8);
    }
    uint32_t *prev = a, *cur = b;
    for (int l = 1; l <= log_max; l++) {
        const size_t n_par = n_leaves >> l;
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
        uint32_t* t = prev;
        prev = cur;
        cur = t;
    }
    for (int k = 0; k < 8; k++) root[k] = from_m(prev[k]);
    free(a), free(b), free(which);
    return 0;
}

/* OpenMP team size of the port (a container's CPU quota is usually far below the host's core count) */
#include <omp.h>
void cp_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

/* One commitment round: LDE of every matrix (canonical n_i x w_i inputs) and the tree over the LDEs; root canonical. */
int cp_commit_round(int n_mats, const uint32_t* const* mats, const uint32_t* log_n, const uint32_t* widths, int log_blowup, uint32_t* root) {
    uint32_t** lde = calloc((size_t)n_mats, sizeof(uint32_t*));
    uint32_t* log_h = malloc(sizeof(uint32_t) * (size_t)n_mats);
    int rc = lde && log_h ? 0 : -1;
    for (int i = 0; i < n_mats && rc == 0; i++) {
        log_h[i] = log_n[i] + (uint32_t)log_blowup;
        lde[i] = malloc(((size_t)widths[i] << log_h[i]) * sizeof(uint32_t));
        rc = lde[i] ? cp_lde((int)log_n[i], (int)widths[i], log_blowup, mats[i], lde[i], 1) : -1;
    }
    if (rc == 0) rc = merkle_root(n_mats, (const uint32_t* const*)lde, log_h, widths, root);
    if (lde)
        for (int i = 0; i < n_mats; i++) free(lde[i]);
    free(lde), free(log_h);
    return rc;
}
