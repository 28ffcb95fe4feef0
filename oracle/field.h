/* ORACLE (test infrastructure, not product code).
 *
 * BabyBear field arithmetic on canonical representatives, written the slow,
 * obviously-correct way (64-bit products and `%`).  It deliberately shares no
 * code with lurk_amd/csrc/babybear.h (the Montgomery implementation the HIP
 * kernels use), so that the two can be checked against each other.
 *
 * Field: p = 2013265921 = 15 * 2^27 + 1 (third-party p3_baby_bear::BabyBear;
 * used throughout the reference, e.g. /root/reference/src/lair/execute.rs:631-640
 * and /root/reference/src/air/builder.rs:159-168).
 * Extension: F[x]/(x^4 - 11) (p3 BinomialExtensionField<BabyBear,4>,
 * [UPSTREAM-RECALL] for the non-residue 11).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 */
#ifndef LURK_ORACLE_FIELD_H
#define LURK_ORACLE_FIELD_H
#include <stdint.h>

#define OR_P 2013265921u

static inline uint32_t or_add(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % OR_P); }
static inline uint32_t or_sub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + OR_P - b) % OR_P); }
static inline uint32_t or_mul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % OR_P); }
static inline uint32_t or_neg(uint32_t a) { return a ? OR_P - a : 0; }

static inline uint32_t or_pow(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = or_mul(r, a);
        a = or_mul(a, a);
        e >>= 1;
    }
    return r;
}
/* Fermat inverse; inverse(0) is defined as 0 here (p3 panics on it). */
static inline uint32_t or_inv(uint32_t a) { return or_pow(a, OR_P - 2); }

/* quartic extension, basis 1,x,x^2,x^3 with x^4 = 11 */
#define OR_W 11u
typedef struct { uint32_t c[4]; } or_ef;

static inline or_ef or_ef_from(uint32_t a) { or_ef r = {{a, 0, 0, 0}}; return r; }
static inline or_ef or_ef_add(or_ef a, or_ef b) {
    or_ef r; for (int i = 0; i < 4; i++) r.c[i] = or_add(a.c[i], b.c[i]); return r;
}
static inline or_ef or_ef_sub(or_ef a, or_ef b) {
    or_ef r; for (int i = 0; i < 4; i++) r.c[i] = or_sub(a.c[i], b.c[i]); return r;
}
static inline or_ef or_ef_mul(or_ef a, or_ef b) {
    uint32_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t[i + j] = or_add(t[i + j], or_mul(a.c[i], b.c[j]));
    or_ef r;
    for (int i = 0; i < 4; i++) r.c[i] = t[i];
    for (int i = 4; i < 7; i++) r.c[i - 4] = or_add(r.c[i - 4], or_mul(OR_W, t[i]));
    return r;
}
static inline or_ef or_ef_scale(or_ef a, uint32_t s) {
    or_ef r; for (int i = 0; i < 4; i++) r.c[i] = or_mul(a.c[i], s); return r;
}
static inline int or_ef_is_zero(or_ef a) { return !(a.c[0] | a.c[1] | a.c[2] | a.c[3]); }
static inline or_ef or_ef_pow(or_ef a, uint64_t e) {
    or_ef r = or_ef_from(1);
    while (e) {
        if (e & 1) r = or_ef_mul(r, a);
        a = or_ef_mul(a, a);
        e >>= 1;
    }
    return r;
}
/* inverse via the norm to the base field: a^-1 = a^(p+p^2+p^3) / N(a), computed
 * the slow way as a^(p^4-2) split in two 64-bit exponent pieces. */
static inline or_ef or_ef_inv(or_ef a) {
    /* a^(p^4 - 2) = (a^(p^2))^(p^2) * ... would need 124-bit exponent; use
     * Frobenius-free route: solve via conjugates.  a * a' with a' = a(-x) gives
     * an element of F[x^2]; repeat once to land in F. */
    or_ef a1 = {{a.c[0], or_neg(a.c[1]), a.c[2], or_neg(a.c[3])}};     /* x -> -x */
    or_ef b = or_ef_mul(a, a1);                                        /* in span{1, x^2} */
    or_ef b1 = {{b.c[0], 0, or_neg(b.c[2]), 0}};                       /* x^2 -> -x^2 */
    or_ef n = or_ef_mul(b, b1);                                        /* in F */
    uint32_t ninv = or_inv(n.c[0]);
    return or_ef_scale(or_ef_mul(a1, b1), ninv);
}
#endif
