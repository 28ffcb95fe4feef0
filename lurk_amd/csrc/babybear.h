// BabyBear field arithmetic for the gfx950 kernels (and the host code that feeds them).
//
// p = 2013265921 = 15 * 2^27 + 1.  Device code keeps every element in Montgomery
// form x~ = x * 2^32 mod p, canonical range [0, p), which is also the in-memory
// form of the reference's field type (p3_baby_bear::BabyBear, [UPSTREAM-RECALL]),
// so a caller holding a RowMajorMatrix<BabyBear> can pass its storage with
// repr = LURKHIP_REPR_MONTY and no conversion happens at all.
//
// Replaces: third-party p3_baby_bear (+, -, *, inverse) as used by the reference at
// e.g. /root/reference/src/lair/execute.rs:631-640, /root/reference/src/air/builder.rs:159-168.
//
// No MFMA here: the products are element-wise 31x31-bit, there is no shared
// operand to turn them into a matrix product.  One Montgomery product is
// 2 x v_mul_lo_u32 + 2 x v_mul_hi_u32 (or v_mad_u64_u32) + 3 full-rate VALU ops.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD inline
#endif

namespace bb {

constexpr uint32_t P = 0x78000001u;        // 2013265921
constexpr uint32_t MU = 0x88000001u;       // p^-1 mod 2^32
constexpr uint32_t R1 = 0x0ffffffeu;       // 2^32 mod p  (Montgomery 1)
constexpr uint32_t R2 = 1172168163u;       // 2^64 mod p  (to_monty multiplier)
constexpr uint32_t GEN = 31u;              // multiplicative generator (canonical)
constexpr int TWO_ADICITY = 27;
constexpr uint32_t EXT_W = 11u;            // F[x]/(x^4 - 11)

constexpr uint32_t cmulmod(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
constexpr uint32_t c_to_monty(uint32_t x) { return (uint32_t)((((uint64_t)x) << 32) % P); }
static_assert(c_to_monty(1) == R1, "R1");
static_assert(cmulmod(R1, R1) == R2, "R2");
static_assert((uint32_t)(P * MU) == 1u, "MU");

BB_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

BB_HD uint32_t add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;
    return umin(s, s - P);
}
BB_HD uint32_t sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;
    return umin(d, d + P);
}
BB_HD uint32_t neg(uint32_t a) { return sub(0u, a); }
BB_HD uint32_t dbl(uint32_t a) { return add(a, a); }

// Montgomery reduction of t < p * 2^32: returns t * 2^-32 mod p in [0, p)
BB_HD uint32_t mred(uint64_t t) {
    uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo * MU;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t u = __umulhi(m, P);
#else
    uint32_t u = (uint32_t)(((uint64_t)m * P) >> 32);
#endif
    uint32_t r = hi - u;
    return umin(r, r + P);
}

// a * b + c on signed 32-bit factors with a 64-bit addend: one v_mad_i64_i32.  Spelled as inline assembly on the device:
// left to itself the compiler expands a product with a wave-uniform factor into an unsigned multiply-add plus sign fix-ups
// (five instructions instead of one).  mad_i64_u takes the uniform factor straight from a scalar register.
// The instruction also writes a carry-out to a scalar register pair nobody reads.  Declared as an output ("=s") every
// multiply-add defines the same pair, and the compiler -- which cannot look inside inline assembly -- separates any two
// statements whose definitions overlap by a wait state: an `s_nop 0` after every product of the permutations (149 in the
// 1918 static instructions of k_row_sponges).  The hardware needs none (the compiler's own v_mad_u64_u32 sequences reuse one
// carry pair back to back).  So the pair is handed over as an *input* whose value nobody uses -- the program counter, one
// s_getpc_b64 per kernel, hoisted -- and the instruction overwrites it: LURK_MAD_CARRY_DECLARED=1 restores the declared form.
// Round 3 (ADVICE): overwriting an input operand is outside the inline-assembly contract -- the compiler believes the pair
// still holds the program counter; nothing else ever reads that value, which is why it works.  The contract-clean spellings
// were measured again on k_row_sponges (tools/kernel_instr_count.py): "=s", early-clobber "=&s" and an explicit `vcc` clobber
// all come out at 190 s_nop in 2373 static instructions against 64 in 2249 (the hazard recogniser gives any inline assembly
// whose definitions overlap an operand of the next one a wait state, whatever the register class).  The trick therefore stays
// the default and is guarded instead: `make variant` builds the whole library with the declared form
// (liblurkhip_declared.so) and tests/test_mad_ab_gpu.py requires both builds to produce identical hashes, roots and proofs.
#ifndef LURK_MAD_CARRY_DECLARED
#define LURK_MAD_CARRY_DECLARED 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && LURK_MAD_CARRY_DECLARED == 0
#define LURK_MAD_ASM(TEMPLATE, D, ...)                                   \
    do {                                                                 \
        const uint64_t carry_scratch_ = __builtin_amdgcn_s_getpc();      \
        asm(TEMPLATE : "=v"(D) : "s"(carry_scratch_), __VA_ARGS__);      \
    } while (0)
#elif defined(__HIP_DEVICE_COMPILE__) && LURK_MAD_CARRY_DECLARED == 1
#define LURK_MAD_ASM(TEMPLATE, D, ...)                                   \
    do {                                                                 \
        uint64_t carry_;                                                 \
        asm(TEMPLATE : "=v"(D), "=s"(carry_) : __VA_ARGS__);             \
    } while (0)
#elif defined(__HIP_DEVICE_COMPILE__) && LURK_MAD_CARRY_DECLARED == 2
#define LURK_MAD_ASM(TEMPLATE, D, ...)                                   \
    do {                                                                 \
        uint64_t carry_;                                                 \
        asm(TEMPLATE : "=v"(D), "=&s"(carry_) : __VA_ARGS__);            \
    } while (0)
#elif defined(__HIP_DEVICE_COMPILE__) && LURK_MAD_CARRY_DECLARED == 3
#define LURK_MAD_ASM(TEMPLATE, D, ...) asm(TEMPLATE : "=v"(D) : __VA_ARGS__ : "vcc")
#endif
#if LURK_MAD_CARRY_DECLARED == 3
#define LURK_MAD_T0 "v_mad_i64_i32 %0, vcc, %1, %2, 0"
#define LURK_MAD_T1 "v_mad_i64_i32 %0, vcc, %1, %2, %3"
#else
#define LURK_MAD_T0 "v_mad_i64_i32 %0, %1, %2, %3, 0"
#define LURK_MAD_T1 "v_mad_i64_i32 %0, %1, %2, %3, %4"
#endif
BB_HD int64_t mad_i64(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    // a plain product takes the inline constant 0: a zero held in a VGPR pair costs the 64-bit operand read (measured
    // 5.0 against 4.3 cycles per wave-instruction, tools/ubench_issue.hip)
    if (__builtin_constant_p(c) && c == 0)
        LURK_MAD_ASM(LURK_MAD_T0, d, "v"(a), "v"(b));
    else
        LURK_MAD_ASM(LURK_MAD_T1, d, "v"(a), "v"(b), "v"(c));
    return d;
#else
    return (int64_t)a * b + c;
#endif
}
BB_HD int64_t mad_i64_u(int32_t a, int32_t b_uniform, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    if (__builtin_constant_p(c) && c == 0)
        LURK_MAD_ASM(LURK_MAD_T0, d, "v"(a), "s"(b_uniform));
    else
        LURK_MAD_ASM(LURK_MAD_T1, d, "v"(a), "s"(b_uniform), "v"(c));
    return d;
#else
    return (int64_t)a * b_uniform + c;
#endif
}
// Signed Montgomery reduction without the final correction: for |t| < 2^62.9 returns r = t * 2^-32 (mod p) with
// |r| <= |t| / 2^32 + p / 2.  m = t * p^-1 mod 2^32 (signed), and t - m * p has a zero low word, so the high word of one
// 64-bit multiply-add is the result: v_mul_lo_u32 + v_mad_i64_i32, no separate high product and subtraction.
BB_HD int32_t sred(int64_t t) {
    const int32_t m = (int32_t)((uint32_t)t * MU);
    const int64_t t2 = mad_i64(m, -(int32_t)P, t);
    return (int32_t)(t2 >> 32);
}
// Signed Montgomery product: operands in (-p, p) give a result in (-p, p) again; three instructions on gfx950
// (v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32) against six for the canonical-range product.  Used for the x^7 chains of
// Poseidon2 and the NTT butterflies, whose intermediates never leave the chain.
BB_HD int32_t smul(int32_t a, int32_t b) { return sred(mad_i64(a, b, 0)); }
// canonical representative of a signed lane value in (-p, p)
BB_HD uint32_t canon(int32_t r) {
    const uint32_t u = (uint32_t)r;
    return umin(u, u + P);
}
BB_HD uint32_t mul(uint32_t a, uint32_t b) { return mred((uint64_t)a * b); }
// The same canonical-range product as the signed product and one correction: five instructions on the device against seven
// (64-bit product + the wait state the compiler puts behind it, low product, high product, subtract, correct).  Opt-in (the
// compiled AIR and trace kernels use it for their constraint products): inside divergent control flow the compiler may move the
// multiply-add's scalar carry operand (LURK_MAD_ASM) into vector registers, which does not assemble -- `mul` stays the default.
BB_HD uint32_t mul_s(uint32_t a, uint32_t b) { return canon(smul((int32_t)a, (int32_t)b)); }
BB_HD uint32_t sqr(uint32_t a) { return mul(a, a); }
// (s + rc)^7 for canonical-range s, rc, given rc_mp = rc - p (mod 2^32): the sum s + rc_mp lies in (-p, p), the chain
// runs on signed values, one correction at the end
BB_HD uint32_t add_pow7_mp(uint32_t s, uint32_t rc_mp) {
    const int32_t x = (int32_t)(s + rc_mp);
    const int32_t x2 = smul(x, x), x3 = smul(x2, x), x6 = smul(x3, x3), x7 = smul(x6, x);
    const uint32_t r = (uint32_t)x7;
    return umin(r, r + P);
}
BB_HD uint32_t add_pow7(uint32_t s, uint32_t rc) {
    const int32_t x = (int32_t)(s + (rc - P));
    const int32_t x2 = smul(x, x), x3 = smul(x2, x), x6 = smul(x3, x3), x7 = smul(x6, x);
    const uint32_t r = (uint32_t)x7;
    return umin(r, r + P);
}

BB_HD uint32_t to_monty(uint32_t x) { return mul(x, R2); }
BB_HD uint32_t from_monty(uint32_t x) { return mred((uint64_t)x); }

// x^7 the way the reference's witness generator computes it (cube, then x * cube^2):
// /root/reference/src/poseidon/wide/trace.rs:44-50
BB_HD uint32_t cube(uint32_t x) { return mul(sqr(x), x); }
BB_HD uint32_t pow7_from_cube(uint32_t x, uint32_t x3) { return mul(x, sqr(x3)); }

// a^e, a in Montgomery form
BB_HD uint32_t pow(uint32_t a, uint32_t e) {
    uint32_t r = R1;
    while (e) {
        if (e & 1u) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}
// centred representative (|r| <= (p - 1) / 2) of a canonical word
BB_HD int32_t centre(uint32_t x) { return (int32_t)(x - (x > (P - 1u) / 2u ? P : 0u)); }
// Fermat inverse a^(p-2) of a signed lane value in (-p, p), result in (-p, p); 0 -> 0.  p - 2 = 0x77ffffff = 0b111_0111 followed
// by 24 ones: a^7, four squarings and a^7 again give the prefix, then eight times "three squarings, times a^7" -- 30 squarings
// and 11 products, all three-instruction signed products (round 4; the canonical-range chain before it took 54 + 8 products
// of seven instructions each and was 60 % of the instructions of the LogUp permutation kernel).
BB_HD int32_t inv_s(int32_t a) {
    const int32_t a2 = smul(a, a), a3 = smul(a2, a), a6 = smul(a3, a3), a7 = smul(a6, a);
    int32_t t = a7;
    for (int i = 0; i < 4; i++) t = smul(t, t);
    t = smul(t, a7);  // 0b1110111
    for (int g = 0; g < 8; g++) {
        t = smul(t, t);
        t = smul(t, t);
        t = smul(t, t);
        t = smul(t, a7);
    }
    return t;
}
BB_HD uint32_t inv(uint32_t a) { return canon(inv_s((int32_t)a)); }
static_assert((0x77u << 24) + 0xffffffu == P - 2u, "inverse exponent");

// ---------------------------------------------------------------------------
// quartic extension F[x]/(x^4 - 11), coefficients in Montgomery form
// (p3 BinomialExtensionField<BabyBear,4>; [UPSTREAM-RECALL] for W = 11)
struct ef {
    uint32_t c[4];
};
constexpr uint32_t EXT_W_M = c_to_monty(EXT_W);

BB_HD ef ef_zero() { return ef{{0, 0, 0, 0}}; }
BB_HD ef ef_one() { return ef{{R1, 0, 0, 0}}; }
BB_HD ef ef_from_base(uint32_t a) { return ef{{a, 0, 0, 0}}; }
BB_HD ef ef_add(const ef& a, const ef& b) {
    return ef{{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}};
}
BB_HD ef ef_sub(const ef& a, const ef& b) {
    return ef{{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}};
}
BB_HD ef ef_scale(const ef& a, uint32_t s) {
    return ef{{canon(smul((int32_t)a.c[0], (int32_t)s)), canon(smul((int32_t)a.c[1], (int32_t)s)), canon(smul((int32_t)a.c[2], (int32_t)s)),
               canon(smul((int32_t)a.c[3], (int32_t)s))}};
}
BB_HD ef ef_add_base(const ef& a, uint32_t s) { return ef{{add(a.c[0], s), a.c[1], a.c[2], a.c[3]}}; }
// Extension products on signed lanes (round 4).  With both operands centred (|.| <= p/2) a coefficient
//   c_k = sum_{i+j=k} a_i b_j + 11 * sum_{i+j=k+4} a_i b_j
// is one 64-bit chain: the wrapped-around products are summed and reduced first (h, |h| < 0.86 p), then h * 11~ (11~ the centred
// Montgomery form of 11, 0.47 p) joins the direct products: every chain stays below 1.06 p^2 -- inside sred's 1.2 p^2 and small
// enough that the result is in (-p, p) again.  16 + 3 multiply-adds and 7 two-instruction reductions: 33 instructions, against 93
// for the canonical-range version (three products by 11, eight two-term dot products with their own reductions, four modular adds).
struct sef {
    int32_t c[4];
};
constexpr int32_t EXT_W_MC = EXT_W_M > P / 2u ? (int32_t)(EXT_W_M - P) : (int32_t)EXT_W_M;
BB_HD sef ef_centre(const ef& a) { return sef{{centre(a.c[0]), centre(a.c[1]), centre(a.c[2]), centre(a.c[3])}}; }
BB_HD ef ef_canon(const sef& a) { return ef{{canon(a.c[0]), canon(a.c[1]), canon(a.c[2]), canon(a.c[3])}}; }
// a, b centred; the coefficients of a * b as signed lane values in (-p, p)
BB_HD int32_t sef_mul_c0(const sef& a, const sef& b) {
    const int32_t h = sred(mad_i64(a.c[1], b.c[3], mad_i64(a.c[2], b.c[2], mad_i64(a.c[3], b.c[1], 0))));
    return sred(mad_i64(a.c[0], b.c[0], mad_i64_u(h, EXT_W_MC, 0)));
}
BB_HD int32_t sef_mul_c1(const sef& a, const sef& b) {
    const int32_t h = sred(mad_i64(a.c[2], b.c[3], mad_i64(a.c[3], b.c[2], 0)));
    return sred(mad_i64(a.c[0], b.c[1], mad_i64(a.c[1], b.c[0], mad_i64_u(h, EXT_W_MC, 0))));
}
BB_HD int32_t sef_mul_c2(const sef& a, const sef& b) {
    const int32_t h = sred(mad_i64(a.c[3], b.c[3], 0));
    return sred(mad_i64(a.c[0], b.c[2], mad_i64(a.c[1], b.c[1], mad_i64(a.c[2], b.c[0], mad_i64_u(h, EXT_W_MC, 0)))));
}
BB_HD int32_t sef_mul_c3(const sef& a, const sef& b) {
    return sred(mad_i64(a.c[0], b.c[3], mad_i64(a.c[1], b.c[2], mad_i64(a.c[2], b.c[1], mad_i64(a.c[3], b.c[0], 0)))));
}
BB_HD sef sef_mul(const sef& a, const sef& b) { return sef{{sef_mul_c0(a, b), sef_mul_c1(a, b), sef_mul_c2(a, b), sef_mul_c3(a, b)}}; }
BB_HD ef ef_mul(const ef& a, const ef& b) { return ef_canon(sef_mul(ef_centre(a), ef_centre(b))); }
BB_HD ef ef_sqr(const ef& a) { return ef_mul(a, a); }
BB_HD bool ef_is_zero(const ef& a) { return (a.c[0] | a.c[1] | a.c[2] | a.c[3]) == 0; }
// inverse by two conjugations down to the base field (x -> -x, then x^2 -> -x^2): with a' = a(-x), b = a a' in span{1, x^2},
// b' = b(-x^2) and n = b b' in F, 1/a = a' b' / n.  Signed lanes throughout: only the coefficients that are not zero by
// construction are computed, the base inverse is the 41-product signed chain (inv_s), one correction per output word.
BB_HD ef ef_inv(const ef& a) {
    const sef x = ef_centre(a);
    const sef x1{{x.c[0], -x.c[1], x.c[2], -x.c[3]}};
    const int32_t b0 = centre(canon(sef_mul_c0(x, x1))), b2 = centre(canon(sef_mul_c2(x, x1)));
    // n = b0^2 - 11 b2^2
    const int32_t n = sred(mad_i64(b0, b0, mad_i64_u(sred(mad_i64(b2, -b2, 0)), EXT_W_MC, 0)));
    const int32_t ninv = inv_s(n);
    // a' * (b0 - b2 x^2)
    const int32_t t0 = sred(mad_i64(x1.c[0], b0, mad_i64_u(sred(mad_i64(x1.c[2], -b2, 0)), EXT_W_MC, 0)));
    const int32_t t1 = sred(mad_i64(x1.c[1], b0, mad_i64_u(sred(mad_i64(x1.c[3], -b2, 0)), EXT_W_MC, 0)));
    const int32_t t2 = sred(mad_i64(x1.c[2], b0, mad_i64(x1.c[0], -b2, 0)));
    const int32_t t3 = sred(mad_i64(x1.c[3], b0, mad_i64(x1.c[1], -b2, 0)));
    return ef{{canon(smul(t0, ninv)), canon(smul(t1, ninv)), canon(smul(t2, ninv)), canon(smul(t3, ninv))}};
}

}  // namespace bb
