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
BB_HD uint32_t mul(uint32_t a, uint32_t b) { return mred((uint64_t)a * b); }
BB_HD uint32_t sqr(uint32_t a) { return mul(a, a); }

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
// Fermat inverse a^(p-2); inv(0) = 0.  p - 2 = 0x77ffffff.
BB_HD uint32_t inv(uint32_t a) {
    // addition chain: a^(2^27-1) then the top nibble 0111 -> exponent 0x77ffffff
    // = 0b0111_0111_1111_1111_1111_1111_1111_1111
    uint32_t a2 = sqr(a);            // 2
    uint32_t a3 = mul(a2, a);        // 3
    uint32_t a6 = sqr(a3);
    uint32_t a7 = mul(a6, a);        // 2^3-1
    uint32_t t = a7;
    // build 2^27 - 1 = 27 ones: (2^3-1) -> 2^6-1 -> 2^12-1 -> 2^24-1 -> 2^27-1
    uint32_t x6 = t;
    for (int i = 0; i < 3; i++) x6 = sqr(x6);
    x6 = mul(x6, a7);                // 2^6-1
    uint32_t x12 = x6;
    for (int i = 0; i < 6; i++) x12 = sqr(x12);
    x12 = mul(x12, x6);              // 2^12-1
    uint32_t x24 = x12;
    for (int i = 0; i < 12; i++) x24 = sqr(x24);
    x24 = mul(x24, x12);             // 2^24-1
    uint32_t x27 = x24;
    for (int i = 0; i < 3; i++) x27 = sqr(x27);
    x27 = mul(x27, a7);              // 2^27-1
    // exponent = 0b111 << 28 | 0 << 27 | (2^27-1)  = 7*2^28 + 2^27 - 1
    uint32_t hi = a7;                // 0b111
    for (int i = 0; i < 28; i++) hi = sqr(hi);
    return mul(hi, x27);
}
static_assert(7u * (1u << 28) + (1u << 27) - 1u == P - 2u, "inverse exponent");

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
    return ef{{mul(a.c[0], s), mul(a.c[1], s), mul(a.c[2], s), mul(a.c[3], s)}};
}
BB_HD ef ef_add_base(const ef& a, uint32_t s) { return ef{{add(a.c[0], s), a.c[1], a.c[2], a.c[3]}}; }
BB_HD ef ef_mul(const ef& a, const ef& b) {
    // c_k = sum_{i+j=k} a_i b_j + 11 * sum_{i+j=k+4} a_i b_j.  With b'_j = 11 b_j every coefficient is a sum of four
    // products of reduced operands; two of them are < 2 p^2 < p * 2^32, the Montgomery-reduction bound, so each
    // coefficient costs four multiply-adds, two reductions and one modular add (no 64-bit comparisons).
    const uint32_t w1 = mul(EXT_W_M, b.c[1]), w2 = mul(EXT_W_M, b.c[2]), w3 = mul(EXT_W_M, b.c[3]);
    auto dot2 = [](uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) -> uint32_t {
        return mred((uint64_t)x0 * y0 + (uint64_t)x1 * y1);
    };
    return ef{{add(dot2(a.c[0], b.c[0], a.c[1], w3), dot2(a.c[2], w2, a.c[3], w1)),
               add(dot2(a.c[0], b.c[1], a.c[1], b.c[0]), dot2(a.c[2], w3, a.c[3], w2)),
               add(dot2(a.c[0], b.c[2], a.c[1], b.c[1]), dot2(a.c[2], b.c[0], a.c[3], w3)),
               add(dot2(a.c[0], b.c[3], a.c[1], b.c[2]), dot2(a.c[2], b.c[1], a.c[3], b.c[0]))}};
}
BB_HD ef ef_sqr(const ef& a) { return ef_mul(a, a); }
BB_HD bool ef_is_zero(const ef& a) { return (a.c[0] | a.c[1] | a.c[2] | a.c[3]) == 0; }
// inverse by two conjugations down to the base field (x -> -x, then x^2 -> -x^2)
BB_HD ef ef_inv(const ef& a) {
    ef a1{{a.c[0], neg(a.c[1]), a.c[2], neg(a.c[3])}};
    ef b = ef_mul(a, a1);  // in span{1, x^2}
    ef b1{{b.c[0], 0, neg(b.c[2]), 0}};
    ef n = ef_mul(b, b1);  // in F
    uint32_t ninv = inv(n.c[0]);
    return ef_scale(ef_mul(a1, b1), ninv);
}

}  // namespace bb
