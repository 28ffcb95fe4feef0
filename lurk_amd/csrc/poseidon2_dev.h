// Poseidon2 over BabyBear, device side (gfx950).
//
// One permutation per lane: the W-lane state lives in VGPRs for the whole
// permutation (W <= 48 registers), every inner loop over the state is fully
// unrolled, the loops over rounds are kept rolled so the kernel body stays a few
// KB (the instruction cache is shared by two CUs), and the round constants are
// read with wave-uniform indices from __constant__ memory, i.e. through the
// scalar cache into SGPRs.  Arithmetic is VALU int32 Montgomery (babybear.h).
//
// Replaces on the reference side (P2/P3/P4 in SURVEY.md section 8a):
//   - p3 Poseidon2::permute as configured by /root/reference/src/poseidon/config.rs:75-94
//   - layer order /root/reference/src/poseidon/wide/trace.rs:12-82
//   - internal layer /root/reference/src/poseidon/config.rs:109-118
//   - Poseidon2Cols row layout /root/reference/src/poseidon/wide/columns.rs:16-32
#pragma once
#include "babybear.h"
#include "p2_params.h"

namespace p2 {

template <int N>
struct MArr {
    uint32_t v[N];
};
template <int N>
constexpr MArr<N> monty_table(const uint32_t (&a)[N]) {
    MArr<N> r{};
    for (int i = 0; i < N; i++) r.v[i] = bb::c_to_monty(a[i]);
    return r;
}

// Per-width parameter block in Montgomery form, placed in __constant__ memory.
template <int W, int RP>
struct Params {
    uint32_t ext_rc[8 * W];
    uint32_t int_rc[RP];
    uint32_t diag[W];
    uint32_t ext_rc_mp[8 * W];  // rc - p (mod 2^32): operands of the signed S-box chain (bb::add_pow7_mp)
    uint32_t int_rc_mp[RP];
    int32_t diag_c[W];          // diag, centred representative in (-p/2, p/2]: multiplier of the lazy internal rounds
};
template <int W, int RP>
constexpr Params<W, RP> make_params(const uint32_t (&ext)[8 * W], const uint32_t (&in)[RP], const uint32_t (&d)[W]) {
    Params<W, RP> p{};
    for (int i = 0; i < 8 * W; i++) p.ext_rc[i] = bb::c_to_monty(ext[i]);
    for (int i = 0; i < RP; i++) p.int_rc[i] = bb::c_to_monty(in[i]);
    for (int i = 0; i < W; i++) p.diag[i] = bb::c_to_monty(d[i]);
    for (int i = 0; i < 8 * W; i++) p.ext_rc_mp[i] = p.ext_rc[i] - bb::P;
    for (int i = 0; i < RP; i++) p.int_rc_mp[i] = p.int_rc[i] - bb::P;
    for (int i = 0; i < W; i++) p.diag_c[i] = p.diag[i] > bb::P / 2 ? (int32_t)(p.diag[i] - bb::P) : (int32_t)p.diag[i];
    return p;
}

#define LURK_P2_WIDTHS(X) X(4, 21) X(8, 12) X(12, 10) X(16, 13) X(20, 18) X(24, 21) X(28, 25) X(32, 30) X(36, 34) X(40, 38) X(44, 42) X(48, 46)

#define LURK_P2_DECL(W, RP) \
    __constant__ const Params<W, RP> kParams##W = make_params<W, RP>(LURK_P2_EXT_RC_##W, LURK_P2_INT_RC_##W, LURK_P2_DIAG_##W);
LURK_P2_WIDTHS(LURK_P2_DECL)
#undef LURK_P2_DECL

template <int W>
struct Cfg;
#define LURK_P2_CFG(W_, RP_)                                                                   \
    template <>                                                                                \
    struct Cfg<W_> {                                                                           \
        static constexpr int W = W_;                                                           \
        static constexpr int RP = RP_;                                                         \
        static constexpr int NUM_COLS = 16 * W_ + W_ + (RP_ - 1) + RP_;                        \
        __device__ __forceinline__ static const Params<W_, RP_>& params() { return kParams##W_; } \
    };
LURK_P2_WIDTHS(LURK_P2_CFG)
#undef LURK_P2_CFG

// M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on one 4-lane chunk, 7 adds + 2 doublings
__device__ __forceinline__ void m4(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
    uint32_t t01 = bb::add(x0, x1);
    uint32_t t23 = bb::add(x2, x3);
    uint32_t t0123 = bb::add(t01, t23);
    uint32_t t01123 = bb::add(t0123, x1);
    uint32_t t01233 = bb::add(t0123, x3);
    uint32_t y3 = bb::add(t01233, bb::dbl(x0));
    uint32_t y1 = bb::add(t01123, bb::dbl(x2));
    uint32_t y0 = bb::add(t01123, t01);
    uint32_t y2 = bb::add(t01233, t23);
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
}

template <int W>
__device__ __forceinline__ void external_layer(uint32_t (&s)[W]) {
#pragma unroll
    for (int i = 0; i < W; i += 4) m4(s[i], s[i + 1], s[i + 2], s[i + 3]);
    // width 4 takes the same arm as the larger widths in p3's Poseidon2ExternalMatrixGeneral: the sums are added there
    // too, i.e. the layer is 2*M4 [UPSTREAM-RECALL; no in-tree vector pins width 4]
    {
        uint32_t sums[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t acc = s[k];
#pragma unroll
            for (int i = k + 4; i < W; i += 4) acc = bb::add(acc, s[i]);
            sums[k] = acc;
        }
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = bb::add(s[i], sums[i & 3]);
    }
}

// x_i <- x_i * diag_i + sum_j x_j with explicit diag pointer (uniform address)
template <int W>
__device__ __forceinline__ void internal_layer(uint32_t (&s)[W], const uint32_t* __restrict__ diag) {
    // two interleaved accumulators halve the dependent-add chain
    uint32_t sa = s[0], sb = s[1];
#pragma unroll
    for (int i = 2; i + 1 < W; i += 2) {
        sa = bb::add(sa, s[i]);
        sb = bb::add(sb, s[i + 1]);
    }
    uint32_t sum = bb::add(sa, sb);
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = bb::add(bb::mul(s[i], diag[i]), sum);
}

// All internal rounds on lazily reduced signed lanes (no witness is recorded, W <= 40).
//
// A lane is an int32 x with |x| < 2^31 standing for x mod p.  With V = R * sum_j x_j held as one 64-bit value, lane i of the
// layer is sred(x_i * d_i + V): one multiply-add, one low product, one multiply-add -- three instructions against nine for
// the canonical mul + add -- and nothing is corrected between rounds.  V is built from groups of at most eight lanes:
// U_g = sum_j x_j * R (64-bit multiply-adds, |U_g| < 8 * 1.05 p * 0.134 p < 2^63), u_g = sred(U_g) = sum_j x_j (mod p),
// V = sum_g u_g * R.  Bounds (R mod p = 0.134 p, |d_i| <= p / 2): |V| < 0.14 p^2 * ceil(W / 8), so for W <= 40
// |x_i * d_i + V| < 1.2 p^2 and the lanes stay below 1.06 p < 2^31 round after round.  Lane 0 is brought to [0, p) before
// its round constant is added; its x^7 stays signed.
// `sum_mult_c`: centred integer multiplier of the lane sum, R mod p for the paper's layer (sum + d_i x_i); a scaled layer
// s (sum + d_i x_i) -- p3's Montgomery-shift diffusion matrix, scale 2^-32 -- passes s R mod p and scaled d_i.  A multiplier
// other than R mod p can be as large as p / 2, so the group sums are combined and reduced once more before it is applied
// (uniform branch: the common case keeps its instruction count).
template <int W>
__device__ __forceinline__ void internal_rounds_lazy(uint32_t (&s)[W], int rounds_p, const uint32_t* __restrict__ int_rc_mp,
                                                     const int32_t* __restrict__ diag_c, int32_t sum_mult_c = (int32_t)bb::R1) {
    static_assert(W <= 40, "lane bound of the lazy internal rounds");
    int32_t x[W];
#pragma unroll
    for (int i = 0; i < W; i++) x[i] = (int32_t)s[i];
#pragma unroll 1
    for (int r = 0; r < rounds_p; r++) {
        {
            const uint32_t c0 = bb::umin((uint32_t)x[0], (uint32_t)x[0] + bb::P);
            const int32_t y = (int32_t)(c0 + int_rc_mp[r]);
            const int32_t y2 = bb::smul(y, y), y3 = bb::smul(y2, y), y6 = bb::smul(y3, y3);
            x[0] = bb::smul(y6, y);
        }
        int64_t v = 0;
        if (sum_mult_c == 1) {
            // Montgomery-shift diffusion layer (scale 2^-32: the multiplier of the lane sum is the integer 1 and the diagonal
            // entries are small integers, powers of two in p3's DiffusionMatrixBabyBear): V is the plain 64-bit sum of the lanes,
            // |V| < W * 1.06 p -- no products by R, no group reductions (uniform branch)
#pragma unroll
            for (int j = 0; j < W; j++) v = bb::mad_i64(x[j], 1, v);
        } else {
#pragma unroll
            for (int g = 0; g < W; g += 8) {
                int64_t u = 0;
#pragma unroll
                for (int j = g; j < g + 8 && j < W; j++) u = bb::mad_i64(x[j], (int32_t)bb::R1, u);
                v = bb::mad_i64(bb::sred(u), (int32_t)bb::R1, v);
            }
            if (sum_mult_c != (int32_t)bb::R1) v = bb::mad_i64(bb::sred(v), sum_mult_c, 0);  // |sred(v)| < 0.7 p, |v| < 0.35 p^2
        }
#pragma unroll
        for (int i = 0; i < W; i++) x[i] = bb::sred(bb::mad_i64_u(x[i], diag_c[i], v));
    }
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = bb::umin((uint32_t)x[i], (uint32_t)x[i] + bb::P);
}

struct NoRecord;
template <class Rec>
struct records_nothing {
    static constexpr bool value = false;
};
template <>
struct records_nothing<NoRecord> {
    static constexpr bool value = true;
};

struct NoRecord {
    __device__ __forceinline__ void ext_state(int, int, uint32_t) {}
    __device__ __forceinline__ void end_ext_state(int) {}
    __device__ __forceinline__ void ext_sbox(int, int, uint32_t) {}
    __device__ __forceinline__ void end_ext_sbox(int) {}
    __device__ __forceinline__ void int_init(int, uint32_t) {}
    __device__ __forceinline__ void end_int_init() {}
    __device__ __forceinline__ void int_state0(int, uint32_t) {}
    __device__ __forceinline__ void int_sbox(int, uint32_t) {}
    __device__ __forceinline__ void end_internal() {}
};

template <int W, class Rec>
__device__ __forceinline__ void external_round(uint32_t (&s)[W], int r, const uint32_t* __restrict__ ext_rc,
                                               const uint32_t* __restrict__ ext_rc_mp, Rec& rec) {
#pragma unroll
    for (int i = 0; i < W; i++) rec.ext_state(r, i, s[i]);
    rec.end_ext_state(r);
    if constexpr (records_nothing<Rec>::value) {
        // nothing observes the intermediates: signed 4-instruction products, one correction per S-box
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = bb::add_pow7_mp(s[i], ext_rc_mp[r * W + i]);
    } else {
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint32_t x = bb::add(s[i], ext_rc[r * W + i]);
            uint32_t x3 = bb::cube(x);
            rec.ext_sbox(r, i, x3);
            s[i] = bb::pow7_from_cube(x, x3);
        }
    }
    rec.end_ext_sbox(r);
    external_layer<W>(s);
}

// The permutation on a Montgomery-form state, parameters given by pointers so the
// same body serves the built-in tables and a caller-supplied width-16 set.
template <int W, class Rec>
__device__ __forceinline__ void permute_core(uint32_t (&s)[W], int rounds_p, const uint32_t* __restrict__ ext_rc,
                                             const uint32_t* __restrict__ int_rc,
                                             const uint32_t* __restrict__ diag, const uint32_t* __restrict__ ext_rc_mp,
                                             const uint32_t* __restrict__ int_rc_mp, const int32_t* __restrict__ diag_c,
                                             Rec& rec, int32_t sum_mult_c = (int32_t)bb::R1) {
    external_layer<W>(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) external_round<W>(s, r, ext_rc, ext_rc_mp, rec);
#pragma unroll
    for (int i = 0; i < W; i++) rec.int_init(i, s[i]);
    rec.end_int_init();
    if constexpr (records_nothing<Rec>::value && W <= 40) {
        internal_rounds_lazy<W>(s, rounds_p, int_rc_mp, diag_c, sum_mult_c);
    } else {
#pragma unroll 1
        for (int r = 0; r < rounds_p; r++) {
            if constexpr (records_nothing<Rec>::value) {
                s[0] = bb::add_pow7_mp(s[0], int_rc_mp[r]);
            } else {
                if (r > 0) rec.int_state0(r - 1, s[0]);
                uint32_t x = bb::add(s[0], int_rc[r]);
                uint32_t x3 = bb::cube(x);
                rec.int_sbox(r, x3);
                s[0] = bb::pow7_from_cube(x, x3);
            }
            internal_layer<W>(s, diag);
        }
    }
    rec.end_internal();
#pragma unroll 1
    for (int r = 4; r < 8; r++) external_round<W>(s, r, ext_rc, ext_rc_mp, rec);
}

template <int W>
__device__ __forceinline__ void permute(uint32_t (&s)[W]) {
    NoRecord rec;
    const auto& p = Cfg<W>::params();
    permute_core<W>(s, Cfg<W>::RP, p.ext_rc, p.int_rc, p.diag, p.ext_rc_mp, p.int_rc_mp, p.diag_c, rec);
}

}  // namespace p2
