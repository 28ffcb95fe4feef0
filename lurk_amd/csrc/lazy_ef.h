// Lazily reduced extension-field accumulator shared by the AIR sinks (stark.hip) and the opening kernels (fri.hip).
#pragma once
#include "babybear.h"

namespace lurkhip {

using bb::ef;

// Extension-field accumulator for sums of (lane value) x (wave-uniform extension constant): four 64-bit lanes holding
// R * value, fed by one v_mad_i64_i32 per coefficient and term and reduced only when the next term would not fit.
// A term is (|v| <= p) x (|w_c| <= p/2) <= p^2 / 2; a freshly folded lane is below 0.08 p^2 and sred needs |t| < 1.2 p^2,
// so two terms fit between folds: 4 multiply-adds + 2 fold multiply-adds per term against 36 instructions for scale + add in
// canonical form.
struct LazyEf {
    int64_t a[4];
    uint32_t room;  // terms that still fit (wave-uniform)
    // a = hi * 2^32 + lo (hi signed, lo unsigned) is congruent to hi * R1 + lo, R1 = 2^32 mod p < 2^28: one multiply-add per lane,
    // and the folded lane is below 2^58.1 = 0.08 p^2 (|hi| < 2^30.1 for |a| < 1.2 p^2).  (Rounds 2-3 folded through a full
    // Montgomery reduction and a product by R1: three multiply-class instructions per lane instead of one.)
    __device__ __forceinline__ void fold() {
#pragma unroll
        for (int c = 0; c < 4; c++) a[c] = bb::mad_i64((int32_t)(a[c] >> 32), (int32_t)bb::R1, (int64_t)(uint64_t)(uint32_t)a[c]);
        room = 2;
    }
    __device__ __forceinline__ void set(const ef& x) {  // canonical x
#pragma unroll
        for (int c = 0; c < 4; c++) a[c] = bb::mad_i64((int32_t)x.c[c], (int32_t)bb::R1, 0);
        room = 2;
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int c = 0; c < 4; c++) a[c] = 0;
        room = 2;
    }
    // += v * w, v a canonical lane value, w[0..3] the centred coefficients of a uniform constant
    __device__ __forceinline__ void add_base(uint32_t v, const int32_t (&w)[8]) {
        if (room == 0) fold();
#pragma unroll
        for (int c = 0; c < 4; c++) a[c] = bb::mad_i64_u((int32_t)v, w[c], a[c]);
        room--;
    }
    // += v * w with a per-lane centred w (|w_c| <= p/2)
    __device__ __forceinline__ void add_base_v(uint32_t v, const int32_t (&w)[4]) {
        if (room == 0) fold();
#pragma unroll
        for (int c = 0; c < 4; c++) a[c] = bb::mad_i64((int32_t)v, w[c], a[c]);
        room--;
    }
    // += v * w for a canonical lane extension element v; w[4..6] = 11 * w[1..3]
    __device__ __forceinline__ void add_ext(const ef& v, const int32_t (&w)[8]) {
        if (room < 2) fold();
        const int32_t v0 = (int32_t)v.c[0], v1 = (int32_t)v.c[1], v2 = (int32_t)v.c[2], v3 = (int32_t)v.c[3];
        a[0] = bb::mad_i64_u(v1, w[6], bb::mad_i64_u(v0, w[0], a[0]));
        a[1] = bb::mad_i64_u(v1, w[0], bb::mad_i64_u(v0, w[1], a[1]));
        a[2] = bb::mad_i64_u(v1, w[1], bb::mad_i64_u(v0, w[2], a[2]));
        a[3] = bb::mad_i64_u(v1, w[2], bb::mad_i64_u(v0, w[3], a[3]));
        fold();
        a[0] = bb::mad_i64_u(v3, w[4], bb::mad_i64_u(v2, w[5], a[0]));
        a[1] = bb::mad_i64_u(v3, w[5], bb::mad_i64_u(v2, w[6], a[1]));
        a[2] = bb::mad_i64_u(v3, w[6], bb::mad_i64_u(v2, w[0], a[2]));
        a[3] = bb::mad_i64_u(v3, w[0], bb::mad_i64_u(v2, w[1], a[3]));
        room = 0;
    }
    __device__ __forceinline__ ef value() const {  // canonical
        ef r;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t x = (uint32_t)bb::sred(a[c]);
            r.c[c] = bb::umin(x, x + bb::P);
        }
        return r;
    }
};

// Eight words of a power table (alpha^k / beta^t, written by an earlier kernel: constant for this one) at a wave-uniform
// address.  Read through the constant address space so that the loads may be scalar ones even in a kernel that also stores
// (the permutation rows: there they were 135 vector loads of one address each, behind the stores' possible aliases).
#if defined(LURK_AB_NO_SMEM)  // diagnostic (wrong values, same control flow): what the power tables' scalar loads cost
__device__ __forceinline__ void load_w8(int32_t (&w)[8], const uint32_t* __restrict__ p) {
#pragma unroll
    for (int c = 0; c < 8; c++) w[c] = (int32_t)(uintptr_t)p + c;
}
#elif !defined(LURK_POWER_TABLES_GENERIC)
__device__ __forceinline__ void load_w8(int32_t (&w)[8], const uint32_t* __restrict__ p) {
    const __attribute__((address_space(4))) uint32_t* cp = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)p;
#pragma unroll
    for (int c = 0; c < 8; c++) w[c] = (int32_t)cp[c];
}
#else
__device__ __forceinline__ void load_w8(int32_t (&w)[8], const uint32_t* __restrict__ p) {
#pragma unroll
    for (int c = 0; c < 8; c++) w[c] = (int32_t)p[c];
}
#endif

}  // namespace lurkhip
