// Coset LDE of a HEIGHT GROUP of trace matrices (round 4): three kernels, nine matrix transfers, dense 32-column tiles.
//
// Replaces (S1 commit in SURVEY.md 8a; third-party, source absent from /root/reference):
//   p3 TwoAdicFriPcs::commit -> Radix2DitParallel::coset_lde_batch(evals, log_blowup = 1, shift) + bit_reverse_rows
//   [UPSTREAM-RECALL, Plonky3 @ a0b92870] -- the same function ntt.hip computes (which stays the route of every shape this
//   file does not take: blow-ups other than 2, host inputs, kept coefficients, more than 2^20 or fewer than 2^5 rows).
//
// What changed against ntt.hip's per-matrix two-pass transforms:
//  * The matrices of one height (and coset shift) are ONE virtual row: a tile takes 32 consecutive virtual columns whatever
//    matrix they belong to, one column per lane (4-byte accesses), so odd widths and ragged chunks cost what they hold --
//    the 2^19-row chips of a fib shard (148 + 107 + 52 + 7 columns) are 10 tiles per row block instead of 15.
//  * The inverse transform's last pass, the coset scalings and both forward transforms' first pass are one kernel (k_mid):
//    a tile of the inverse's last pass holds the coefficients of one forward first-pass tile in bit-reversed order, and with
//    the tile in REGISTERS (below) the hand-over is a renaming of registers -- the coefficients are never written.  9 matrix
//    transfers per LDE instead of 12.
//  * Between the kernels the data lives in 32-column slabs ([slab][N][32 words]): every row segment there is one aligned
//    128-byte line, and the contiguous-row passes stream whole slabs.  Only the first read (the caller's row-major traces)
//    and the last write (the committed row-major LDE) touch unaligned w*4-byte rows.
//  * A thread is (row slot s, column c) and holds U = 32 rows of its column in registers: the five top stages of a tile run on
//    rows s + S j (j = register index) with no memory access at all, ONE exchange through LDS regroups the tile so that the
//    thread holds rows 32 s + j, and the remaining stages run in registers again -- two LDS transfers per tile instead of
//    ten; lanes run along a row everywhere, so there is no transposing load or store.  The twiddles of the second group are
//    the same for every thread and sit in scalar registers.
//  * k_mid's tiles are at most 64 KiB (16 columns for 2^10-row tiles), two workgroups per CU: one's exchange and load/store
//    phases run under the other's butterflies.
//
// Index arithmetic: tools/lde_model.py is this file's decomposition in numpy, checked there against the CPU restatement of the LDE.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <vector>

#include "babybear.h"
#include "commit.h"
#include "ctx.h"
#include "lde.h"

namespace lurkhip {

namespace {

constexpr int SLAB_LOG_W = 5;  // columns per slab of the intermediate layout: one 128-byte line per row
// 16-byte accesses through quad transposes (walk_load_x4 / walk_store_x4 below), measured and off: the tile kernels' memory time does
// not depend on the access width (2^20 x 64, butterflies skipped: k_out 218 us with dwords, 235 us with 16-byte pieces), only on the
// ALIGNMENT of the row segments (pitch 78 words: 421 us) -- tools/ubench_tilemem.hip's 5.2 against 3.5 TB/s is a copy loop's gain.
#ifndef LDE_SLAB_X4
#define LDE_SLAB_X4 0  // k_in's stores and k_out's loads
#endif
#ifndef LDE_MID_X4
#define LDE_MID_X4 0   // the fused pass's slab rows
#endif

struct LdeArgs {
    const uint32_t* src[LDE_MAX_MATS];  // N x width[m], row-major
    uint32_t* dst[LDE_MAX_MATS];        // 2N x width[m]: block q = coset q, rows in the DIF's (bit-reversed) order
    uint32_t width[LDE_MAX_MATS];
    uint32_t dpitch[LDE_MAX_MATS];      // row pitch of dst[m] in words: width[m], or the pitch of the padded group buffer dst[m] is a column range of
    uint32_t spitch[LDE_MAX_MATS];      // row pitch of src[m] in words (round 5: the prover's own traces are column ranges of aligned group buffers too)
    uint32_t start[LDE_MAX_MATS];       // virtual column of the matrix's first column (2^32 - 1 for unused slots)
    // Round 5: the LAST pass may walk another virtual row than the first two -- the output layout, in which entry m starts at
    // ostart[m] and the columns between the end of one entry and the start of the next are identically-zero columns of the
    // caller's matrices (commit.hip: live_runs; an entry of width 0 marks the start of a matrix whose first columns are dead):
    // their lanes load nothing, run the butterflies on zeros and store zeros, so that the pass still writes whole lines.
    // Without dead columns ostart == start and W_out == W.
    uint32_t ostart[LDE_MAX_MATS];
    uint32_t W_out;
    uint32_t cls[LDE_MAX_MATS];         // shift class of the matrix
    const uint32_t* scale[2][LDE_MAX_CLASSES];  // per coset and class: s_q^k / N, k < N
    const uint32_t *tw_inv, *tw_fwd;    // N/2 powers of the inverse / forward size-N root
    uint32_t* A;                        // inverse first pass -> fused pass: [slabs][N][32]
    uint32_t* B;                        // fused pass -> forward last pass: [2 cosets][slabs][N][32], then one spare slab (k_lde_out's padding lanes store there)
    uint32_t n_mats, W, n_cls, slabs;
    uint32_t col_base;                  // host side: first virtual column of the launches (a slab batch)
    int log_n, r1, r2;
    int col0, n_chunks;                 // this launch: chunks of (1 << LOG_C) virtual columns from col0 on
    uint32_t n_tiles, xcd_run;
    int in_canonical, out_canonical;    // convert the caller's words on the first load / the last store
    int stagger;                        // s_sleep(127) rounds the second half of the grid waits before its first tile (A/B hook)
    int x4;                             // 16-byte accesses on the matrices' side too (the slabs' side always has them)
    int dbg;                            // measurement hook (LURKHIP_LDE_DBG): bit 0 = skip the butterflies (memory traffic and exchanges only)
};

template <int LOG_R>
struct Geo {
    static constexpr int R = 1 << LOG_R;
    static constexpr int LOG_U = LOG_R < 5 ? LOG_R : 5;  // rows a thread holds
    static constexpr int U = 1 << LOG_U;
    static constexpr int LOG_S = LOG_R - LOG_U;          // row slots of a tile
    static constexpr int S = 1 << LOG_S;
};

// decimation-in-frequency butterflies on canonical Montgomery words: x <- x + y, y <- (x - y) tw
__device__ __forceinline__ void bfly(uint32_t& x, uint32_t& y, uint32_t tw) {
    const uint32_t sum = bb::add(x, y);
    const uint32_t r = (uint32_t)bb::smul((int32_t)(x - y), (int32_t)tw);
    y = bb::umin(r, r + bb::P);
    x = sum;
}
__device__ __forceinline__ void bfly_u(uint32_t& x, uint32_t& y, uint32_t tw_uniform) {  // twiddle in a scalar register
    const uint32_t sum = bb::add(x, y);
    const uint32_t r = (uint32_t)bb::sred(bb::mad_i64_u((int32_t)(x - y), (int32_t)tw_uniform, 0));
    y = bb::umin(r, r + bb::P);
    x = sum;
}
__device__ __forceinline__ uint32_t mulc(uint32_t a, uint32_t b) {  // canonical product through the signed chain
    const uint32_t r = (uint32_t)bb::smul((int32_t)a, (int32_t)b);
    return bb::umin(r, r + bb::P);
}
constexpr int brev_c(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
__device__ __forceinline__ void bfly_1(uint32_t& x, uint32_t& y) {  // twiddle one
    const uint32_t sum = bb::add(x, y);
    y = bb::sub(x, y);
    x = sum;
}

// K butterflies in lockstep: every step of the dependent chain (difference, product, m = lo * p^-1, reduction, the two range
// corrections) is issued for all K before the next step, so that a wave's consecutive instructions do not wait for one another
// (tools/ubench_bfly.hip: 43 -> 37 cycles per wave-butterfly at four waves per SIMD).  UNIFORM: the twiddles are scalar registers.
#ifndef LDE_BFLY_K
#define LDE_BFLY_K 4
#endif
template <int K, bool UNIFORM>
__device__ __forceinline__ void bfly_k(uint32_t* (&xs)[K], uint32_t* (&ys)[K], const uint32_t (&tw)[K]) {
    int32_t d[K], m[K];
    int64_t t[K];
    uint32_t sum[K];
#pragma unroll
    for (int i = 0; i < K; i++) d[i] = (int32_t)(*xs[i] - *ys[i]);
#pragma unroll
    for (int i = 0; i < K; i++) t[i] = UNIFORM ? bb::mad_i64_u(d[i], (int32_t)tw[i], 0) : bb::mad_i64(d[i], (int32_t)tw[i], 0);
#pragma unroll
    for (int i = 0; i < K; i++) sum[i] = *xs[i] + *ys[i];
#pragma unroll
    for (int i = 0; i < K; i++) m[i] = (int32_t)((uint32_t)t[i] * bb::MU);
#pragma unroll
    for (int i = 0; i < K; i++) t[i] = bb::mad_i64(m[i], -(int32_t)bb::P, t[i]);
#pragma unroll
    for (int i = 0; i < K; i++) *xs[i] = bb::umin(sum[i], sum[i] - bb::P);
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t r = (uint32_t)(t[i] >> 32);
        *ys[i] = bb::umin(r, r + bb::P);
    }
}

// Stage group 1: register j of slot s is tile row s + S j; in-register bit g is tile-row bit LOG_S + g.  The butterfly of rows
// (t, t + 2^b) takes tw[(1 << b) + (t mod 2^b)] = tw[s + S ((1 << g) + (j mod 2^g))]: thirty-one entries per thread at
// compile-time offsets from tw + s.
template <int LOG_R>
__device__ __forceinline__ void group1(uint32_t (&x)[Geo<LOG_R>::U], const uint32_t* __restrict__ tw_s) {
    using G = Geo<LOG_R>;
    constexpr int K = (G::U / 2) % LDE_BFLY_K == 0 && LDE_BFLY_K > 1 ? LDE_BFLY_K : 1;
#pragma unroll
    for (int g = G::LOG_U - 1; g >= 0; g--) {
        if constexpr (K > 1) {
#pragma unroll
            for (int p0 = 0; p0 < G::U / 2; p0 += K) {
                uint32_t *xs[K], *ys[K], tws[K];
#pragma unroll
                for (int i = 0; i < K; i++) {
                    const int p = p0 + i;
                    const int j = ((p >> g) << (g + 1)) | (p & ((1 << g) - 1));  // pair p of the stage: bit g clear
                    xs[i] = &x[j];
                    ys[i] = &x[j | (1 << g)];
                    tws[i] = tw_s[G::S * ((1 << g) + (j & ((1 << g) - 1)))];
                }
                bfly_k<K, false>(xs, ys, tws);
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::U; j++) {
                if (j & (1 << g)) continue;
                const int m = (1 << g) + (j & ((1 << g) - 1));
                bfly(x[j], x[j | (1 << g)], tw_s[G::S * m]);
            }
        }
    }
}
// Stage group 2: register j of slot s is tile row 32 s + j; stages LOG_S-1 .. 0 on in-register bits, the twiddle
// tw[(1 << g) + (j mod 2^g)] is the same for every thread: read once into scalar registers.  ONE0: the tile's lowest
// row bits are the transform's (a last pass) -- entry (1 << g) + 0 is one, no product.
// one stage (in-register bit GBIT) of group 2, K butterflies in lockstep: pairs in the order of their twiddle index jl, so that a last
// pass's twiddle-one butterflies (jl = 0) come in whole blocks
template <int LOG_R, bool ONE0, int GBIT, int K>
__device__ __forceinline__ void group2_stage(uint32_t (&x)[Geo<LOG_R>::U], const uint32_t (&tws)[Geo<LOG_R>::S]) {
    using G = Geo<LOG_R>;
    constexpr int PER_JL = (G::U / 2) >> GBIT;  // pairs sharing one jl
    constexpr int KK = PER_JL < K ? PER_JL : K;
#pragma unroll
    for (int jl = 0; jl < (1 << GBIT); jl++) {
#pragma unroll
        for (int h0 = 0; h0 < PER_JL; h0 += KK) {
            if (ONE0 && jl == 0) {
#pragma unroll
                for (int i = 0; i < KK; i++) {
                    const int j = ((h0 + i) << (GBIT + 1)) | jl;
                    bfly_1(x[j], x[j | (1 << GBIT)]);
                }
            } else {
                uint32_t *xs[KK], *ys[KK], tk[KK];
#pragma unroll
                for (int i = 0; i < KK; i++) {
                    const int j = ((h0 + i) << (GBIT + 1)) | jl;
                    xs[i] = &x[j];
                    ys[i] = &x[j | (1 << GBIT)];
                    tk[i] = tws[(1 << GBIT) + jl];
                }
                bfly_k<KK, true>(xs, ys, tk);
            }
        }
    }
    if constexpr (GBIT > 0) group2_stage<LOG_R, ONE0, GBIT - 1, K>(x, tws);
}
template <int LOG_R, bool ONE0>
__device__ __forceinline__ void group2(uint32_t (&x)[Geo<LOG_R>::U], const uint32_t* __restrict__ tw) {
    using G = Geo<LOG_R>;
    constexpr int K = (G::U / 2) % LDE_BFLY_K == 0 && LDE_BFLY_K > 1 ? LDE_BFLY_K : 1;
    if constexpr (G::LOG_S > 0) {
        uint32_t tws[G::S];
        tws[0] = 0;
#pragma unroll
        for (int i = 1; i < G::S; i++) tws[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)tw[i]);
        if constexpr (K > 1) {
            group2_stage<LOG_R, ONE0, G::LOG_S - 1, K>(x, tws);
        } else {
#pragma unroll
            for (int g = G::LOG_S - 1; g >= 0; g--) {
#pragma unroll
                for (int j = 0; j < G::U; j++) {
                    if (j & (1 << g)) continue;
                    const int jl = j & ((1 << g) - 1);
                    if (ONE0 && jl == 0) bfly_1(x[j], x[j | (1 << g)]);
                    else bfly_u(x[j], x[j | (1 << g)], tws[(1 << g) + jl]);
                }
            }
        }
    }
}

// LDS tile: row t, column c at word ((t + (t >> 5)) << LOG_C) + c for tiles narrower than 32 columns (one spare row per 32:
// in the read pattern a 32-lane group covers rows 32 apart, which the padding moves to different banks), (t << 5) + c for
// 32-column tiles (a group is one row).  Write pattern rows s + S j, read pattern rows 32 s + j: both are compile-time offsets
// from a per-thread base.
template <int LOG_R, int LOG_C>
constexpr int tile_words() {
    return LOG_C >= 5 ? (1 << (LOG_R + LOG_C)) : (((1 << LOG_R) + ((1 << LOG_R) >> 5) + 1) << LOG_C);
}
template <int LOG_R, int LOG_C>
__device__ __forceinline__ void tile_write(uint32_t* __restrict__ tile, int s, int c, const uint32_t (&x)[Geo<LOG_R>::U]) {
    using G = Geo<LOG_R>;
    uint32_t* __restrict__ p = tile + (s << LOG_C) + c;
#pragma unroll
    for (int j = 0; j < G::U; j++) {
        const int t = G::S * j;  // + s, which is below S: no carry into the padding term
        p[LOG_C >= 5 ? (t << LOG_C) : ((t + (t >> 5)) << LOG_C)] = x[j];
    }
}
template <int LOG_R, int LOG_C>
__device__ __forceinline__ void tile_read(const uint32_t* __restrict__ tile, int s, int c, uint32_t (&x)[Geo<LOG_R>::U]) {
    using G = Geo<LOG_R>;
    const uint32_t* __restrict__ p = tile + (LOG_C >= 5 ? ((G::U * s) << LOG_C) : ((G::U * s + s) << LOG_C)) + c;  // (32 s) >> 5 = s
#pragma unroll
    for (int j = 0; j < G::U; j++) x[j] = p[j << LOG_C];
}

// Global accesses of a thread's U rows walk ONE pointer by a stride.  (Spelled as base + f(j) the compiler hoists the thirty-two
// tile-invariant row offsets of a thread out of the persistent loop -- as 64-bit pairs -- and spills them.)
// (The pointers are cast to the global address space: read out of the LDS descriptor table their address space is unknown to the
// compiler, and a FLAT load also counts on the LDS counter -- every wait for an LDS read would wait for the prefetched rows.)
typedef __attribute__((address_space(1))) uint32_t global_u32;
template <int U>
__device__ __forceinline__ void walk_load(uint32_t (&x)[U], const uint32_t* __restrict__ p0, size_t stride) {
    const global_u32* p = (const global_u32*)p0;
#pragma unroll
    for (int j = 0; j < U; j++) {
        x[j] = *p;
        p += stride;
    }
}
template <int U>
__device__ __forceinline__ void walk_store(const uint32_t (&x)[U], uint32_t* __restrict__ p0, size_t stride) {
    global_u32* p = (global_u32*)p0;
#pragma unroll
    for (int j = 0; j < U; j++) {
        *p = x[j];
        p += stride;
    }
}
// 16-byte accesses.  One dword per lane sustains about 3.5 TB/s on this chip however well the lanes coalesce, four dwords per lane
// 5.2 TB/s (tools/ubench_tilemem.hip) -- and with the butterflies removed these kernels run at their memory time.
// A thread owns ONE column, so the four lanes of a quad (four adjacent columns of one row slot) each move a 16-byte piece of a
// different row -- lane q rows 4 m + q of the thread's rows -- and a 4 x 4 transpose inside the quad (two conditional register
// rotations around three DPP quad_perm moves, 19 full-rate instructions) turns row pieces into the per-column registers and back.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 global_u32x4;
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
// lane q of a quad holds a[0..3]: on return it holds what lanes 0..3 held in their a[q]
__device__ __forceinline__ void quad_transpose(uint32_t (&a)[4], int q) {
    const bool b0 = (q & 1) != 0, b1 = (q & 2) != 0;
    uint32_t t[4], u[4], w[4];
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = b0 ? a[(r + 1) & 3] : a[r];
#pragma unroll
    for (int r = 0; r < 4; r++) u[r] = b1 ? t[(r + 2) & 3] : t[r];  // u[r] = a[(r + q) mod 4]
    w[0] = u[0];
    w[1] = quad_perm<147>(u[1]);  // quad_perm:[3,0,1,2]: lane q takes register r from lane (q - r) mod 4
    w[2] = quad_perm<78>(u[2]);   // [2,3,0,1]
    w[3] = quad_perm<57>(u[3]);   // [1,2,3,0]
    const uint32_t z[4] = {w[0], w[3], w[2], w[1]};  // w[r] came from lane (q - r): z[i] from lane (q + i)
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = b0 ? z[(j + 3) & 3] : z[j];
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = b1 ? t[(j + 2) & 3] : t[j];  // a[j] = z[(j - q) mod 4]: from lane j
}
// p0: element (first row of the thread, first column of its quad); register j is row j * row_step words further on.
// The loads leave the row pieces as they arrive (raw[4 m + i] = word i of the m-th piece); quad_unpack turns them into the thread's
// column -- separately, so that a prefetch can stay in flight as plain registers and be transposed when it is consumed.
template <int U>
__device__ __forceinline__ void walk_load_x4(uint32_t (&raw)[U], const uint32_t* __restrict__ p0, size_t row_step, int q) {
    static_assert(U % 4 == 0, "quads of rows");
    const global_u32* p = (const global_u32*)p0 + (size_t)q * row_step;
#pragma unroll
    for (int m = 0; m < U / 4; m++) {
        const u32x4 v = *(const global_u32x4*)p;
        p += 4 * row_step;
        raw[4 * m] = v.x, raw[4 * m + 1] = v.y, raw[4 * m + 2] = v.z, raw[4 * m + 3] = v.w;
    }
}
template <int U>
__device__ __forceinline__ void quad_unpack(uint32_t (&x)[U], const uint32_t (&raw)[U], int q) {
#pragma unroll
    for (int m = 0; m < U / 4; m++) {
        uint32_t a[4] = {raw[4 * m], raw[4 * m + 1], raw[4 * m + 2], raw[4 * m + 3]};
        quad_transpose(a, q);
#pragma unroll
        for (int i = 0; i < 4; i++) x[4 * m + i] = a[i];
    }
}
template <int U>
__device__ __forceinline__ void walk_store_x4(const uint32_t (&x)[U], uint32_t* __restrict__ p0, size_t row_step, int q) {
    static_assert(U % 4 == 0, "quads of rows");
    global_u32* p = (global_u32*)p0 + (size_t)q * row_step;
#pragma unroll
    for (int m = 0; m < U / 4; m++) {
        uint32_t a[4] = {x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3]};
        quad_transpose(a, q);
        u32x4 v;
        v.x = a[0], v.y = a[1], v.z = a[2], v.w = a[3];
        *(global_u32x4*)p = v;
        p += 4 * row_step;
    }
}
template <int U>
__device__ __forceinline__ void zero_rows(uint32_t (&x)[U]) {
#pragma unroll
    for (int j = 0; j < U; j++) x[j] = 0u;
}

struct ColRef {
    const uint32_t* src;  // the column's first element (row 0)
    uint32_t* dst;
    uint32_t w;           // row pitch of its matrix in words
    uint32_t dw;          // row pitch of its LDE in words
    uint32_t cls;
    bool valid;
};
// The matrices' descriptors live in LDS: a lane finds its matrix by comparing its virtual column with the start columns (scalar
// operands) and reads the entry with LDS loads.  (Read from the kernel-argument segment with a per-lane index they were vector
// memory loads -- and a dependent vector load must wait for everything issued before it on the same in-order counter, i.e. for
// the previous tile's stores to drain.)
struct MatDesc {
    const uint32_t* src;
    uint32_t* dst;
    uint32_t w, cls, start, dw, cols, ostart;
};
constexpr int DESC_WORDS = LDE_MAX_MATS * (int)(sizeof(MatDesc) / 4);
__device__ __forceinline__ void stage_descs(const LdeArgs& a, MatDesc* __restrict__ descs) {
    if (threadIdx.x < LDE_MAX_MATS) {
        const int m = (int)threadIdx.x;
        MatDesc d;
        d.src = a.src[m];
        d.dst = a.dst[m];
        d.w = a.spitch[m];
        d.cls = a.cls[m];
        d.start = a.start[m];
        d.dw = a.dpitch[m];
        d.cols = a.width[m];
        d.ostart = a.ostart[m];
        descs[m] = d;
    }
}
__device__ __forceinline__ ColRef locate_col(const LdeArgs& a, const MatDesc* __restrict__ descs, uint32_t vc) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 1; i < LDE_MAX_MATS; i++)
        if (vc >= a.start[i]) m = (uint32_t)i;  // start[i] = 2^32 - 1 past the last matrix
    const MatDesc d = descs[m];
    ColRef r;
    r.valid = vc < a.W;
    const uint32_t col = r.valid ? vc - d.start : 0u;
    r.src = d.src + col;
    r.dst = d.dst + col;
    r.w = d.w;
    r.dw = d.dw;
    r.cls = d.cls;
    return r;
}

// the last pass's view of a column of the OUTPUT layout (LdeArgs::ostart)
struct OutRef {
    uint32_t* dst;  // the column's first element in the LDE matrix it belongs to
    uint32_t dw;
    uint32_t svc;   // its virtual column in the slabs (live columns only)
    bool live, valid;
};
__device__ __forceinline__ OutRef locate_out(const LdeArgs& a, const MatDesc* __restrict__ descs, uint32_t vc) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 1; i < LDE_MAX_MATS; i++)
        if (vc >= a.ostart[i]) m = (uint32_t)i;
    const MatDesc d = descs[m];
    OutRef r;
    r.valid = vc < a.W_out;
    const uint32_t off = r.valid ? vc - d.ostart : 0u;
    r.live = r.valid && off < d.cols;
    r.dst = d.dst + off;
    r.dw = d.dw;
    r.svc = d.start + off;
    return r;
}

// persistent, XCD-contiguous tile order (as ntt.hip): consecutive tile ids share rows (adjacent column chunks) or hold adjacent
// rows; workgroups are dealt round-robin to the 8 XCDs, so every XCD walks a contiguous run of ids with its workgroups abreast
struct TileWalk {
    uint32_t first, step, count;
};
// Workgroups that share a CU start in lockstep and stay there (same tile, same phases): the second half of the grid -- the
// second workgroup of every CU under round-robin dispatch -- may start a fraction of a tile late, so that one's memory
// phases fall under the other's butterflies.
__device__ __forceinline__ void stagger_start(const LdeArgs& a) {
    if (a.stagger > 0 && blockIdx.x >= (gridDim.x >> 1))
        for (int i = 0; i < a.stagger; i++) __builtin_amdgcn_s_sleep(127);
}
__device__ __forceinline__ TileWalk tile_walk(const LdeArgs& a) {
    const uint32_t wg = a.xcd_run ? (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t wgs = a.xcd_run ? (gridDim.x >> 3) : gridDim.x;
    const uint32_t run = a.xcd_run ? a.xcd_run : a.n_tiles;
    const uint32_t run0 = a.xcd_run ? (blockIdx.x & 7u) * a.xcd_run : 0u;
    TileWalk w;
    w.first = run0 + wg;
    w.step = wgs;
    w.count = wg >= run ? 0u : (run - wg + wgs - 1) / wgs;
    return w;
}

__device__ __forceinline__ size_t slab_off(const LdeArgs& a, uint32_t vc) {  // word offset of virtual column vc in row 0 of its slab
    return ((size_t)(vc >> SLAB_LOG_W) << (a.log_n + SLAB_LOG_W)) + (vc & ((1u << SLAB_LOG_W) - 1u));
}

// entry idx = (1 << b) + tl of a tile's twiddle table: w^(((tl << bit_lo) | lo) << (log_n - bit_lo - b - 1)), w the size-N root
__device__ __forceinline__ uint32_t tile_twiddle(const uint32_t* __restrict__ tw, int idx, int log_n, int bit_lo, uint32_t lo) {
    const int b = 31 - __clz(idx | 1);
    const uint32_t tl = idx ? (uint32_t)idx - (1u << b) : 0u;  // entry 0 is never read
    return tw[(size_t)((tl << bit_lo) | lo) << (log_n - bit_lo - b - 1)];
}

template <int LOG_R, int LOG_C>
constexpr int lde_threads() {
    return (Geo<LOG_R>::S << LOG_C) < 64 ? 64 : (Geo<LOG_R>::S << LOG_C);
}

// ------------------------------------------------------------------------------------------------ k_in
// Inverse transform, top r1 = LOG_R stages: tile `lo` = rows (t << r2) | lo of the caller's matrices (strided), written to the
// same rows of the slabs.  The next tile's rows are requested as soon as this one's are in LDS.
template <int LOG_R, int LOG_C>
__global__ __launch_bounds__((lde_threads<LOG_R, LOG_C>()), 4) void k_lde_in(LdeArgs a) {
    using G = Geo<LOG_R>;
    constexpr int C = 1 << LOG_C, THREADS = G::S * C, NT = lde_threads<LOG_R, LOG_C>();
    constexpr int TWN = (G::R + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* tile = smem;
    uint32_t* twb = smem + tile_words<LOG_R, LOG_C>();  // two tables of R entries, used in turn
    MatDesc* descs = reinterpret_cast<MatDesc*>(twb + 2 * G::R);
    const int tid = NT > THREADS ? (int)threadIdx.x % THREADS : (int)threadIdx.x;  // surplus threads of tiny tiles shadow real ones
    const int s = tid >> LOG_C, c = tid & (C - 1);
    const int r2 = a.r2;
    const TileWalk walk = tile_walk(a);
    if (walk.count == 0) return;
    stagger_start(a);
    stage_descs(a, descs);
    __syncthreads();

    // (memory schedule: see k_lde_out -- next tile's rows first, this tile's stores after they have landed)
    uint32_t x[G::U], nx[G::U], y[G::U], tv[TWN], ntv[TWN];
    struct At {
        uint32_t lo, vc;
    };
    auto fetch = [&](uint32_t it, uint32_t (&dst)[G::U], uint32_t (&dtv)[TWN]) {
        const uint32_t id = walk.first + it * walk.step;
        At at;
        at.lo = id / (uint32_t)a.n_chunks;
        at.vc = (uint32_t)a.col0 + (id - at.lo * (uint32_t)a.n_chunks) * C + (uint32_t)c;
        if (G::U >= 4 && a.x4) {  // every matrix of the group starts and ends on a 16-byte boundary: a quad shares one matrix
            const ColRef ref = locate_col(a, descs, at.vc & ~3u);
            walk_load_x4<G::U>(dst, ref.src + (((size_t)s << r2) | at.lo) * ref.w, ((size_t)G::S << r2) * ref.w, c & 3);
        } else {
            const ColRef ref = locate_col(a, descs, at.vc);
            // a padding column reads column 0 of the last matrix (its values go nowhere): no branch around memory operations
            walk_load<G::U>(dst, ref.src + (((size_t)s << r2) | at.lo) * ref.w, ((size_t)G::S << r2) * ref.w);
        }
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            const int idx = (int)threadIdx.x + i * NT;
            dtv[i] = tile_twiddle(a.tw_inv, idx < G::R ? idx : 0, a.log_n, r2, at.lo);
        }
        return at;
    };
    At nxt = fetch(0, nx, ntv);
#pragma unroll
    for (int j = 0; j < G::U; j++) asm volatile("" : "+v"(nx[j]));  // landed before the loop: no load is pending on any path into its head
#pragma unroll
    for (int i = 0; i < TWN; i++) asm volatile("" : "+v"(ntv[i]));
    for (uint32_t it = 0; it < walk.count; it++) {
        const At cur = nxt;
        if (G::U >= 4 && a.x4) quad_unpack<G::U>(x, nx, c & 3);
        else {
#pragma unroll
            for (int j = 0; j < G::U; j++) x[j] = nx[j];
        }
#pragma unroll
        for (int i = 0; i < TWN; i++) tv[i] = ntv[i];
        if (a.dbg & 2) nxt = fetch(it + 1 < walk.count ? it + 1 : it, nx, ntv);
        uint32_t* __restrict__ tw = twb + (it & 1u) * G::R;
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            const int idx = (int)threadIdx.x + i * NT;
            if (idx < G::R) tw[idx] = tv[i];
        }
        __syncthreads();  // the table is complete; every thread has left the previous tile
        if (a.in_canonical) {
#pragma unroll
            for (int j = 0; j < G::U; j++) x[j] = bb::to_monty(x[j]);
        }
        if (!(a.dbg & 1)) group1<LOG_R>(x, tw + s);
        if constexpr (G::LOG_S > 0) {
            tile_write<LOG_R, LOG_C>(tile, s, c, x);
            if (!(a.dbg & 2)) nxt = fetch(it + 1 < walk.count ? it + 1 : it, nx, ntv);  // (see k_lde_out: half a tile after the stores)
            __syncthreads();
            tile_read<LOG_R, LOG_C>(tile, s, c, y);
            if (!(a.dbg & 1)) group2<LOG_R, false>(y, tw);
        } else {
#pragma unroll
            for (int j = 0; j < G::U; j++) y[j] = x[j];
            if (!(a.dbg & 2)) nxt = fetch(it + 1 < walk.count ? it + 1 : it, nx, ntv);
        }
#pragma unroll
        for (int j = 0; j < G::U; j++) asm volatile("" : "+v"(nx[j]));  // the next tile's rows have landed BEFORE this tile's stores are issued
#pragma unroll
        for (int i = 0; i < TWN; i++) asm volatile("" : "+v"(ntv[i]));
        if constexpr (G::U >= 4 && LDE_SLAB_X4)
            walk_store_x4<G::U>(y, a.A + slab_off(a, cur.vc & ~3u) + (((((size_t)(G::U * s)) << r2) | cur.lo) << SLAB_LOG_W), (size_t)1 << (r2 + SLAB_LOG_W), c & 3);
        else
            walk_store<G::U>(y, a.A + slab_off(a, cur.vc) + (((((size_t)(G::U * s)) << r2) | cur.lo) << SLAB_LOG_W), (size_t)1 << (r2 + SLAB_LOG_W));
    }
}

// ------------------------------------------------------------------------------------------------ k_out
// Forward transform, low r1 = LOG_R stages on contiguous rows of a coset's slabs, stored into the coset's block of the LDE
// matrices.  Tile id = (coset, row block hi, column chunk), chunk fastest.
template <int LOG_R, int LOG_C>
__global__ __launch_bounds__((lde_threads<LOG_R, LOG_C>()), 4) void k_lde_out(LdeArgs a) {
    using G = Geo<LOG_R>;
    constexpr int C = 1 << LOG_C, THREADS = G::S * C, NT = lde_threads<LOG_R, LOG_C>();
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* tile = smem;
    uint32_t* tw = smem + tile_words<LOG_R, LOG_C>();
    MatDesc* descs = reinterpret_cast<MatDesc*>(tw + G::R);
    const int tid = NT > THREADS ? (int)threadIdx.x % THREADS : (int)threadIdx.x;
    const int s = tid >> LOG_C, c = tid & (C - 1);
    const int r2 = a.r2;
    const TileWalk walk = tile_walk(a);
    if (walk.count == 0) return;
    stagger_start(a);
    stage_descs(a, descs);
    for (int idx = (int)threadIdx.x; idx < G::R; idx += NT) tw[idx] = tile_twiddle(a.tw_fwd, idx, a.log_n, 0, 0u);
    __syncthreads();

    // Memory schedule of a tile: the NEXT tile's rows are requested first (into nx), then this tile runs, and its stores are
    // issued only after nx has landed and moved to x.  A wave's loads and stores share one in-order counter and the compiler waits
    // for "everything" whenever both kinds are pending: with the loads requested after the stores (or the stores pending when the
    // loads are needed) every tile waited for the previous tile's stores to drain.  In this order the loads have a whole tile to
    // arrive and nothing ever waits for a store.
    uint32_t x[G::U], nx[G::U], y[G::U];
    struct At {
        uint32_t q, hi, vc;
    };
    auto fetch = [&](uint32_t it, uint32_t (&dst)[G::U]) {
        const uint32_t id = walk.first + it * walk.step;
        const uint32_t blk = id / (uint32_t)a.n_chunks;
        At at;
        at.vc = (uint32_t)a.col0 + (id - blk * (uint32_t)a.n_chunks) * C + (uint32_t)c;  // a column of the OUTPUT layout
        at.q = blk >> r2;
        at.hi = blk & ((1u << r2) - 1u);
        const uint32_t* __restrict__ in = a.B + ((size_t)at.q * a.slabs << (a.log_n + SLAB_LOG_W)) + ((((size_t)at.hi << LOG_R) | (size_t)s) << SLAB_LOG_W);
        const OutRef o = locate_out(a, descs, at.vc);
        if constexpr (G::U >= 4 && LDE_SLAB_X4) {
            // (the quad loads know nothing of dead columns: a dead column's slab offset is some live column's.  The variant is an A/B
            // switch for dense groups only; launch_kind refuses it together with dead columns -- ADVICE round 5)
            walk_load_x4<G::U>(dst, in + slab_off(a, o.svc & ~3u), (size_t)G::S << SLAB_LOG_W, c & 3);
        } else {
            // A dead or padding column is the zero polynomial, and so is every butterfly of it: it loads nothing.  (The zeros are
            // written BEFORE the live lanes' loads are issued: moves into registers that loads of other lanes are in flight for make
            // the compiler wait for those loads.  Measured alternatives, all slower: every row of a dead column loaded from one
            // line of zeros at stride 0 -- a per-lane stride costs the address arithmetic of 32 loads --; the slabs of the last
            // hand-over laid out by output columns so that this pass loads and stores 1:1 -- the fused pass then scatters.)
            zero_rows<G::U>(dst);
            if (o.live) walk_load<G::U>(dst, in + slab_off(a, o.svc), (size_t)G::S << SLAB_LOG_W);
        }
        return at;
    };
    At nxt = fetch(0, nx);
#pragma unroll
    for (int j = 0; j < G::U; j++) asm volatile("" : "+v"(nx[j]));  // landed before the loop: no load is pending on any path into its head
    for (uint32_t it = 0; it < walk.count; it++) {
        const At cur = nxt;
        if constexpr (G::U >= 4 && LDE_SLAB_X4) quad_unpack<G::U>(x, nx, c & 3);
        else {
#pragma unroll
            for (int j = 0; j < G::U; j++) x[j] = nx[j];
        }
        if (a.dbg & 2) nxt = fetch(it + 1 < walk.count ? it + 1 : it, nx);  // (A/B: the request at the head of the tile)
        const OutRef ref = locate_out(a, descs, cur.vc);
        const size_t row0 = ((size_t)cur.q << a.log_n) | ((size_t)cur.hi << LOG_R);
        __syncthreads();  // (first tile: the table is complete) every thread has left the previous tile
        if (!(a.dbg & 1)) group1<LOG_R>(x, tw + s);
        if constexpr (G::LOG_S > 0) {
            tile_write<LOG_R, LOG_C>(tile, s, c, x);
            // The next tile's rows are requested HERE: a wave has 64 memory operations in flight at most, so at the head of the
            // tile the thirty-two requests queue behind the previous tile's thirty-two stores (the wave stands at the issue until
            // they have drained); half a tile later those are gone and the rows still have the second stage group to arrive.
            if (!(a.dbg & 2)) nxt = fetch(it + 1 < walk.count ? it + 1 : it, nx);  // (the last tile re-reads itself: no branch around memory operations)
            __syncthreads();
            tile_read<LOG_R, LOG_C>(tile, s, c, y);
            if (!(a.dbg & 1)) group2<LOG_R, true>(y, tw);
        } else {
#pragma unroll
            for (int j = 0; j < G::U; j++) y[j] = x[j];
            if (!(a.dbg & 2)) nxt = fetch(it + 1 < walk.count ? it + 1 : it, nx);
        }
        if (a.out_canonical) {
#pragma unroll
            for (int j = 0; j < G::U; j++) y[j] = bb::from_monty(y[j]);
        }
#pragma unroll
        for (int j = 0; j < G::U; j++) asm volatile("" : "+v"(nx[j]));  // the next tile's rows have landed BEFORE this tile's stores are issued
        // a padding column (past the last matrix: the ragged last chunk) stores into the spare slab behind the two cosets': no branch
        uint32_t* const spare = a.B + ((size_t)2 * a.slabs << (a.log_n + SLAB_LOG_W));
        if (G::U >= 4 && a.x4) {
            const OutRef rq = locate_out(a, descs, cur.vc & ~3u);
            uint32_t* __restrict__ out = rq.valid ? rq.dst + (row0 + (size_t)(G::U * s)) * rq.dw : spare + (cur.vc & 28u);
            walk_store_x4<G::U>(y, out, rq.valid ? (size_t)rq.dw : (size_t)0, c & 3);
        } else {
            uint32_t* __restrict__ out = ref.valid ? ref.dst + (row0 + (size_t)(G::U * s)) * ref.dw : spare + (cur.vc & 31u);
            walk_store<G::U>(y, out, ref.valid ? (size_t)ref.dw : (size_t)0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ k_mid / k_small
// The fused pass.  Tile i: contiguous rows (hi << r2) | t of the slabs, hi = bitrev(i): after the inverse's low r2 = LOG_R stages
// register j of slot s holds coefficient k = bitrev_n(hi << r2 | 32 s + j), i.e. row t' = bitrev(s) + S bitrev5(j) of the forward
// first-pass tile lo' = i -- slot bitrev(s), register bitrev5(j) of stage group 1's layout.  Per coset: times s_q^k / N,
// forward top r2 stages, rows (t' << r1) | lo' of the coset's slabs.  (Tiles are walked by lo', so that the workgroups
// running at one time write adjacent rows.)
// DIRECT (k_small, N = 2^LOG_R): the tile is the whole column -- rows come from the caller's matrices and go to the LDE.
template <int LOG_R, int LOG_C, bool DIRECT>
__global__ __launch_bounds__((lde_threads<LOG_R, LOG_C>()), 4) void k_lde_mid(LdeArgs a) {
    using G = Geo<LOG_R>;
    constexpr int C = 1 << LOG_C, THREADS = G::S * C, NT = lde_threads<LOG_R, LOG_C>();
    constexpr int TWN = (G::R + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* tile = smem;
    MatDesc* descs = reinterpret_cast<MatDesc*>(smem + tile_words<LOG_R, LOG_C>());
    uint32_t* twi = smem + tile_words<LOG_R, LOG_C>() + DESC_WORDS;  // inverse twiddles of the pass (every tile)
    uint32_t* twf = twi + G::R;                         // forward twiddles of the tile
    uint32_t* scl = twf + G::R;                         // [class][R]: the current coset's scales of the tile's rows
    const int tid = NT > THREADS ? (int)threadIdx.x % THREADS : (int)threadIdx.x;
    const int s = tid >> LOG_C, c = tid & (C - 1);
    const int s2 = G::LOG_S > 0 ? (int)(__brev((uint32_t)s) >> (32 - (G::LOG_S > 0 ? G::LOG_S : 1))) : 0;
    const int r1 = DIRECT ? 0 : a.r1;
    const TileWalk walk = tile_walk(a);
    if (walk.count == 0) return;
    stagger_start(a);
    stage_descs(a, descs);
    for (int idx = (int)threadIdx.x; idx < G::R; idx += NT) twi[idx] = tile_twiddle(a.tw_inv, idx, a.log_n, 0, 0u);
    __syncthreads();

    uint32_t x[G::U], coef[G::U], tfv[TWN], scv[2][LDE_MAX_CLASSES][TWN];
    ColRef ref;
    uint32_t lo2 = 0, vc = 0;
    auto fetch = [&](uint32_t it) {
        const uint32_t id = walk.first + it * walk.step;
        lo2 = id / (uint32_t)a.n_chunks;
        vc = (uint32_t)a.col0 + (id - lo2 * (uint32_t)a.n_chunks) * C + (uint32_t)c;
        ref = locate_col(a, descs, vc);
        if constexpr (DIRECT) {
            if (ref.valid) walk_load<G::U>(x, ref.src + (size_t)s * ref.w, (size_t)G::S * ref.w);
            else zero_rows<G::U>(x);
        } else {
            const uint32_t hi = r1 ? (__brev(lo2) >> (32 - r1)) : 0u;
            if constexpr (G::U >= 4 && LDE_MID_X4)  // raw row pieces: transposed into the thread's column where the tile starts (quad_unpack)
                walk_load_x4<G::U>(x, a.A + slab_off(a, vc & ~3u) + ((((size_t)hi << LOG_R) | (size_t)s) << SLAB_LOG_W), (size_t)G::S << SLAB_LOG_W, c & 3);
            else
                walk_load<G::U>(x, a.A + slab_off(a, vc) + ((((size_t)hi << LOG_R) | (size_t)s) << SLAB_LOG_W), (size_t)G::S << SLAB_LOG_W);
        }
    };
    fetch(0);
    for (uint32_t it = 0; it < walk.count; it++) {
        // the tile's forward twiddles and both cosets' scales: requested now, written to LDS where the tables are free
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            const int idx = (int)threadIdx.x + i * NT;
            const bool ok = idx < G::R;
            tfv[i] = ok ? tile_twiddle(a.tw_fwd, idx, a.log_n, r1, lo2) : 0u;
#pragma unroll
            for (int cl = 0; cl < LDE_MAX_CLASSES; cl++) {
                if ((uint32_t)cl < a.n_cls) {
                    const size_t k = ((size_t)(ok ? idx : 0) << r1) | lo2;
                    scv[0][cl][i] = a.scale[0][cl][k];
                    scv[1][cl][i] = a.scale[1][cl][k];
                }
            }
        }
        const uint32_t cur_lo2 = lo2, cur_vc = vc;
        const ColRef cur = ref;
        __syncthreads();  // B0: (first tile: twi is complete) every thread has left the previous tile: tile, twf and scl are free
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            const int idx = (int)threadIdx.x + i * NT;
            if (idx < G::R) {
                twf[idx] = tfv[i];
#pragma unroll
                for (int cl = 0; cl < LDE_MAX_CLASSES; cl++)
                    if ((uint32_t)cl < a.n_cls) scl[cl * G::R + idx] = scv[0][cl][i];
            }
        }
        if constexpr (!DIRECT && G::U >= 4 && LDE_MID_X4) {
            uint32_t raw[G::U];
#pragma unroll
            for (int j = 0; j < G::U; j++) raw[j] = x[j];
            quad_unpack<G::U>(x, raw, c & 3);
        }
        if (DIRECT && a.in_canonical) {
#pragma unroll
            for (int j = 0; j < G::U; j++) x[j] = bb::to_monty(x[j]);
        }
        if (!(a.dbg & 1)) group1<LOG_R>(x, twi + s);
        if constexpr (G::LOG_S > 0) {
            tile_write<LOG_R, LOG_C>(tile, s, c, x);
            __syncthreads();  // B1
            tile_read<LOG_R, LOG_C>(tile, s, c, coef);
            if (!(a.dbg & 1)) group2<LOG_R, true>(coef, twi);
        } else {
#pragma unroll
            for (int j = 0; j < G::U; j++) coef[j] = x[j];
            __syncthreads();  // B1: twf and scl are complete
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t* __restrict__ sc = scl + cur.cls * G::R + s2;
#pragma unroll
            for (int j = 0; j < G::U; j++) {
                const int j2 = brev_c(j, G::LOG_U);
                x[j2] = mulc(coef[j], sc[G::S * j2]);
            }
            if (!(a.dbg & 1)) group1<LOG_R>(x, twf + s2);
            uint32_t z[G::U];
            if constexpr (G::LOG_S == 0) {  // no second group: the results leave x before the next tile's rows are requested into it
#pragma unroll
                for (int j = 0; j < G::U; j++) z[j] = x[j];
            }
            if constexpr (G::LOG_S > 0) {
                __syncthreads();  // B2 / B4: every thread has read the tile (and this coset's scales)
                tile_write<LOG_R, LOG_C>(tile, s2, c, x);
            }
            if (q == 0) {
                if constexpr (G::LOG_S == 0) __syncthreads();  // every thread has read coset 0's scales
#pragma unroll
                for (int i = 0; i < TWN; i++) {
                    const int idx = (int)threadIdx.x + i * NT;
                    if (idx < G::R) {
#pragma unroll
                        for (int cl = 0; cl < LDE_MAX_CLASSES; cl++)
                            if ((uint32_t)cl < a.n_cls) scl[cl * G::R + idx] = scv[1][cl][i];
                    }
                }
            } else {
                fetch(it + 1 < walk.count ? it + 1 : it);  // x is free: the next tile's rows fly under this coset's second stage group and its stores
            }
            if constexpr (G::LOG_S > 0) {
                __syncthreads();  // B3 / B5
                tile_read<LOG_R, LOG_C>(tile, s, c, z);
                if (!(a.dbg & 1)) group2<LOG_R, false>(z, twf);
            } else {
                if (q == 0) __syncthreads();  // coset 1's scales are complete
            }
            if constexpr (DIRECT) {
                if (a.out_canonical) {
#pragma unroll
                    for (int j = 0; j < G::U; j++) z[j] = bb::from_monty(z[j]);
                }
                if (cur.valid) walk_store<G::U>(z, cur.dst + (((size_t)q << a.log_n) + (size_t)(G::U * s)) * cur.dw, (size_t)cur.dw);
            } else {
                uint32_t* __restrict__ out = a.B + ((size_t)q * a.slabs << (a.log_n + SLAB_LOG_W)) + slab_off(a, cur_vc);
                if constexpr (G::U >= 4 && LDE_MID_X4) {
                    uint32_t* __restrict__ outq = a.B + ((size_t)q * a.slabs << (a.log_n + SLAB_LOG_W)) + slab_off(a, cur_vc & ~3u);
                    walk_store_x4<G::U>(z, outq + (((((size_t)(G::U * s)) << r1) | cur_lo2) << SLAB_LOG_W), (size_t)1 << (r1 + SLAB_LOG_W), c & 3);
                } else {
                    walk_store<G::U>(z, out + (((((size_t)(G::U * s)) << r1) | cur_lo2) << SLAB_LOG_W), (size_t)1 << (r1 + SLAB_LOG_W));
                }
            }
        }
    }
}

template <class K>
void opt_in_lds(K kern, size_t lds) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

constexpr size_t LDS_PER_CU = 160 * 1024;

template <int LOG_R, int LOG_C>
size_t lds_in() { return ((size_t)tile_words<LOG_R, LOG_C>() + 2u * Geo<LOG_R>::R + DESC_WORDS) * 4; }
template <int LOG_R, int LOG_C>
size_t lds_out() { return ((size_t)tile_words<LOG_R, LOG_C>() + Geo<LOG_R>::R + DESC_WORDS) * 4; }
template <int LOG_R, int LOG_C>
size_t lds_mid(uint32_t n_cls) { return ((size_t)tile_words<LOG_R, LOG_C>() + (2u + n_cls) * Geo<LOG_R>::R + DESC_WORDS) * 4; }

// grid of a persistent launch: as many workgroups as stay resident (LDS, 16 waves per CU at 128 VGPRs), a multiple of 8 when
// the tile order is XCD-contiguous
void size_grid(lurkhip_ctx* ctx, LdeArgs& a, size_t tiles, int threads, size_t lds, unsigned* blocks) {
    a.n_tiles = (uint32_t)tiles;
    a.xcd_run = (tiles % 8 == 0 && tiles >= 64) ? (uint32_t)(tiles / 8) : 0u;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(1024 / (size_t)threads, LDS_PER_CU / (lds + 128)));
    size_t b = std::min<size_t>(tiles, (size_t)per_cu * (size_t)ctx->num_cus);
    if (a.xcd_run) b = std::max<size_t>(8, b / 8 * 8);
    static const int stagger = getenv("LURKHIP_LDE_STAGGER") ? atoi(getenv("LURKHIP_LDE_STAGGER")) : 0;
    a.stagger = (per_cu >= 2 && b >= (size_t)2 * ctx->num_cus) ? stagger : 0;
    static const int dbg = getenv("LURKHIP_LDE_DBG") ? atoi(getenv("LURKHIP_LDE_DBG")) : 0;
    a.dbg = dbg;
    *blocks = (unsigned)b;
}

enum Kind { K_IN, K_OUT, K_MID, K_SMALL };

template <int LOG_R, int LOG_C>
int32_t launch_kind(lurkhip_ctx* ctx, Kind kind, LdeArgs& a, size_t tiles) {
    constexpr int NT = lde_threads<LOG_R, LOG_C>();
    unsigned blocks = 0;
    switch (kind) {
        case K_IN: {
            const size_t lds = lds_in<LOG_R, LOG_C>();
            auto kern = k_lde_in<LOG_R, LOG_C>;
            static bool once = (opt_in_lds(kern, lds), true);
            (void)once;
            size_grid(ctx, a, tiles, NT, lds, &blocks);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, ctx->stream, a);
            break;
        }
        case K_OUT: {
            const size_t lds = lds_out<LOG_R, LOG_C>();
            auto kern = k_lde_out<LOG_R, LOG_C>;
            static bool once = (opt_in_lds(kern, lds), true);
            (void)once;
            size_grid(ctx, a, tiles, NT, lds, &blocks);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, ctx->stream, a);
            break;
        }
        case K_MID: {
            const size_t lds = lds_mid<LOG_R, LOG_C>(a.n_cls);
            auto kern = k_lde_mid<LOG_R, LOG_C, false>;
            static bool once = (opt_in_lds(kern, lds_mid<LOG_R, LOG_C>(LDE_MAX_CLASSES)), true);
            (void)once;
            size_grid(ctx, a, tiles, NT, lds, &blocks);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, ctx->stream, a);
            break;
        }
        case K_SMALL: {
            const size_t lds = lds_mid<LOG_R, LOG_C>(a.n_cls);
            auto kern = k_lde_mid<LOG_R, LOG_C, true>;
            static bool once = (opt_in_lds(kern, lds_mid<LOG_R, LOG_C>(LDE_MAX_CLASSES)), true);
            (void)once;
            size_grid(ctx, a, tiles, NT, lds, &blocks);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, ctx->stream, a);
            break;
        }
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

template <int LOG_C>
int32_t launch_c(lurkhip_ctx* ctx, Kind kind, int log_r, LdeArgs& a, size_t tiles) {
    switch (log_r) {
#ifdef LDE_DEV_ONLY_R  // development: one tile height only (fast compiles for register-pressure experiments)
        case LDE_DEV_ONLY_R: return launch_kind<LDE_DEV_ONLY_R, LOG_C>(ctx, kind, a, tiles);
#else
        case 5: return launch_kind<5, LOG_C>(ctx, kind, a, tiles);
        case 6: return launch_kind<6, LOG_C>(ctx, kind, a, tiles);
        case 7: return launch_kind<7, LOG_C>(ctx, kind, a, tiles);
        case 8: return launch_kind<8, LOG_C>(ctx, kind, a, tiles);
        case 9: return launch_kind<9, LOG_C>(ctx, kind, a, tiles);
        case 10: return launch_kind<10, LOG_C>(ctx, kind, a, tiles);
#endif
    }
    return set_error(ctx, LURKHIP_ERR_INVALID_ARG, "LDE tile height 2^%d", log_r);
}

// One kernel of the LDE over all virtual columns: full chunks of `log_c_full` columns, then one narrower launch for the ragged
// rest (16 columns when it fits, else a partly idle 32-column tile).  tiles_per_chunk = row tiles (times cosets for k_out).
int32_t launch_cols(lurkhip_ctx* ctx, Kind kind, int log_r, LdeArgs a, size_t tiles_per_chunk, int log_c_full) {
    const uint32_t cw = 1u << log_c_full;
    const uint32_t span = (kind == K_OUT ? a.W_out : a.W) - a.col_base;  // virtual columns [col_base, W): col_base is a multiple of the slab width
    if (span == 0) return LURKHIP_OK;
    const uint32_t n_full = span / cw, rest = span % cw;
    if (n_full) {
        a.col0 = (int)a.col_base;
        a.n_chunks = (int)n_full;
        if (log_c_full == 5) LH_TRY(launch_c<5>(ctx, kind, log_r, a, tiles_per_chunk * n_full));
        else LH_TRY(launch_c<4>(ctx, kind, log_r, a, tiles_per_chunk * n_full));
    }
    if (rest) {
        a.col0 = (int)(a.col_base + n_full * cw);
        a.n_chunks = 1;
        if (rest <= 16) LH_TRY(launch_c<4>(ctx, kind, log_r, a, tiles_per_chunk));
        else LH_TRY(launch_c<5>(ctx, kind, log_r, a, tiles_per_chunk));
    }
    return LURKHIP_OK;
}

}  // namespace

bool lde_group_enabled() {
    static const bool on = getenv("LURKHIP_LDE_V2") == nullptr || atoi(getenv("LURKHIP_LDE_V2")) != 0;
    return on;
}

// Round 6 (opt-in, measured slower: see lde_group): the dead columns of a sparse group zero-filled by ONE launch of this kernel, a job
// per gap between the live runs, instead of being walked by the last pass -- k_lde_out's tiles are 32 columns of the OUTPUT layout, so
// a tile of 20 live and 12 dead columns costs what 32 live ones cost (2.7 ms for the permutation commitment of a fib shard where the
// live columns are 58 %); with this route the last pass walks the live columns like the first two and scatters to its entries' destinations.
namespace {
struct ZeroJob {
    uint32_t* dst;
    uint32_t dpitch, width, rows, first_block;
};
constexpr uint32_t ZERO_PER_BLOCK = 256 * 16;
uint32_t zero_job_blocks(const ZeroJob& j) {  // (the kernel's own test for 16-byte stores)
    const bool x4 = ((j.width | j.dpitch) & 3u) == 0 && ((uintptr_t)j.dst & 15u) == 0;
    const uint64_t items = (uint64_t)j.rows * (x4 ? j.width >> 2 : j.width);
    return (uint32_t)((items + ZERO_PER_BLOCK - 1) / ZERO_PER_BLOCK);
}
__global__ __launch_bounds__(256) void k_zero_jobs(const ZeroJob* __restrict__ jobs, uint32_t n_jobs) {
    uint32_t lo = 0, hi = n_jobs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs[mid].first_block <= blockIdx.x) lo = mid;
        else hi = mid;
    }
    const ZeroJob j = jobs[lo];
    // (extension-field columns: the gaps are multiples of four words on 16-byte boundaries -- 16-byte stores; else word by word)
    const bool x4 = ((j.width | j.dpitch) & 3u) == 0 && ((uintptr_t)j.dst & 15u) == 0;
    const uint32_t w = x4 ? j.width >> 2 : j.width;
    const uint32_t total = j.rows * w;  // (< 2^32: 2^21 rows of at most a few hundred columns)
    const float inv_w = 1.0f / (float)w;
    uint32_t idx = (blockIdx.x - j.first_block) * ZERO_PER_BLOCK + threadIdx.x;
    for (int k = 0; k < 16; k++, idx += 256) {
        if (idx >= total) break;
        uint32_t r = (uint32_t)((float)idx * inv_w);
        r += (r + 1u) * w <= idx ? 1u : 0u;
        r -= r * w > idx ? 1u : 0u;
        const uint32_t c = idx - r * w;
        if (x4) *reinterpret_cast<uint4*>(j.dst + (size_t)r * j.dpitch + 4u * c) = make_uint4(0u, 0u, 0u, 0u);
        else j.dst[(size_t)r * j.dpitch + c] = 0u;
    }
}
}  // namespace

bool lde_group_takes(int log_n) { return lde_group_enabled() && log_n >= LDE_GROUP_MIN_LOG_N && log_n <= LDE_GROUP_MAX_LOG_N; }

int32_t lde_group(lurkhip_ctx* ctx, int log_n, int n_mats, const uint32_t* const* evals, const uint32_t* widths, uint32_t* const* ldes,
                  const uint32_t* cls, int n_cls, const uint32_t* const (*scales)[LDE_MAX_CLASSES], bool in_canonical, bool out_canonical, const uint32_t* lde_pitches,
                  const uint32_t* src_pitches, const uint32_t* out_starts, uint32_t out_width) {
    LH_ARG(ctx, n_mats >= 1 && n_mats <= LDE_MAX_MATS && n_cls >= 1 && n_cls <= LDE_MAX_CLASSES, "LDE group shape");
    LH_ARG(ctx, log_n >= LDE_GROUP_MIN_LOG_N && log_n <= LDE_GROUP_MAX_LOG_N, "LDE group height 2^%d", log_n);
    LH_ARG(ctx, !(LDE_SLAB_X4 && out_starts), "LDE group: the quad-load variant (LDE_SLAB_X4) does not leave dead columns out");
    const NttPlan* plan = nullptr;
    LH_TRY(get_ntt_plan(ctx, log_n, &plan));
    // the zero columns of a sparse group: walked by the last pass (default), or -- LURKHIP_LDE_SPARSE_OUT=1, measured and OFF -- filled by
    // k_zero_jobs while the last pass walks the live columns only.  Alternating on the fib-mix step: `lde` 9.29 / 9.32 / 9.40 ms walked,
    // 10.37 / 10.42 / 10.33 / 10.44 compact (the zero fill with 16-byte stores or without): a tile of 32 LIVE columns spans several runs
    // whose destinations are not lined up with 128-byte lines, and what the pass saves in tiles it loses in scattered stores.
    static const bool compact_out = getenv("LURKHIP_LDE_SPARSE_OUT") != nullptr && atoi(getenv("LURKHIP_LDE_SPARSE_OUT")) != 0;
    std::vector<ZeroJob> zero_jobs;
    if (out_starts && compact_out && log_n > 10) {
        uint32_t blocks = 0;
        for (int m = 0; m < n_mats; m++) {
            const uint32_t end = out_starts[m] + widths[m], next = m + 1 < n_mats ? out_starts[m + 1] : out_width;
            LH_ARG(ctx, next >= end, "LDE group: output columns of entry %d overlap the next entry's", m);
            if (next == end) continue;
            ZeroJob j{ldes[m] + widths[m], lde_pitches ? lde_pitches[m] : widths[m], next - end, 2u << log_n, blocks};
            blocks += zero_job_blocks(j);
            zero_jobs.push_back(j);
        }
        if (zero_jobs.empty()) zero_jobs.push_back(ZeroJob{nullptr, 0, 0, 0, 0});  // (marks the compact route even when nothing is dead)
        out_starts = nullptr;  // the kernels see a dense group: entries side by side, each with its own destination
    }
    LdeArgs a{};
    a.n_mats = (uint32_t)n_mats;
    uint32_t at = 0;
    for (int m = 0; m < n_mats; m++) {
        a.src[m] = evals[m];
        a.dst[m] = ldes[m];
        a.width[m] = widths[m];
        a.dpitch[m] = lde_pitches ? lde_pitches[m] : widths[m];
        a.spitch[m] = src_pitches ? src_pitches[m] : widths[m];
        LH_ARG(ctx, a.spitch[m] >= widths[m] && a.dpitch[m] >= widths[m], "LDE group: matrix %d has a row pitch below its width", m);
        a.start[m] = at;
        a.ostart[m] = out_starts ? out_starts[m] : at;
        a.cls[m] = cls[m];
        LH_ARG(ctx, m == 0 || a.ostart[m] >= a.ostart[m - 1] + widths[m - 1], "LDE group: output columns of entry %d overlap the previous entry's", m);
        at += widths[m];
    }
    for (int m = n_mats; m < LDE_MAX_MATS; m++) a.start[m] = a.ostart[m] = 0xffffffffu;
    a.W = at;
    a.W_out = out_starts ? out_width : at;
    const bool dead_columns = a.W_out != a.W;
    LH_ARG(ctx, a.W_out >= a.W && (n_mats == 0 || a.W_out >= a.ostart[n_mats - 1] + widths[n_mats - 1]), "LDE group: output width below its entries");
    LH_ARG(ctx, !dead_columns || log_n > 10, "LDE group: dead columns are for the three-pass shapes");
    a.n_cls = (uint32_t)n_cls;
    for (int q = 0; q < 2; q++)
        for (int c = 0; c < n_cls; c++) a.scale[q][c] = scales[q][c];
    a.tw_inv = plan->tw_inv;
    a.tw_fwd = plan->tw_fwd;
    a.log_n = log_n;
    a.in_canonical = in_canonical ? 1 : 0;
    a.out_canonical = out_canonical ? 1 : 0;
    a.slabs = (a.W + (1u << SLAB_LOG_W) - 1) >> SLAB_LOG_W;
    // 16-byte accesses: the slabs always; the matrices' side when every matrix of the group is 16-byte aligned and a multiple of
    // four columns wide (the extension-field matrices of the permutation and quotient rounds always are).  LURKHIP_LDE_X4 masks the bits.
    static const int x4_mask = getenv("LURKHIP_LDE_X4") ? atoi(getenv("LURKHIP_LDE_X4")) : 0;
    bool mats_x4 = true;
    for (int m = 0; m < n_mats; m++)
        mats_x4 = mats_x4 && widths[m] % 4 == 0 && a.dpitch[m] % 4 == 0 && a.spitch[m] % 4 == 0 && ((uintptr_t)evals[m] & 15u) == 0 && ((uintptr_t)ldes[m] & 15u) == 0;
    a.x4 = dead_columns ? 0 : ((mats_x4 ? 1 : 0) & x4_mask);
    if (log_n <= 10) {
        a.r1 = 0;
        a.r2 = log_n;
        // one tile per column chunk: 16-column tiles for the tallest (64 KiB of LDS), so that more than W / 32 workgroups exist
        return launch_cols(ctx, K_SMALL, log_n, a, 1, log_n >= 10 ? 4 : 5);
    }
    a.r1 = (log_n + 1) / 2;
    a.r2 = log_n - a.r1;
    const size_t slab_words = (size_t)a.slabs << (log_n + SLAB_LOG_W);
    void *A = nullptr, *B = nullptr;
    LH_TRY(pool_alloc(ctx, std::max<size_t>(slab_words, 8) * 4, &A));
    int32_t st = pool_alloc(ctx, slab_words * 8 + ((size_t)4 << (log_n + SLAB_LOG_W)), &B);  // two cosets + the spare slab of k_lde_out's padding lanes
    if (st != LURKHIP_OK) {
        pool_release(ctx, A);
        return st;
    }
    a.A = (uint32_t*)A;
    a.B = (uint32_t*)B;
    // first / last pass: LURKHIP_LDE_IO_LOG_C = 4 gives 2^10-row tiles 16 columns (two workgroups per CU) -- A/B hook
    static const int io_log_c = getenv("LURKHIP_LDE_IO_LOG_C") ? std::max(4, std::min(5, atoi(getenv("LURKHIP_LDE_IO_LOG_C")))) : 5;
    const int io_c = a.r1 >= 10 ? io_log_c : 5;
    // LURKHIP_LDE_SLAB_BATCH = k (A/B hook): the three kernels run over k slabs at a time, so that a batch's intermediates
    // (3 k N 128 bytes) may stay in the 256 MiB Infinity Cache between the kernel that writes them and the one that reads them
    static const int slab_batch = getenv("LURKHIP_LDE_SLAB_BATCH") ? atoi(getenv("LURKHIP_LDE_SLAB_BATCH")) : 0;
    const uint32_t W_all = a.W;
    const uint32_t step = (slab_batch > 0 && !dead_columns) ? (uint32_t)slab_batch << SLAB_LOG_W : std::max(W_all, 1u);
    for (uint32_t c0 = 0; c0 < std::max(W_all, 1u) && st == LURKHIP_OK; c0 += step) {
        LdeArgs b = a;
        b.col_base = c0;
        b.W = std::min(W_all, c0 + step);
        if (!dead_columns) b.W_out = b.W;
        st = launch_cols(ctx, K_IN, a.r1, b, (size_t)1 << a.r2, io_c);
        // the fused pass keeps two workgroups on a CU: 2^10-row tiles are 16 columns wide
        static const int mid_log_c = getenv("LURKHIP_LDE_MID_LOG_C") ? std::max(4, std::min(5, atoi(getenv("LURKHIP_LDE_MID_LOG_C")))) : 5;  // (A/B hook)
        if (st == LURKHIP_OK) st = launch_cols(ctx, K_MID, a.r2, b, (size_t)1 << a.r1, a.r2 >= 10 ? 4 : mid_log_c);
        if (st == LURKHIP_OK) st = launch_cols(ctx, K_OUT, a.r1, b, (size_t)2 << a.r2, io_c);
    }
    pool_release(ctx, A);  // stream-ordered
    pool_release(ctx, B);
    if (st == LURKHIP_OK && !zero_jobs.empty() && zero_jobs[0].dst != nullptr) {
        void* tbl = nullptr;
        LH_TRY(pool_alloc(ctx, zero_jobs.size() * sizeof(ZeroJob), &tbl));
        static_assert(sizeof(ZeroJob) % 4 == 0, "uploaded as words");
        st = upload_words(ctx, (uint32_t*)tbl, (const uint32_t*)zero_jobs.data(), zero_jobs.size() * sizeof(ZeroJob) / 4);
        const ZeroJob& last = zero_jobs.back();
        const uint32_t blocks = last.first_block + zero_job_blocks(last);
        if (st == LURKHIP_OK) hipLaunchKernelGGL(k_zero_jobs, dim3(blocks), dim3(256), 0, ctx->stream, (const ZeroJob*)tbl, (uint32_t)zero_jobs.size());
        pool_release(ctx, tbl);
        if (st == LURKHIP_OK && hipGetLastError() != hipSuccess) st = set_error(ctx, LURKHIP_ERR_HIP, "k_zero_jobs launch failed");
    }
    return st;
}

}  // namespace lurkhip
