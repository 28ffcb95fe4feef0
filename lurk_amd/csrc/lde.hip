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
// Index arithmetic: tools/lde_model.py is this file's decomposition in numpy, checked against oracle/stark.py.
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

struct LdeArgs {
    const uint32_t* src[LDE_MAX_MATS];  // N x width[m], row-major
    uint32_t* dst[LDE_MAX_MATS];        // 2N x width[m]: block q = coset q, rows in the DIF's (bit-reversed) order
    uint32_t width[LDE_MAX_MATS];
    uint32_t start[LDE_MAX_MATS];       // virtual column of the matrix's first column (2^32 - 1 for unused slots)
    uint32_t cls[LDE_MAX_MATS];         // shift class of the matrix
    const uint32_t* scale[2][LDE_MAX_CLASSES];  // per coset and class: s_q^k / N, k < N
    const uint32_t *tw_inv, *tw_fwd;    // N/2 powers of the inverse / forward size-N root
    uint32_t* A;                        // inverse first pass -> fused pass: [slabs][N][32]
    uint32_t* B;                        // fused pass -> forward last pass: [2 cosets][slabs][N][32]
    uint32_t n_mats, W, n_cls, slabs;
    uint32_t col_base;                  // host side: first virtual column of the launches (a slab batch)
    int log_n, r1, r2;
    int col0, n_chunks;                 // this launch: chunks of (1 << LOG_C) virtual columns from col0 on
    uint32_t n_tiles, xcd_run;
    int in_canonical, out_canonical;    // convert the caller's words on the first load / the last store
    int stagger;                        // s_sleep(127) rounds the second half of the grid waits before its first tile (A/B hook)
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

// Stage group 1: register j of slot s is tile row s + S j; in-register bit g is tile-row bit LOG_S + g.  The butterfly of rows
// (t, t + 2^b) takes tw[(1 << b) + (t mod 2^b)] = tw[s + S ((1 << g) + (j mod 2^g))]: thirty-one entries per thread at
// compile-time offsets from tw + s.
template <int LOG_R>
__device__ __forceinline__ void group1(uint32_t (&x)[Geo<LOG_R>::U], const uint32_t* __restrict__ tw_s) {
    using G = Geo<LOG_R>;
#pragma unroll
    for (int g = G::LOG_U - 1; g >= 0; g--) {
#pragma unroll
        for (int j = 0; j < G::U; j++) {
            if (j & (1 << g)) continue;
            const int m = (1 << g) + (j & ((1 << g) - 1));
            bfly(x[j], x[j | (1 << g)], tw_s[G::S * m]);
        }
    }
}
// Stage group 2: register j of slot s is tile row 32 s + j; stages LOG_S-1 .. 0 on in-register bits, the twiddle
// tw[(1 << g) + (j mod 2^g)] is the same for every thread: read once into scalar registers.  ONE0: the tile's lowest
// row bits are the transform's (a last pass) -- entry (1 << g) + 0 is one, no product.
template <int LOG_R, bool ONE0>
__device__ __forceinline__ void group2(uint32_t (&x)[Geo<LOG_R>::U], const uint32_t* __restrict__ tw) {
    using G = Geo<LOG_R>;
    if constexpr (G::LOG_S > 0) {
        uint32_t tws[G::S];
#pragma unroll
        for (int i = 1; i < G::S; i++) tws[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)tw[i]);
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
template <int U>
__device__ __forceinline__ void walk_load(uint32_t (&x)[U], const uint32_t* __restrict__ p, size_t stride) {
#pragma unroll
    for (int j = 0; j < U; j++) {
        x[j] = *p;
        p += stride;
    }
}
template <int U>
__device__ __forceinline__ void walk_store(const uint32_t (&x)[U], uint32_t* __restrict__ p, size_t stride) {
#pragma unroll
    for (int j = 0; j < U; j++) {
        *p = x[j];
        p += stride;
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
    uint32_t w, cls, start, pad;
};
constexpr int DESC_WORDS = LDE_MAX_MATS * (int)(sizeof(MatDesc) / 4);
__device__ __forceinline__ void stage_descs(const LdeArgs& a, MatDesc* __restrict__ descs) {
    if (threadIdx.x < LDE_MAX_MATS) {
        const int m = (int)threadIdx.x;
        MatDesc d;
        d.src = a.src[m];
        d.dst = a.dst[m];
        d.w = a.width[m];
        d.cls = a.cls[m];
        d.start = a.start[m];
        d.pad = 0;
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
    r.cls = d.cls;
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

    uint32_t x[G::U], y[G::U], tv[TWN];
    ColRef ref;
    uint32_t lo = 0, vc = 0;
    auto fetch = [&](uint32_t it) {
        const uint32_t id = walk.first + it * walk.step;
        lo = id / (uint32_t)a.n_chunks;
        vc = (uint32_t)a.col0 + (id - lo * (uint32_t)a.n_chunks) * C + (uint32_t)c;
        ref = locate_col(a, descs, vc);
        if (ref.valid) walk_load<G::U>(x, ref.src + (((size_t)s << r2) | lo) * ref.w, ((size_t)G::S << r2) * ref.w);
        else zero_rows<G::U>(x);
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            const int idx = (int)threadIdx.x + i * NT;
            tv[i] = idx < G::R ? tile_twiddle(a.tw_inv, idx, a.log_n, r2, lo) : 0u;
        }
    };
    fetch(0);
    for (uint32_t it = 0; it < walk.count; it++) {
        uint32_t* __restrict__ tw = twb + (it & 1u) * G::R;
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            const int idx = (int)threadIdx.x + i * NT;
            if (idx < G::R) tw[idx] = tv[i];
        }
        const size_t out_off = slab_off(a, vc);
        const uint32_t cur_lo = lo;
        __syncthreads();  // the table is complete; every thread has left the previous tile
        if (a.in_canonical) {
#pragma unroll
            for (int j = 0; j < G::U; j++) x[j] = bb::to_monty(x[j]);
        }
        group1<LOG_R>(x, tw + s);
        if constexpr (G::LOG_S > 0) {
            tile_write<LOG_R, LOG_C>(tile, s, c, x);
            fetch(it + 1 < walk.count ? it + 1 : it);  // (the last tile re-reads itself: no branch around memory operations)
            __syncthreads();
            tile_read<LOG_R, LOG_C>(tile, s, c, y);
            group2<LOG_R, false>(y, tw);
        } else {
#pragma unroll
            for (int j = 0; j < G::U; j++) y[j] = x[j];
            fetch(it + 1 < walk.count ? it + 1 : it);  // (the last tile re-reads itself: no branch around memory operations)
        }
        walk_store<G::U>(y, a.A + out_off + (((((size_t)(G::U * s)) << r2) | cur_lo) << SLAB_LOG_W), (size_t)1 << (r2 + SLAB_LOG_W));
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

    uint32_t x[G::U], y[G::U];
    uint32_t q = 0, hi = 0, vc = 0;
    auto fetch = [&](uint32_t it) {
        const uint32_t id = walk.first + it * walk.step;
        const uint32_t blk = id / (uint32_t)a.n_chunks;
        vc = (uint32_t)a.col0 + (id - blk * (uint32_t)a.n_chunks) * C + (uint32_t)c;
        q = blk >> r2;
        hi = blk & ((1u << r2) - 1u);
        const uint32_t* __restrict__ in = a.B + ((size_t)q * a.slabs << (a.log_n + SLAB_LOG_W)) + slab_off(a, vc);
        walk_load<G::U>(x, in + ((((size_t)hi << LOG_R) | (size_t)s) << SLAB_LOG_W), (size_t)G::S << SLAB_LOG_W);
    };
    fetch(0);
    for (uint32_t it = 0; it < walk.count; it++) {
        const ColRef ref = locate_col(a, descs, vc);
        const size_t row0 = ((size_t)q << a.log_n) | ((size_t)hi << LOG_R);
        __syncthreads();  // (first tile: the table is complete) every thread has left the previous tile
        group1<LOG_R>(x, tw + s);
        if constexpr (G::LOG_S > 0) {
            tile_write<LOG_R, LOG_C>(tile, s, c, x);
            fetch(it + 1 < walk.count ? it + 1 : it);  // (the last tile re-reads itself: no branch around memory operations)
            __syncthreads();
            tile_read<LOG_R, LOG_C>(tile, s, c, y);
            group2<LOG_R, true>(y, tw);
        } else {
#pragma unroll
            for (int j = 0; j < G::U; j++) y[j] = x[j];
            fetch(it + 1 < walk.count ? it + 1 : it);  // (the last tile re-reads itself: no branch around memory operations)
        }
        if (a.out_canonical) {
#pragma unroll
            for (int j = 0; j < G::U; j++) y[j] = bb::from_monty(y[j]);
        }
        if (ref.valid) walk_store<G::U>(y, ref.dst + (row0 + (size_t)(G::U * s)) * ref.w, (size_t)ref.w);
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
        if (DIRECT && a.in_canonical) {
#pragma unroll
            for (int j = 0; j < G::U; j++) x[j] = bb::to_monty(x[j]);
        }
        group1<LOG_R>(x, twi + s);
        if constexpr (G::LOG_S > 0) {
            tile_write<LOG_R, LOG_C>(tile, s, c, x);
            __syncthreads();  // B1
            tile_read<LOG_R, LOG_C>(tile, s, c, coef);
            group2<LOG_R, true>(coef, twi);
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
            group1<LOG_R>(x, twf + s2);
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
                group2<LOG_R, false>(z, twf);
            } else {
                if (q == 0) __syncthreads();  // coset 1's scales are complete
            }
            if constexpr (DIRECT) {
                if (a.out_canonical) {
#pragma unroll
                    for (int j = 0; j < G::U; j++) z[j] = bb::from_monty(z[j]);
                }
                if (cur.valid) walk_store<G::U>(z, cur.dst + (((size_t)q << a.log_n) + (size_t)(G::U * s)) * cur.w, (size_t)cur.w);
            } else {
                uint32_t* __restrict__ out = a.B + ((size_t)q * a.slabs << (a.log_n + SLAB_LOG_W)) + slab_off(a, cur_vc);
                walk_store<G::U>(z, out + (((((size_t)(G::U * s)) << r1) | cur_lo2) << SLAB_LOG_W), (size_t)1 << (r1 + SLAB_LOG_W));
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
    const uint32_t span = a.W - a.col_base;  // virtual columns [col_base, W): col_base is a multiple of the slab width
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

bool lde_group_takes(int log_n) { return lde_group_enabled() && log_n >= LDE_GROUP_MIN_LOG_N && log_n <= LDE_GROUP_MAX_LOG_N; }

int32_t lde_group(lurkhip_ctx* ctx, int log_n, int n_mats, const uint32_t* const* evals, const uint32_t* widths, uint32_t* const* ldes,
                  const uint32_t* cls, int n_cls, const uint32_t* const (*scales)[LDE_MAX_CLASSES], bool in_canonical, bool out_canonical) {
    LH_ARG(ctx, n_mats >= 1 && n_mats <= LDE_MAX_MATS && n_cls >= 1 && n_cls <= LDE_MAX_CLASSES, "LDE group shape");
    LH_ARG(ctx, log_n >= LDE_GROUP_MIN_LOG_N && log_n <= LDE_GROUP_MAX_LOG_N, "LDE group height 2^%d", log_n);
    const NttPlan* plan = nullptr;
    LH_TRY(get_ntt_plan(ctx, log_n, &plan));
    LdeArgs a{};
    a.n_mats = (uint32_t)n_mats;
    uint32_t at = 0;
    for (int m = 0; m < n_mats; m++) {
        a.src[m] = evals[m];
        a.dst[m] = ldes[m];
        a.width[m] = widths[m];
        a.start[m] = at;
        a.cls[m] = cls[m];
        at += widths[m];
    }
    for (int m = n_mats; m < LDE_MAX_MATS; m++) a.start[m] = 0xffffffffu;
    a.W = at;
    a.n_cls = (uint32_t)n_cls;
    for (int q = 0; q < 2; q++)
        for (int c = 0; c < n_cls; c++) a.scale[q][c] = scales[q][c];
    a.tw_inv = plan->tw_inv;
    a.tw_fwd = plan->tw_fwd;
    a.log_n = log_n;
    a.in_canonical = in_canonical ? 1 : 0;
    a.out_canonical = out_canonical ? 1 : 0;
    a.slabs = (a.W + (1u << SLAB_LOG_W) - 1) >> SLAB_LOG_W;
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
    LH_TRY(pool_alloc(ctx, slab_words * 4, &A));
    int32_t st = pool_alloc(ctx, slab_words * 8, &B);
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
    const uint32_t step = slab_batch > 0 ? (uint32_t)slab_batch << SLAB_LOG_W : W_all;
    for (uint32_t c0 = 0; c0 < W_all && st == LURKHIP_OK; c0 += step) {
        LdeArgs b = a;
        b.col_base = c0;
        b.W = std::min(W_all, c0 + step);
        st = launch_cols(ctx, K_IN, a.r1, b, (size_t)1 << a.r2, io_c);
        // the fused pass keeps two workgroups on a CU: 2^10-row tiles are 16 columns wide
        if (st == LURKHIP_OK) st = launch_cols(ctx, K_MID, a.r2, b, (size_t)1 << a.r1, a.r2 >= 10 ? 4 : 5);
        if (st == LURKHIP_OK) st = launch_cols(ctx, K_OUT, a.r1, b, (size_t)2 << a.r2, io_c);
    }
    pool_release(ctx, A);  // stream-ordered
    pool_release(ctx, B);
    return st;
}

}  // namespace lurkhip
