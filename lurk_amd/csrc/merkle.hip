// Poseidon2-width-16 Merkle tree over row-major matrices (the MMCS of the commit stage).
//
// Replaces (S1 commit in SURVEY.md 8a; third-party, source absent from /root/reference):
//   p3 FieldMerkleTreeMmcs<.., PaddingFreeSponge<Perm,16,8,8>, TruncatedPermutation<Perm,2,8,16>, 8>::commit
//   [UPSTREAM-RECALL, Plonky3 @ a0b92870]:
//   - leaf digest of row i = sponge over the concatenation of row i of every matrix of maximal height
//     (rate 8, overwrite-absorb, permute after every full or final partial chunk, squeeze 8 lanes);
//   - parent = first 8 lanes of Perm(left || right);
//   - matrices of smaller height h are injected at the level with h nodes:
//     node = compress(compress(left, right), sponge(row i of those matrices)).
// The width-16 permutation parameters come from a per-ctx device table (P16Params) so a caller can
// install sphinx's RC_16_30 / DiffusionMatrixBabyBear constants, which are not in /root/reference;
// the default is the reference's own BabyBearConfig16 (src/poseidon/config.rs:190-199).
//
// One row (or one tree node) per lane; the 16-lane sponge state stays in VGPRs; the concatenated
// row is described by a uniform LeafCol table read through the scalar cache, so the absorb loop
// indexes the state with compile-time lane numbers (no scratch).  VALU-bound: ceil(w/8)
// permutations (~5.0 k int32 instructions each) per w*4 bytes read.
#include <string.h>

#include <string>
#include <utility>

#include "commit.h"
#include "p16_coop.h"
#include "poseidon2_dev.h"

namespace lurkhip {

namespace {

constexpr int MBLOCK = 256;
// a height group of at most this many rows, each at least this many permutations long, is hashed sixteen lanes to the row
constexpr uint32_t SPONGE_COOP_MIN_PERMS = 16;
constexpr uint64_t SPONGE_COOP_MAX_ROWS = 4096;
// levels of at most this many parents run lane-cooperatively (above it one permutation per lane already fills the SIMDs)
constexpr size_t COOP_MAX_PARENTS = MERKLE_COOP_MAX_PARENTS;

// (inlined at its three call sites of k_level on purpose: as a call the state goes through scratch and the step loses 3 ms)
__device__ __forceinline__ void perm16(uint32_t (&s)[16], const P16Params* __restrict__ p) {
    p2::NoRecord rec;
    p2::permute_core<16>(s, p->rounds_p, p->ext_rc, p->int_rc, p->diag, p->ext_rc_mp, p->int_rc_mp, p->diag_c, rec, p->sum_mult_c);
}

// Row words are read with global (not flat) loads; a dword-aligned 16-byte global load is legal on gfx950.
typedef const __attribute__((address_space(1))) uint32_t* gwords;
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) u32x4_a4* gquads;

// The (up to) eight words chunk `g` of the concatenated row `row` absorbs.  When the eight columns are consecutive columns of
// one matrix -- every chunk except those straddling two matrices and the ragged last one -- they are 32 contiguous bytes:
// two descriptor loads, one address, two 16-byte loads; otherwise one descriptor and one load per word.
__device__ __forceinline__ void load_chunk(uint32_t (&v)[8], const LeafCol* __restrict__ cols, uint32_t g, uint32_t total_w, size_t row) {
    if (g + 8 <= total_w) {
        const LeafCol d0 = cols[g], d7 = cols[g + 7];
        if (d0.base == d7.base && d7.col == d0.col + 7) {
            gquads a = (gquads)((gwords)d0.base + (row * d0.width + d0.col));
            const u32x4_a4 lo = a[0], hi = a[1];
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
            v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (g + j < total_w) {
            const LeafCol d = cols[g + j];
            v[j] = ((gwords)d.base)[row * d.width + d.col];
        }
    }
}

// sponge over the uniform column table for row `row`; state must be zero on entry.  The words of chunk g + 1 are requested
// before the permutation of chunk g runs, so a wave never waits for its row with nothing else to do.
__device__ __forceinline__ void sponge_row(uint32_t (&s)[16], const P16Params* __restrict__ p,
                                           const LeafCol* __restrict__ cols, uint32_t total_w, size_t row) {
    uint32_t nxt[8];
    load_chunk(nxt, cols, 0, total_w, row);
    for (uint32_t g = 0; g < total_w; g += 8) {
        if (g + 8 <= total_w) {
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] = nxt[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (g + j < total_w) s[j] = nxt[j];
        }
        if (g + 8 < total_w) load_chunk(nxt, cols, g + 8, total_w, row);
        perm16(s, p);
    }
}

__global__ __launch_bounds__(MBLOCK) void k_leaves(const P16Params* __restrict__ p, const LeafCol* __restrict__ cols,
                                                    uint32_t total_w, size_t n_rows, uint32_t* __restrict__ out) {
    size_t row = (size_t)blockIdx.x * MBLOCK + threadIdx.x;
    if (row >= n_rows) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0;
    sponge_row(s, p, cols, total_w, row);
    uint4* dst = reinterpret_cast<uint4*>(out + row * 8);
    dst[0] = make_uint4(s[0], s[1], s[2], s[3]);
    dst[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// the permutation alone, one state per lane (lurkhip_perm16_dev)
__global__ __launch_bounds__(MBLOCK) void k_perm16_states(const P16Params* __restrict__ p, const uint32_t* __restrict__ in,
                                                           uint32_t* __restrict__ out, size_t n, bool canonical) {
    const size_t i = (size_t)blockIdx.x * MBLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t s[16];
    const uint4* src = reinterpret_cast<const uint4*>(in + i * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 v = src[k];
        s[4 * k] = v.x, s[4 * k + 1] = v.y, s[4 * k + 2] = v.z, s[4 * k + 3] = v.w;
    }
    if (canonical) {
#pragma unroll
        for (int k = 0; k < 16; k++) s[k] = bb::to_monty(s[k]);
    }
    perm16(s, p);
    if (canonical) {
#pragma unroll
        for (int k = 0; k < 16; k++) s[k] = bb::from_monty(s[k]);
    }
    uint4* dst = reinterpret_cast<uint4*>(out + i * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = make_uint4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
}

__device__ __forceinline__ void load_pair(const uint32_t* __restrict__ children, size_t i, uint32_t (&s)[16]) {
    const uint4* src = reinterpret_cast<const uint4*>(children + i * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 v = src[k];
        s[4 * k] = v.x;
        s[4 * k + 1] = v.y;
        s[4 * k + 2] = v.z;
        s[4 * k + 3] = v.w;
    }
}

__global__ __launch_bounds__(MBLOCK) void k_level(const P16Params* __restrict__ p, const uint32_t* __restrict__ children,
                                                   size_t n_parents, const LeafCol* __restrict__ inject_cols,
                                                   uint32_t inject_w, uint32_t* __restrict__ parents) {
    size_t i = (size_t)blockIdx.x * MBLOCK + threadIdx.x;
    if (i >= n_parents) return;
    uint32_t s[16];
    load_pair(children, i, s);
    perm16(s, p);
    if (inject_cols) {
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = 0;
        sponge_row(t, p, inject_cols, inject_w, i);
#pragma unroll
        for (int k = 0; k < 8; k++) s[8 + k] = t[k];
        perm16(s, p);
    }
    uint4* dst = reinterpret_cast<uint4*>(parents + i * 8);
    dst[0] = make_uint4(s[0], s[1], s[2], s[3]);
    dst[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// One row of a few very long ones, by 16 lanes (p16_coop.h): lanes 0..7 absorb.  A row of the hash chips' group of a fib machine
// is 2067 columns -- 259 permutations one after the other -- and there are 512 such rows: one row per lane that is eight waves
// each walking a chain of 259 full-latency permutations (13.6 us alone on a SIMD, 50 us among five other waves), 3.4 ms of a
// small proof's 12 and the last 0.8 ms of the 2^20-row shard's main sponge launch, alone on the chip.  Sixteen lanes per row
// the chain is 259 cooperative permutations (3.5 us alone).  The word of chunk c + 1 and the descriptor of chunk c + 2 are
// requested before the permutation of chunk c.
__device__ __forceinline__ void coop_sponge_row(const P16Params* __restrict__ p, const LeafCol* __restrict__ cols, uint32_t w, size_t row,
                                                int j, bool store, uint32_t* __restrict__ out) {
    const bool absorbs = j < 8;
    auto word = [&](const LeafCol& d) { return ((gwords)d.base)[row * d.width + d.col]; };
    uint32_t t = 0, nxt = 0;
    LeafCol dn{};
    if (absorbs && (uint32_t)j < w) nxt = word(cols[j]);
    if (absorbs && 8u + j < w) dn = cols[8 + j];
    for (uint32_t c0 = 0; c0 < w; c0 += 8) {
        if (absorbs && c0 + j < w) t = nxt;
        if (absorbs && c0 + 8 + j < w) nxt = word(dn);
        if (absorbs && c0 + 16 + j < w) dn = cols[c0 + 16 + j];
        t = coop_perm16(t, p, j);
    }
    if (store && absorbs) out[row * 8 + j] = t;
}

// several height groups' row sponges in one grid (merkle_row_sponges): the block's group by its first_block range
#ifndef LURK_SPONGE_WAVES_PER_EU
#define LURK_SPONGE_WAVES_PER_EU 0
#endif
#if LURK_SPONGE_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(LURK_SPONGE_WAVES_PER_EU, LURK_SPONGE_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(MBLOCK) void k_row_sponges(const P16Params* __restrict__ p, SpongeGroups g) {
    int k = 0;
    while (k + 1 < g.n && blockIdx.x >= g.first_block[k + 1]) k++;
    if (g.coop[k]) {  // (uniform per block)
        const size_t r = (size_t)(blockIdx.x - g.first_block[k]) * (MBLOCK / 16) + (threadIdx.x >> 4);
        const bool live = r < g.n_rows[k];
        // every lane of a group runs the permutations (DPP reads need all 16 active): groups past the end redo row 0
        coop_sponge_row(p, g.cols[k], g.total_w[k], live ? r : 0, threadIdx.x & 15, live, g.out[k]);
        return;
    }
    const size_t row = (size_t)(blockIdx.x - g.first_block[k]) * MBLOCK + threadIdx.x;
    if (row >= g.n_rows[k]) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0;
    sponge_row(s, p, g.cols[k], g.total_w[k], row);
    uint4* dst = reinterpret_cast<uint4*>(g.out[k] + row * 8);
    dst[0] = make_uint4(s[0], s[1], s[2], s[3]);
    dst[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// a level whose injected rows were hashed ahead (k_row_sponges): two permutations per parent at most
__global__ __launch_bounds__(MBLOCK) void k_level_digests(const P16Params* __restrict__ p, const uint32_t* __restrict__ children,
                                                           size_t n_parents, const uint32_t* __restrict__ inject, uint32_t* __restrict__ parents) {
    size_t i = (size_t)blockIdx.x * MBLOCK + threadIdx.x;
    if (i >= n_parents) return;
    uint32_t s[16];
    load_pair(children, i, s);
    perm16(s, p);
    if (inject) {
        const uint4* src = reinterpret_cast<const uint4*>(inject + i * 8);
        const uint4 a = src[0], b = src[1];
        s[8] = a.x; s[9] = a.y; s[10] = a.z; s[11] = a.w;
        s[12] = b.x; s[13] = b.y; s[14] = b.z; s[15] = b.w;
        perm16(s, p);
    }
    uint4* dst = reinterpret_cast<uint4*>(parents + i * 8);
    dst[0] = make_uint4(s[0], s[1], s[2], s[3]);
    dst[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// ---- lane-cooperative levels: 16 lanes per parent (p16_coop.h).  A level of a few thousand parents run one per lane is a
// handful of waves, each a ~5 k-instruction dependent chain (~13 us whatever the level's size); spread over 16 lanes the chain
// is ~1 k instructions and the level fills the CUs.

// sponge of row `row` of the injected matrices, then compress(node, sponge): lanes 0..7 of the group hold the node on entry
__device__ __forceinline__ uint32_t coop_inject(uint32_t x, const P16Params* __restrict__ p, const LeafCol* __restrict__ cols,
                                                uint32_t w, size_t row, int j) {
    uint32_t t = 0;
    for (uint32_t c0 = 0; c0 < w; c0 += 8) {
        if (j < 8 && c0 + j < w) {
            const LeafCol d = cols[c0 + j];
            t = d.base[row * d.width + d.col];
        }
        t = coop_perm16(t, p, j);
    }
    const uint32_t up = dpp<DPP_ROW_ROR8>(t);  // lane j >= 8 reads lane j - 8
    return coop_perm16(j < 8 ? x : up, p, j);
}

// the same when the row's sponge was hashed ahead: lanes 8..15 take its digest
__device__ __forceinline__ uint32_t coop_inject_digest(uint32_t x, const P16Params* __restrict__ p, const uint32_t* __restrict__ dig, size_t row,
                                                       int j) {
    const uint32_t d = dig[row * 8 + (size_t)(j & 7)];
    return coop_perm16(j < 8 ? x : d, p, j);
}

// leaves of a short tree (the later FRI layers, small commitments): one row per 16-lane group, lanes 0..7 absorb
__global__ __launch_bounds__(MBLOCK) void k_leaves_coop(const P16Params* __restrict__ p, const LeafCol* __restrict__ cols, uint32_t total_w,
                                                         size_t n_rows, uint32_t* __restrict__ out) {
    const int j = threadIdx.x & 15;
    const size_t g = ((size_t)blockIdx.x * MBLOCK + threadIdx.x) >> 4;
    const size_t row = g < n_rows ? g : 0;
    uint32_t t = 0;
    for (uint32_t c0 = 0; c0 < total_w; c0 += 8) {
        if (j < 8 && c0 + j < total_w) {
            const LeafCol d = cols[c0 + j];
            t = d.base[row * d.width + d.col];
        }
        t = coop_perm16(t, p, j);
    }
    if (g < n_rows && j < 8) out[g * 8 + j] = t;
}

__global__ __launch_bounds__(MBLOCK) void k_level_coop(const P16Params* __restrict__ p, const uint32_t* __restrict__ children,
                                                        size_t n_parents, const LeafCol* __restrict__ inject_cols,
                                                        uint32_t inject_w, uint32_t* __restrict__ parents) {
    const int j = threadIdx.x & 15;
    const size_t g = ((size_t)blockIdx.x * MBLOCK + threadIdx.x) >> 4;
    // every lane of a group runs the permutation (DPP reads need all 16 active): groups past the end redo parent 0
    const size_t gg = g < n_parents ? g : 0;
    uint32_t x = coop_perm16(children[gg * 16 + j], p, j);
    if (inject_cols) x = coop_inject(x, p, inject_cols, inject_w, gg, j);
    if (g < n_parents && j < 8) parents[g * 8 + j] = x;
}

// Several consecutive cooperative levels in one launch: workgroup b owns the 2^levels children [b << levels, (b + 1) << levels)
// of the first level and collapses them to one node, storing every level where the one-launch-per-level path stores it (the
// levels lie back to back behind `children`, n_children / 2 parents first).  A cooperative level is one permutation's latency
// whatever its width; launched one by one each also paid a dependent launch (14 us per level measured, 6 of them the permutation):
// the 8 levels between 16384 nodes and the one-workgroup top are two launches instead of eight.
// blockDim = 16 lanes x 2^(levels - 1) groups; inj as for k_top: entry t = the rows absorbed into the parents of step t.
__global__ __launch_bounds__(256) void k_levels_coop(const P16Params* __restrict__ p, uint32_t* __restrict__ children, size_t n_children,
                                                      int levels, TopInject inj) {
    const int j = threadIdx.x & 15;
    const uint32_t group = threadIdx.x >> 4;
    uint32_t* cur = children;
    size_t len = n_children;              // nodes of the current level in the whole tree
    uint32_t local = 1u << levels;        // ... of them in this workgroup's subtree
    for (int t = 0; t < levels; t++) {
        const uint32_t half = local >> 1;
        uint32_t* next = cur + len * 8;
        const LeafCol* icols = inj.cols[t];
        const size_t first = (size_t)blockIdx.x * half;  // this subtree's first parent of the level
        // every lane of a group runs the permutation (DPP reads need all 16 active): groups past the subtree redo its first parent
        // (a wave whose four groups all lie past the level skips it: with four workgroups per CU the idle waves' permutations
        // took issue slots from the working ones)
        if (((threadIdx.x & ~63u) >> 4) < half) {
            const size_t g = first + (group < half ? group : 0);
            uint32_t x = coop_perm16(cur[g * 16 + j], p, j);
            if (inj.dig[t]) x = coop_inject_digest(x, p, inj.dig[t], g, j);
            else if (icols) x = coop_inject(x, p, icols, inj.w[t], g, j);
            if (group < half && j < 8) next[g * 8 + j] = x;
        }
        // same-workgroup hand-off through global memory
        __threadfence_block();
        __syncthreads();
        cur = next;
        len >>= 1;
        local = half;
    }
}

// Collapse n (<= 2048, power of two) nodes to the root in one workgroup.  The levels are stored back
// to back after `level_base` exactly as the multi-launch path would store them.  Wide levels without injected matrices
// run one permutation per lane; from 128 parents down, and wherever rows are injected, the permutations are lane-cooperative.
__global__ __launch_bounds__(1024) void k_top(const P16Params* __restrict__ p, uint32_t* __restrict__ level_base, size_t n,
                                              TopInject inj) {
    uint32_t* cur = level_base;
    size_t len = n;
    int lvl = 0;
    while (len > 1) {
        size_t half = len >> 1;
        uint32_t* next = cur + len * 8;
        const LeafCol* icols = inj.cols[lvl];
        if (half > 128 && !icols && !inj.dig[lvl]) {
            for (size_t i = threadIdx.x; i < half; i += blockDim.x) {
                uint32_t s[16];
                load_pair(cur, i, s);
                perm16(s, p);
                uint4* dst = reinterpret_cast<uint4*>(next + i * 8);
                dst[0] = make_uint4(s[0], s[1], s[2], s[3]);
                dst[1] = make_uint4(s[4], s[5], s[6], s[7]);
            }
        } else {
            const int j = threadIdx.x & 15;
            for (size_t g0 = 0; g0 < half; g0 += blockDim.x / 16) {
                const size_t g = g0 + (threadIdx.x >> 4);
                // a wave whose four groups all lie past the level has nothing to do: sixteen waves redoing parent 0 shared the
                // SIMDs four to one and made every level of the top cost four permutations' issue time (6.3 us against 3.5)
                if (g0 + ((threadIdx.x & ~63u) >> 4) >= half) continue;
                const size_t gg = g < half ? g : 0;
                uint32_t x = coop_perm16(cur[gg * 16 + j], p, j);
                if (inj.dig[lvl]) x = coop_inject_digest(x, p, inj.dig[lvl], gg, j);
                else if (icols) x = coop_inject(x, p, icols, inj.w[lvl], gg, j);
                if (g < half && j < 8) next[g * 8 + j] = x;
            }
        }
        // same-workgroup hand-off through global memory: make the stores visible to the whole group
        __threadfence_block();
        __syncthreads();
        cur = next;
        len = half;
        lvl++;
    }
}

}  // namespace

int32_t merkle_leaves(lurkhip_ctx* ctx, const P16Params* params_dev, const LeafCol* cols_dev, uint32_t total_w,
                      size_t n_rows, uint32_t* digests_out) {
    if (n_rows <= COOP_MAX_PARENTS && total_w <= 64) {
        // few rows: a handful of one-row-per-lane waves would each run ceil(w / 8) full-latency permutations
        const size_t blocks = (n_rows * 16 + MBLOCK - 1) / MBLOCK;
        hipLaunchKernelGGL(k_leaves_coop, dim3((unsigned)blocks), dim3(MBLOCK), 0, ctx->stream, params_dev, cols_dev, total_w, n_rows,
                           digests_out);
        LH_HIP(ctx, hipGetLastError());
        return LURKHIP_OK;
    }
    size_t blocks = (n_rows + MBLOCK - 1) / MBLOCK;
    LH_ARG(ctx, blocks <= 0x7fffffffu, "too many Merkle leaves for one launch");
    hipLaunchKernelGGL(k_leaves, dim3((unsigned)blocks), dim3(MBLOCK), 0, ctx->stream, params_dev, cols_dev, total_w,
                       n_rows, digests_out);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t merkle_level(lurkhip_ctx* ctx, const P16Params* params_dev, const uint32_t* children, size_t n_parents,
                     const LeafCol* inject_cols_dev, uint32_t inject_w, uint32_t* parents) {
    if (n_parents <= COOP_MAX_PARENTS) {
        const size_t blocks = (n_parents * 16 + MBLOCK - 1) / MBLOCK;
        hipLaunchKernelGGL(k_level_coop, dim3((unsigned)blocks), dim3(MBLOCK), 0, ctx->stream, params_dev, children, n_parents,
                           inject_cols_dev, inject_w, parents);
    } else {
        const size_t blocks = (n_parents + MBLOCK - 1) / MBLOCK;
        hipLaunchKernelGGL(k_level, dim3((unsigned)blocks), dim3(MBLOCK), 0, ctx->stream, params_dev, children, n_parents,
                           inject_cols_dev, inject_w, parents);
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t merkle_row_sponges(lurkhip_ctx* ctx, const P16Params* params_dev, SpongeGroups g) {
    if (g.n == 0) return LURKHIP_OK;
    LH_ARG(ctx, g.n <= SPONGE_MAX_GROUPS, "too many height groups for one sponge launch");
    // longest rows first (most permutations per lane): the grid's tail is then made of the cheapest lanes
    for (int a = 1; a < g.n; a++)
        for (int b = a; b > 0 && (g.total_w[b] + 7) / 8 > (g.total_w[b - 1] + 7) / 8; b--) {
            std::swap(g.cols[b], g.cols[b - 1]);
            std::swap(g.total_w[b], g.total_w[b - 1]);
            std::swap(g.n_rows[b], g.n_rows[b - 1]);
            std::swap(g.out[b], g.out[b - 1]);
        }
    // few rows of many permutations each go sixteen lanes to the row (coop_sponge_row); LURKHIP_SPONGE_COOP=0: none do
    static const bool coop_on = getenv("LURKHIP_SPONGE_COOP") == nullptr || atoi(getenv("LURKHIP_SPONGE_COOP")) != 0;
    size_t blocks = 0;
    for (int k = 0; k < g.n; k++) {
        g.first_block[k] = (uint32_t)blocks;
        g.coop[k] = coop_on && (g.total_w[k] + 7) / 8 >= SPONGE_COOP_MIN_PERMS && g.n_rows[k] <= SPONGE_COOP_MAX_ROWS;
        const size_t rows_per_block = g.coop[k] ? MBLOCK / 16 : MBLOCK;
        blocks += (g.n_rows[k] + rows_per_block - 1) / rows_per_block;
    }
    LH_ARG(ctx, blocks <= 0x7fffffffu, "too many rows for one sponge launch");
    g.first_block[g.n] = (uint32_t)blocks;
    hipLaunchKernelGGL(k_row_sponges, dim3((unsigned)blocks), dim3(MBLOCK), 0, ctx->stream, params_dev, g);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t merkle_level_digests(lurkhip_ctx* ctx, const P16Params* params_dev, const uint32_t* children, size_t n_parents,
                             const uint32_t* inject_digests, uint32_t* parents) {
    const size_t blocks = (n_parents + MBLOCK - 1) / MBLOCK;
    hipLaunchKernelGGL(k_level_digests, dim3((unsigned)blocks), dim3(MBLOCK), 0, ctx->stream, params_dev, children, n_parents, inject_digests,
                       parents);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t merkle_levels_coop(lurkhip_ctx* ctx, const P16Params* params_dev, uint32_t* children, size_t n_children, int levels,
                           const TopInject& inject) {
    LH_ARG(ctx, levels >= 1 && levels <= 5 && (n_children >> levels) >= 1 && (n_children & (n_children - 1)) == 0, "merkle_levels_coop: bad subtree shape");
    const size_t blocks = n_children >> levels;
    hipLaunchKernelGGL(k_levels_coop, dim3((unsigned)blocks), dim3(16u << (levels - 1)), 0, ctx->stream, params_dev, children, n_children, levels, inject);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t merkle_top(lurkhip_ctx* ctx, const P16Params* params_dev, uint32_t* level_base, size_t n, const TopInject& inject) {
    LH_ARG(ctx, n <= 2048 && (n & (n - 1)) == 0, "merkle_top needs a power of two <= 2048");
    if (n <= 1) return LURKHIP_OK;
    hipLaunchKernelGGL(k_top, dim3(1), dim3(1024), 0, ctx->stream, params_dev, level_base, n, inject);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

// ---- protocol profile (lurkhip.h): presets, the context's copy, and the width-16 permutation tables derived from it
namespace {

void preset_default(lurkhip_protocol_profile* p) {
    memset(p, 0, sizeof *p);
    p->struct_bytes = (uint32_t)sizeof *p;
    p->p16_rounds_p = 13;
    // the reference's own BabyBearConfig16 (/root/reference/src/poseidon/config.rs:190-199): sphinx's RC_16_30 are not in the tree
    for (int i = 0; i < 128; i++) p->p16_ext_rc[i] = LURK_P2_EXT_RC_16[i];
    for (int i = 0; i < 13; i++) p->p16_int_rc[i] = LURK_P2_INT_RC_16[i];
    for (int i = 0; i < 16; i++) p->p16_diag[i] = LURK_P2_DIAG_16[i];
    p->p16_internal_scale = 1;
    // DuplexChallenger<Val, Perm, 16, 8>: the RATE parameter bounds both the absorbed and the offered lanes (sponge_state[..RATE])
    // [UPSTREAM-RECALL]; offering the capacity lanes (16, preset "whole-state-squeeze") is opt-in until an upstream vector pins it
    p->challenger_squeeze = 8;
    p->challenger_pop_front = 0;
    p->observe_openings = 0;
    p->observe_chip_meta = 0;
    p->constraint_alpha_ascending = 0;
    p->fri_alpha_global = 0;
    p->fri_log_arity = 1;
    p->fri_log_blowup = 1;
    p->fri_num_queries = 100;
    p->fri_pow_bits = 16;
    p->serialize_montgomery = 0;
}

uint32_t pow_canonical(uint32_t a, uint64_t e) {
    uint64_t r = 1, b = a % bb::P;
    while (e) {
        if (e & 1) r = r * b % bb::P;
        b = b * b % bb::P;
        e >>= 1;
    }
    return (uint32_t)r;
}

int32_t validate_profile(lurkhip_ctx* ctx, const lurkhip_protocol_profile* p) {
    LH_ARG(ctx, p != nullptr, "null profile");
    LH_ARG(ctx, p->struct_bytes == sizeof *p, "profile struct_bytes %u, this library expects %zu", p->struct_bytes, sizeof *p);
    LH_ARG(ctx, p->p16_rounds_p >= 1 && p->p16_rounds_p <= (uint32_t)P16_MAX_RP, "p16_rounds_p must be in 1..%d", P16_MAX_RP);
    LH_ARG(ctx, p->p16_internal_scale % bb::P != 0, "p16_internal_scale must be non-zero");
    LH_ARG(ctx, p->challenger_squeeze == 8 || p->challenger_squeeze == 16, "challenger_squeeze must be 8 or 16");
    LH_ARG(ctx, p->challenger_pop_front <= 1 && p->observe_openings <= 1 && p->observe_chip_meta <= 1 && p->constraint_alpha_ascending <= 1 &&
                    p->fri_alpha_global <= 1 && p->serialize_montgomery <= 1,
           "profile flags must be 0 or 1");
    if (p->fri_log_arity != 1) return set_error(ctx, LURKHIP_ERR_UNSUPPORTED, "FRI folding arity 2^%u is not implemented (only 2)", p->fri_log_arity);
    LH_ARG(ctx, p->fri_log_blowup >= 1 && p->fri_log_blowup <= 4 && p->fri_num_queries >= 1 && p->fri_num_queries <= 1024 && p->fri_pow_bits <= 30,
           "FRI defaults out of range");
    return LURKHIP_OK;
}

P16Params tables_of(const lurkhip_protocol_profile& p) {
    P16Params h{};
    const uint32_t scale = p.p16_internal_scale % bb::P;
    for (int i = 0; i < 128; i++) h.ext_rc[i] = bb::to_monty(p.p16_ext_rc[i] % bb::P);
    for (uint32_t i = 0; i < p.p16_rounds_p; i++) h.int_rc[i] = bb::to_monty(p.p16_int_rc[i] % bb::P);
    for (int i = 0; i < 16; i++) h.diag[i] = bb::to_monty((uint32_t)((uint64_t)(p.p16_diag[i] % bb::P) * scale % bb::P));
    h.sum_mult = bb::to_monty(scale);
    h.rounds_p = (int32_t)p.p16_rounds_p;
    h.finish();
    return h;
}

}  // namespace

P16Params p16_tables_of(const lurkhip_protocol_profile& p) { return tables_of(p); }  // the host verifier's (verify.cpp)

const lurkhip_protocol_profile& profile_of(lurkhip_ctx* ctx) {
    if (!ctx->profile) {
        auto* p = new lurkhip_protocol_profile();
        preset_default(p);
        ctx->profile = p;
        ctx->cleanups.push_back([p]() { delete p; });
    }
    return *ctx->profile;
}

int32_t get_merkle_params(lurkhip_ctx* ctx, const P16Params** out_dev) {
    if (!ctx->merkle_params_dev) {
        const P16Params h = tables_of(profile_of(ctx));
        void* d = nullptr;
        LH_HIP(ctx, hipMalloc(&d, sizeof(P16Params)));
        LH_HIP(ctx, hipMemcpyAsync(d, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->merkle_params_dev = d;
        P16Params* hc = new P16Params(h);
        ctx->merkle_params_host = hc;
        ctx->cleanups.push_back([d, hc]() {
            (void)hipFree(d);
            delete hc;
        });
    }
    *out_dev = (const P16Params*)ctx->merkle_params_dev;
    return LURKHIP_OK;
}

}  // namespace lurkhip

extern "C" {

using namespace lurkhip;

int32_t lurkhip_protocol_profile_preset(const char* name, lurkhip_protocol_profile* out) {
    if (!name || !out) return LURKHIP_ERR_INVALID_ARG;
    preset_default(out);
    const std::string n = name;
    if (n == "default") return LURKHIP_OK;
    if (n == "hardened") {
        out->observe_openings = 1;
        out->observe_chip_meta = 1;
        out->challenger_squeeze = 8;
        return LURKHIP_OK;
    }
    if (n == "whole-state-squeeze") {  // round 2's default: all 16 lanes offered after a permutation
        out->challenger_squeeze = 16;
        return LURKHIP_OK;
    }
    if (n == "p3-monty-diffusion") {
        // DiffusionMatrixBabyBear as recalled: monty_reduce(sum + (x_i << shift_i)) on Montgomery words, i.e. 2^-32 (1 + diag)
        // with diag = [-2, 1, 2, 4, ..., 2^13, 2^15] [UPSTREAM-RECALL]
        static const int shifts[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
        out->p16_diag[0] = bb::P - 2;
        for (int i = 1; i < 16; i++) out->p16_diag[i] = 1u << shifts[i - 1];
        out->p16_internal_scale = pow_canonical(pow_canonical(2, 32), bb::P - 2);
        return LURKHIP_OK;
    }
    return LURKHIP_ERR_INVALID_ARG;
}

int32_t lurkhip_set_protocol_profile(lurkhip_ctx* ctx, const lurkhip_protocol_profile* profile) {
    LH_CHECK_CTX(ctx);
    LH_TRY(validate_profile(ctx, profile));
    (void)profile_of(ctx);
    *ctx->profile = *profile;
    if (ctx->merkle_params_dev) {  // the tables are already on the device: refresh them
        const P16Params h = tables_of(*profile);
        LH_HIP(ctx, hipSetDevice(ctx->device));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        LH_HIP(ctx, hipMemcpyAsync(ctx->merkle_params_dev, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *(P16Params*)ctx->merkle_params_host = h;
    }
    return LURKHIP_OK;
}

int32_t lurkhip_get_protocol_profile(lurkhip_ctx* ctx, lurkhip_protocol_profile* out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out != nullptr, "null argument");
    *out = profile_of(ctx);
    return LURKHIP_OK;
}

/* The profile's width-16 permutation itself, one state per lane: what the Merkle kernels, the sponge and the transcript run
 * (an upstream `poseidon2_16` vector is checked against this, tests/test_profile_gpu.py). */
int32_t lurkhip_perm16_dev(lurkhip_ctx* ctx, size_t n, const uint32_t* in, uint32_t* out, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, in && out, "null argument");
    LH_ARG(ctx, repr == LURKHIP_REPR_CANONICAL || repr == LURKHIP_REPR_MONTY, "bad repr");
    if (n == 0) return LURKHIP_OK;
    const P16Params* params = nullptr;
    LH_TRY(get_merkle_params(ctx, &params));
    hipLaunchKernelGGL(k_perm16_states, dim3((unsigned)((n + MBLOCK - 1) / MBLOCK)), dim3(MBLOCK), 0, ctx->stream, params, in, out, n,
                       repr == LURKHIP_REPR_CANONICAL);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t lurkhip_perm16(lurkhip_ctx* ctx, size_t n, const uint32_t* in, uint32_t* out, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, in && out, "null argument");
    if (n == 0) return LURKHIP_OK;
    void *din = nullptr, *dout = nullptr;
    LH_TRY(pool_alloc(ctx, n * 64, &din));
    LH_TRY(pool_alloc(ctx, n * 64, &dout));
    LH_HIP(ctx, hipMemcpyAsync(din, in, n * 64, hipMemcpyHostToDevice, ctx->stream));
    int32_t st = lurkhip_perm16_dev(ctx, n, (const uint32_t*)din, (uint32_t*)dout, repr);
    if (st == LURKHIP_OK) {
        hipError_t e = hipMemcpyAsync(out, dout, n * 64, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = set_error(ctx, LURKHIP_ERR_HIP, "lurkhip_perm16: %s", hipGetErrorString(e));
    }
    pool_release(ctx, din);
    pool_release(ctx, dout);
    return st;
}

/* older entry point: only the permutation tables (internal scale 1); kept for callers of ABI 1 */
int32_t lurkhip_set_merkle_poseidon2(lurkhip_ctx* ctx, int32_t rounds_p, const uint32_t* ext_rc, const uint32_t* int_rc,
                                     const uint32_t* diag) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, rounds_p > 0 && rounds_p <= P16_MAX_RP, "rounds_p must be in 1..%d", P16_MAX_RP);
    LH_ARG(ctx, ext_rc && int_rc && diag, "null parameter table");
    lurkhip_protocol_profile p = profile_of(ctx);
    p.p16_rounds_p = (uint32_t)rounds_p;
    for (int i = 0; i < 128; i++) p.p16_ext_rc[i] = ext_rc[i] % bb::P;
    for (int i = 0; i < 32; i++) p.p16_int_rc[i] = i < rounds_p ? int_rc[i] % bb::P : 0;
    for (int i = 0; i < 16; i++) p.p16_diag[i] = diag[i] % bb::P;
    p.p16_internal_scale = 1;
    return lurkhip_set_protocol_profile(ctx, &p);
}

}  // extern "C"
