// Kernels of the opening phase: evaluation of committed matrices at out-of-domain points, the FRI input
// ("reduced openings"), FRI folding, proof-of-work search and batched Merkle openings.
//
// Replaces (third-party, source absent from /root/reference; [UPSTREAM-RECALL] Plonky3 @ a0b92870, parity
// unpinned): p3_fri::TwoAdicFriPcs::open (interpolate_coset, compute_inverse_denominators, the
// "reduce rows" loop), p3_fri::prover::{commit_phase, fold_even_odd, answer_query},
// DuplexChallenger::grind, FieldMerkleTreeMmcs::open_batch -- reached from the reference through
// machine.prove::<LocalProver>, /root/reference/benches/fib.rs:124.
//
// All matrices are the committed LDEs: row-major, bit-reversed row order, Montgomery words; extension
// elements are 4 consecutive words.  Storage row s of a height-M matrix is the point x_s = 31 * w_M^bitrev(s).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "babybear.h"
#include "commit.h"
#include "ctx.h"
#include "fri.h"
#include "lazy_ef.h"
#include "p16_coop.h"
#include "poseidon2_dev.h"

namespace lurkhip {

namespace {

using bb::ef;

__device__ __forceinline__ ef ef_load(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return ef{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void ef_store(uint32_t* p, const ef& e) { *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]); }

__device__ __forceinline__ uint32_t brev_bits(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// ---------------------------------------------------------------- point-dependent row weights
// mode 0: u[s] = w^i / (z - g w^i)      (barycentric weights on the low coset, p3 interpolate_coset)
// mode 1: d[s] = 1 / (g w^i - z)        (p3 compute_inverse_denominators)
// with i = bitrev(s) over log_m bits, w = w_M.
// centred: store the coefficients as signed representatives in (-p/2, p/2] (operands of the lazy accumulators).
// tw[i] = w^i for i < M/2 is the NTT plan's twiddle table (w^(M/2) = -1), so no power ladder; a thread takes four
// consecutive rows and inverts their four extension elements with ONE base-field inversion: 1 / a = conj(a) / N(a) with the
// norm N(a) = a * a' * (a a')'' in the base field (two conjugations), and the four norms are inverted together
// (Montgomery's trick: 9 products + 1 Fermat ladder instead of 4 ladders).
constexpr int PW_BATCH = 4;
__device__ __forceinline__ void point_weights_body(int mode, int log_m, uint32_t g_m, const uint32_t* __restrict__ tw, const ef& z,
                                                   uint32_t* __restrict__ out, int centred) {
    const uint32_t m = 1u << log_m, half = m >> 1;
    const uint32_t s0 = (blockIdx.x * blockDim.x + threadIdx.x) * PW_BATCH;
    if (s0 >= m) return;
    ef c[PW_BATCH];       // conj-product a' * (a a')'' of each element: inverse = c / norm
    uint32_t nrm[PW_BATCH], wi[PW_BATCH];
#pragma unroll
    for (int k = 0; k < PW_BATCH; k++) {
        const uint32_t s = s0 + k < m ? s0 + k : m - 1;  // m < PW_BATCH: the tail lanes repeat the last row
        const uint32_t i = brev_bits(s, log_m);
        wi[k] = half == 0 ? bb::R1 : (i < half ? tw[i] : bb::neg(tw[i - half]));
        const uint32_t x = bb::mul(g_m, wi[k]);
        ef a;
        if (mode == 0) {
            a = z;
            a.c[0] = bb::sub(a.c[0], x);
        } else {
            a = ef{{bb::sub(x, z.c[0]), bb::neg(z.c[1]), bb::neg(z.c[2]), bb::neg(z.c[3])}};
        }
        const ef a1{{a.c[0], bb::neg(a.c[1]), a.c[2], bb::neg(a.c[3])}};
        const ef b = bb::ef_mul(a, a1);  // in span{1, x^2}
        const ef b1{{b.c[0], 0, bb::neg(b.c[2]), 0}};
        nrm[k] = bb::sub(bb::sqr(b.c[0]), bb::mul(bb::EXT_W_M, bb::sqr(b.c[2])));  // b * b1 = b0^2 - 11 b2^2
        c[k] = bb::ef_mul(a1, b1);
    }
    // batch inversion; a zero norm (z on the domain: cannot happen for a sampled challenge) inverts to zero like bb::inv
    uint32_t pre[PW_BATCH];
    uint32_t acc = bb::R1;
#pragma unroll
    for (int k = 0; k < PW_BATCH; k++) {
        pre[k] = acc;
        acc = bb::mul(acc, nrm[k] ? nrm[k] : bb::R1);
    }
    uint32_t inv = bb::inv(acc);
#pragma unroll
    for (int k = PW_BATCH - 1; k >= 0; k--) {
        const uint32_t ninv = nrm[k] ? bb::mul(inv, pre[k]) : 0u;
        inv = bb::mul(inv, nrm[k] ? nrm[k] : bb::R1);
        ef r = bb::ef_scale(c[k], mode == 0 ? bb::mul(ninv, wi[k]) : ninv);
        if (centred)
            for (int j = 0; j < 4; j++) r.c[j] = r.c[j] > bb::P / 2 ? r.c[j] - bb::P : r.c[j];
        if (s0 + k < m) ef_store(out + 4 * (size_t)(s0 + k), r);
    }
}
__global__ void k_point_weights(int mode, int log_m, uint32_t g_m, const uint32_t* __restrict__ tw, ef z, uint32_t* __restrict__ out,
                                int centred) {
    point_weights_body(mode, log_m, g_m, tw, z, out, centred);
}
// every table of an opening in one launch (blockIdx.y = table): launched one by one the forty tables of a fib-mix proof were
// forty dependent launches of 9-23 us on the main lane
struct PwBatchArgs {
    int mode[PW_BATCH_MAX], log_m[PW_BATCH_MAX];
    const uint32_t* tw[PW_BATCH_MAX];
    ef z[PW_BATCH_MAX];
    uint32_t* out[PW_BATCH_MAX];
};
__global__ void k_point_weights_batch(PwBatchArgs a, uint32_t g_m) {
    const int t = blockIdx.y;
    point_weights_body(a.mode[t], a.log_m[t], g_m, a.tw[t], a.z[t], a.out[t], a.mode[t] == 0 ? 1 : 0);
}

// ---------------------------------------------------------------- column-wise dot products with EF weights
// partial[blk][p][c] = sum over the block's rows of mat[s][c] * u_p[s].  A workgroup walks its rows R at a time, R =
// floor(256 / w): thread t holds element (row t / w, column t % w) of the current R x w slab, so the 256 lanes read
// consecutive words whatever the width (narrow matrices -- quotient chunks, memory tables -- fill the wave too);
// wider matrices (w > 256) take one row per step in column chunks of 256.
constexpr int DOT_ROWS = 1024;

__device__ __forceinline__ void column_dot_body(const uint32_t* __restrict__ mat, uint32_t w, uint32_t pitch, size_t n_rows,
                                                const uint32_t* __restrict__ u0, const uint32_t* __restrict__ u1,
                                                uint32_t* __restrict__ partial) {
    __shared__ uint32_t sh[2][256][4];
    const size_t row0 = (size_t)blockIdx.x * DOT_ROWS;
    const size_t row_end = row0 + DOT_ROWS < n_rows ? row0 + DOT_ROWS : n_rows;
    const int n_pts = u1 ? 2 : 1;
    const uint32_t R = w >= 256 ? 1u : 256u / w;
    const uint32_t rl = w >= 256 ? 0u : threadIdx.x / w;
    const uint32_t cl = w >= 256 ? threadIdx.x : threadIdx.x - rl * w;
    const bool active = rl < R;
    for (uint32_t cb = 0; cb < w; cb += 256) {
        const uint32_t c = cb + cl;
        // the weights are centred (k_point_weights): one multiply-add per coefficient into 64-bit lanes, folded every
        // second row
        LazyEf l0, l1;
        l0.zero();
        l1.zero();
        if (active && c < w) {
            // eight rows per trip: the matrix words go out as one batch of loads (one word in flight per lane leaves the
            // HBM latency uncovered), then the accumulation
            constexpr int UR = 8;
            size_t r = row0 + rl;
            for (; r + (size_t)(UR - 1) * R < row_end; r += (size_t)UR * R) {
                uint32_t m[UR];
                uint4 p0[UR], p1[UR];
#pragma unroll
                for (int k = 0; k < UR; k++) {
                    const size_t rr = r + (size_t)k * R;
                    m[k] = mat[rr * pitch + c];
                    p0[k] = *reinterpret_cast<const uint4*>(u0 + 4 * rr);
                    p1[k] = *reinterpret_cast<const uint4*>((u1 ? u1 : u0) + 4 * rr);
                }
#pragma unroll
                for (int k = 0; k < UR; k++) {
                    const int32_t w0[4] = {(int32_t)p0[k].x, (int32_t)p0[k].y, (int32_t)p0[k].z, (int32_t)p0[k].w};
                    l0.add_base_v(m[k], w0);
                    if (u1) {
                        const int32_t w1[4] = {(int32_t)p1[k].x, (int32_t)p1[k].y, (int32_t)p1[k].z, (int32_t)p1[k].w};
                        l1.add_base_v(m[k], w1);
                    }
                }
            }
            for (; r < row_end; r += R) {
                const uint32_t m = mat[r * pitch + c];
                const uint4 p0 = *reinterpret_cast<const uint4*>(u0 + 4 * r);
                const int32_t w0[4] = {(int32_t)p0.x, (int32_t)p0.y, (int32_t)p0.z, (int32_t)p0.w};
                l0.add_base_v(m, w0);
                if (u1) {
                    const uint4 p1 = *reinterpret_cast<const uint4*>(u1 + 4 * r);
                    const int32_t w1[4] = {(int32_t)p1.x, (int32_t)p1.y, (int32_t)p1.z, (int32_t)p1.w};
                    l1.add_base_v(m, w1);
                }
            }
        }
        const ef a0 = l0.value(), a1 = l1.value();
        for (int k = 0; k < 4; k++) {
            sh[0][threadIdx.x][k] = a0.c[k];
            sh[1][threadIdx.x][k] = a1.c[k];
        }
        __syncthreads();
        if (rl == 0 && c < w) {
            for (int p = 0; p < n_pts; p++) {
                ef t = bb::ef_zero();
                for (uint32_t q = 0; q < R; q++) {
                    const uint32_t* v = sh[p][q * (w >= 256 ? 0u : w) + cl];
                    t = bb::ef_add(t, ef{{v[0], v[1], v[2], v[3]}});
                }
                ef_store(partial + (((size_t)blockIdx.x * 2 + p) * w + c) * 4, t);
            }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_column_dot(const uint32_t* __restrict__ mat, uint32_t w, uint32_t pitch, size_t n_rows,
                                                     const uint32_t* __restrict__ u0, const uint32_t* __restrict__ u1,
                                                     uint32_t* __restrict__ partial) {
    column_dot_body(mat, w, pitch, n_rows, u0, u1, partial);
}
// the narrow matrices of an opening in one launch (blockIdx.y = matrix, blockIdx.x = its block of DOT_ROWS rows): the memory
// tables, quotient chunks and permutation traces of the small chips were sixty dependent launches of 9-18 us
struct NarrowDotArgs {
    const uint32_t* mat[NARROW_DOT_MAX];
    const uint32_t *u0[NARROW_DOT_MAX], *u1[NARROW_DOT_MAX];
    uint32_t* partial[NARROW_DOT_MAX];
    uint32_t w[NARROW_DOT_MAX], pitch[NARROW_DOT_MAX], n_rows[NARROW_DOT_MAX];
};
__global__ __launch_bounds__(256) void k_column_dot_batch(NarrowDotArgs a) {
    const int t = blockIdx.y;
    const size_t n_rows = a.n_rows[t];
    if ((size_t)blockIdx.x * DOT_ROWS >= n_rows) return;
    column_dot_body(a.mat[t], a.w[t], a.pitch[t], n_rows, a.u0[t], a.u1[t], a.partial[t]);
}

// Matrices of 24 columns and more: a wave takes 64 consecutive columns of one row at a time, so the row's weights are
// wave-uniform -- scalar loads, SGPR operands of the multiply-adds -- and a lane's only vector load is its matrix word
// (k_column_dot reads two 16-byte weights per word through the vector path: 36 bytes requested per 4 bytes of matrix).
// Workgroup = (block of `rows_per_block` rows, chunk of 64 columns); its four waves split the rows.
__global__ __launch_bounds__(256) void k_column_dot_wave(const uint32_t* __restrict__ mat, uint32_t w, uint32_t pitch, size_t n_rows, uint32_t rows_per_block,
                                                          uint32_t n_chunks, const uint32_t* __restrict__ u0, const uint32_t* __restrict__ u1,
                                                          uint32_t* __restrict__ partial) {
    __shared__ uint32_t sh[2][4][64][4];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t per_wave = rows_per_block / 4u;
    // (row block, column chunk) of this workgroup: workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and the
    // chunks of a row block share the cache lines their rows straddle -- every XCD takes a contiguous run of (block, chunk)
    // pairs, chunk fastest, so those lines are fetched from HBM once
    uint32_t g = blockIdx.x;
    if ((gridDim.x & 7u) == 0) g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t blk = g / n_chunks, chunk = g - blk * n_chunks;
    const size_t blk_row0 = (size_t)blk * rows_per_block;
    size_t r = blk_row0 + (size_t)wave * per_wave;
    const size_t wave_end = r + per_wave < n_rows ? r + per_wave : n_rows;
    const uint32_t c = chunk * 64u + lane, cc = c < w ? c : w - 1u;
    const uint32_t* __restrict__ col = mat + cc;
    LazyEf l0, l1;
    l0.zero();
    l1.zero();
    auto weights = [&](const uint32_t* __restrict__ u, size_t row, int32_t (&o)[8]) {
        const uint4 q = *reinterpret_cast<const uint4*>(u + 4 * row);  // uniform address: a scalar load
        o[0] = (int32_t)q.x; o[1] = (int32_t)q.y; o[2] = (int32_t)q.z; o[3] = (int32_t)q.w;
        o[4] = o[5] = o[6] = o[7] = 0;
    };
    constexpr int UR = 8;  // rows per trip: their words go out as one batch of loads (16 rows and a software-pipelined variant measured the same)
    for (; r + UR <= wave_end; r += UR) {
        uint32_t m[UR];
#pragma unroll
        for (int k = 0; k < UR; k++) m[k] = col[(r + k) * pitch];
#pragma unroll
        for (int k = 0; k < UR; k++) {
            int32_t w0[8];
            weights(u0, r + k, w0);
            l0.add_base(m[k], w0);
            if (u1) {
                int32_t w1[8];
                weights(u1, r + k, w1);
                l1.add_base(m[k], w1);
            }
        }
    }
    for (; r < wave_end; r++) {
        const uint32_t m = col[r * pitch];
        int32_t w0[8];
        weights(u0, r, w0);
        l0.add_base(m, w0);
        if (u1) {
            int32_t w1[8];
            weights(u1, r, w1);
            l1.add_base(m, w1);
        }
    }
    const ef a0 = l0.value(), a1 = l1.value();
    for (int k = 0; k < 4; k++) {
        sh[0][wave][lane][k] = a0.c[k];
        sh[1][wave][lane][k] = a1.c[k];
    }
    __syncthreads();
    // threads 0..63 finish point 0, 64..127 point 1
    const uint32_t p = threadIdx.x >> 6;
    if (p < (u1 ? 2u : 1u) && c < w) {
        ef t = bb::ef_zero();
        for (int q = 0; q < 4; q++) t = bb::ef_add(t, ef{{sh[p][q][lane][0], sh[p][q][lane][1], sh[p][q][lane][2], sh[p][q][lane][3]}});
        ef_store(partial + (((size_t)blk * 2 + p) * w + c) * 4, t);
    }
}

// Matrices of 24 .. 256 columns (round 4): the slab walk of k_column_dot -- thread = (row of the slab, column), so the 256 lanes
// read CONSECUTIVE words of the row-major matrix whatever its width and alignment -- with the block's weights staged in LDS once
// (k_column_dot_wave read one word per lane at a stride of w words: 64-column pieces of 312-byte rows, the last piece mostly
// idle lanes, and two scalar loads per row on the critical path -- 1.7 TB/s, 76 % of the wave cycles waiting; k_column_dot's own
// per-lane weight loads were 36 bytes requested per 4 bytes of matrix).  V = 2: rows wider than 128 columns are cut in two halves
// of T = ceil(w / 2) columns and a thread takes column ct of both, so that R = 256 / T >= 2 rows fit a step.
template <int V>
__global__ __launch_bounds__(256) void k_column_dot_slab(const uint32_t* __restrict__ mat, uint32_t w, uint32_t pitch, size_t n_rows, uint32_t rows_per_block,
                                                          const uint32_t* __restrict__ u0, const uint32_t* __restrict__ u1,
                                                          uint32_t* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dot_lds[];
    uint4* wts = reinterpret_cast<uint4*>(dot_lds);                     // [row][point]
    uint32_t* red = dot_lds + (size_t)rows_per_block * 8;              // [point][V][256][4]
    const size_t row0 = (size_t)blockIdx.x * rows_per_block;
    const uint32_t rows = (uint32_t)(row0 + rows_per_block <= n_rows ? rows_per_block : n_rows - row0);
    for (uint32_t i = threadIdx.x; i < rows; i += 256) {
        wts[2 * i] = *reinterpret_cast<const uint4*>(u0 + 4 * (row0 + i));
        wts[2 * i + 1] = u1 ? *reinterpret_cast<const uint4*>(u1 + 4 * (row0 + i)) : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const uint32_t T = (w + V - 1) / V, R = 256u / T;
    const uint32_t rl = threadIdx.x / T, ct = threadIdx.x - rl * T;
    const bool active = rl < R;
    uint32_t col[V];
    bool real[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
        const uint32_t c = ct + (uint32_t)v * T;
        real[v] = c < w;
        col[v] = real[v] ? c : w - 1u;  // an idle half-column shadows the last one: no branch around the loads
    }
    LazyEf acc[V][2];
#pragma unroll
    for (int v = 0; v < V; v++) acc[v][0].zero(), acc[v][1].zero();
    const uint32_t* __restrict__ base = mat + row0 * pitch;
    auto step = [&](const uint32_t (&m)[V], uint32_t r) {
        const uint4 p0 = wts[2 * r], p1 = wts[2 * r + 1];
        const int32_t w0[4] = {(int32_t)p0.x, (int32_t)p0.y, (int32_t)p0.z, (int32_t)p0.w};
        const int32_t w1[4] = {(int32_t)p1.x, (int32_t)p1.y, (int32_t)p1.z, (int32_t)p1.w};
#pragma unroll
        for (int v = 0; v < V; v++) {
            acc[v][0].add_base_v(m[v], w0);
            if (u1) acc[v][1].add_base_v(m[v], w1);
        }
    };
    if (active) {
        constexpr int UR = 8;  // rows in flight per lane
        uint32_t r = rl;
        for (; r + (UR - 1) * R < rows; r += UR * R) {
            uint32_t m[UR][V];
#pragma unroll
            for (int k = 0; k < UR; k++)
#pragma unroll
                for (int v = 0; v < V; v++) m[k][v] = base[(size_t)(r + k * R) * pitch + col[v]];
#pragma unroll
            for (int k = 0; k < UR; k++) step(m[k], r + k * R);
        }
        for (; r < rows; r += R) {
            uint32_t m[V];
#pragma unroll
            for (int v = 0; v < V; v++) m[v] = base[(size_t)r * pitch + col[v]];
            step(m, r);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int v = 0; v < V; v++) {
            const ef a = acc[v][p].value();
            for (int k = 0; k < 4; k++) red[((p * V + v) * 256 + threadIdx.x) * 4 + k] = a.c[k];
        }
    __syncthreads();
    if (rl == 0) {
        const int n_pts = u1 ? 2 : 1;
        for (int p = 0; p < n_pts; p++)
#pragma unroll
            for (int v = 0; v < V; v++) {
                if (!real[v]) continue;
                ef t = bb::ef_zero();
                for (uint32_t q = 0; q < R; q++) t = bb::ef_add(t, ef_load(red + ((p * V + v) * 256 + q * T + ct) * 4));
                ef_store(partial + (((size_t)blockIdx.x * 2 + p) * w + col[v]) * 4, t);
            }
    }
}

// out[p][c] = sum over blocks of partial[blk][p][c]; one workgroup per (p, c),
// every matrix of a proof in one launch: block -> (matrix, point, column) through the prefix sums in the arguments
__global__ __launch_bounds__(256) void k_dot_finish_all(DotFinishArgs a) {
    __shared__ uint32_t sh[256][4];
    uint32_t m = 0;
    while (m + 1 < a.n && blockIdx.x >= a.col_start[m + 1]) m++;
    const uint32_t local = blockIdx.x - a.col_start[m], w = a.w[m], n_blocks = a.n_blocks[m];
    const uint32_t p = local >= w ? 1u : 0u, c = local - p * w;
    const uint32_t* __restrict__ partial = a.partial[m];
    ef t = bb::ef_zero();
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) t = bb::ef_add(t, ef_load(partial + (((size_t)b * 2 + p) * w + c) * 4));
    for (int k = 0; k < 4; k++) sh[threadIdx.x][k] = t.c[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 4; k++) sh[threadIdx.x][k] = bb::add(sh[threadIdx.x][k], sh[threadIdx.x + off][k]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; k++) a.out[a.out_off[m] + ((size_t)p * w + c) * 4 + k] = sh[0][k];
}

// ---------------------------------------------------------------- reduced openings
// ro[s] += apow0 * (rr - ys0) * d0[s] + apow1 * (rr - ys1) * d1[s],  rr = sum_c alpha^c mat[s][c]
struct ReduceArgs {
    const uint32_t* mat;
    uint32_t w;
    uint32_t m_rows;
    const uint32_t* alpha_pows;  // alpha^c, c < w
    const uint32_t* d0;
    const uint32_t* d1;  // nullable
    ef ys0, ys1, apow0, apow1;
    uint32_t* ro;
    int staged;
};

// 64 rows per workgroup; the rows are staged in LDS with coalesced loads (odd row stride) when they fit
__global__ __launch_bounds__(64) void k_reduce_openings(ReduceArgs a) {
    extern __shared__ uint32_t tile[];
    const uint32_t s0 = blockIdx.x * 64u, s = s0 + threadIdx.x;
    const uint32_t* row = a.mat + (size_t)s * a.w;
    if (a.staged) {
        const uint32_t wp = a.w | 1u;
        const uint32_t rows = a.m_rows - s0 < 64u ? a.m_rows - s0 : 64u;
        const uint32_t total = rows * a.w;  // the 64 rows are one contiguous run of words
        const uint32_t* src = a.mat + (size_t)s0 * a.w;
        if (a.w < 64) {
            // narrow rows: flat copy, every lane busy (one division per element is cheaper than idle lanes)
            for (uint32_t e = threadIdx.x; e < total; e += 64) {
                const uint32_t r = e / a.w;
                tile[r * wp + (e - r * a.w)] = src[e];
            }
        } else {
            for (uint32_t r = 0; r < rows; r++)
                for (uint32_t c = threadIdx.x; c < a.w; c += 64) tile[r * wp + c] = src[r * a.w + c];
        }
        __syncthreads();
        row = tile + threadIdx.x * wp;
    }
    if (s >= a.m_rows) return;
    ef rr = bb::ef_zero();
    {
        // the alpha powers are wave-uniform (scalar loads): fetch four at a time so their latency is paid once per
        // four columns
        const uint32_t* __restrict__ ap = a.alpha_pows;
        uint32_t c = 0;
        for (; c + 4 <= a.w; c += 4) {
            const ef p0 = ef_load(ap + 4 * c), p1 = ef_load(ap + 4 * c + 4), p2 = ef_load(ap + 4 * c + 8), p3 = ef_load(ap + 4 * c + 12);
            const uint32_t v0 = row[c], v1 = row[c + 1], v2 = row[c + 2], v3 = row[c + 3];
            rr = bb::ef_add(rr, bb::ef_add(bb::ef_add(bb::ef_scale(p0, v0), bb::ef_scale(p1, v1)),
                                           bb::ef_add(bb::ef_scale(p2, v2), bb::ef_scale(p3, v3))));
        }
        for (; c < a.w; c++) rr = bb::ef_add(rr, bb::ef_scale(ef_load(ap + 4 * c), row[c]));
    }
    ef acc = ef_load(a.ro + 4 * (size_t)s);
    acc = bb::ef_add(acc, bb::ef_mul(a.apow0, bb::ef_mul(bb::ef_sub(rr, a.ys0), ef_load(a.d0 + 4 * (size_t)s))));
    if (a.d1) acc = bb::ef_add(acc, bb::ef_mul(a.apow1, bb::ef_mul(bb::ef_sub(rr, a.ys1), ef_load(a.d1 + 4 * (size_t)s))));
    ef_store(a.ro + 4 * (size_t)s, acc);
}

// Streaming version for w <= 128: one wave per workgroup, persistent over 64-row tiles.  A tile is 64 * w contiguous
// words: lane l loads words l, l + 64, ... (coalesced) into registers while the previous tile is being reduced, drops them
// into LDS in the same flat order (position e + (e >> 5): one pad word per 32 keeps the row-wise reads of every width off a
// single bank) and then takes row l: sum_c alpha^c * row[c] in lazily reduced 64-bit lanes (lazy_ef.h; alpha powers
// centred, 8 words each).  NV = words per lane, >= w.
// BIG: matrices of 4 GiB and more index their words with 64 bits; the others with one 32-bit add and one min per load.
template <int NV, bool BIG>
__global__ __launch_bounds__(64) void k_reduce_openings_stream(ReduceArgs a) {
    extern __shared__ uint32_t tile[];
    using idx_t = typename std::conditional<BIG, size_t, uint32_t>::type;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_tiles = (a.m_rows + 63u) / 64u;
    const idx_t total = (idx_t)a.m_rows * (idx_t)a.w;  // words in the matrix
    uint32_t v[NV];
    auto fetch = [&](uint32_t t) {
        const idx_t base = (idx_t)t * 64u * a.w + lane;
        const idx_t lim = total - 1;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            // words past the tile (k >= w) or past the matrix re-read a valid word: no branch around a load
            const uint32_t kk = (uint32_t)k < a.w ? (uint32_t)k : a.w - 1u;
            idx_t e = base + (idx_t)(kk * 64u);
            e = e < lim ? e : lim;
            v[k] = a.mat[e];
        }
    };
    uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    fetch(t);
    // The alpha powers (first four words of each centred entry) sit in LDS behind the tile: read from global memory inside the
    // column loop they were vector loads, and waiting for one also waited for the next tile's rows in flight.
    uint32_t* __restrict__ apw = tile + (NV * 66 + 64 + 8);
    for (uint32_t e = lane; e < 4u * a.w; e += 64u) apw[e] = a.alpha_pows[8u * (e >> 2) + (e & 3u)];
    const uint32_t wbase = lane + (lane >> 5);  // position of flat word e = k * 64 + lane: e + (e >> 5) = k * 66 + wbase
    for (; t < n_tiles; t += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NV; k++) tile[wbase + (uint32_t)k * 66u] = v[k];
        __syncthreads();
        const uint32_t t_next = t + gridDim.x < n_tiles ? t + gridDim.x : t;
        fetch(t_next);
        const uint32_t s = t * 64u + lane;
        LazyEf rr;
        rr.zero();
        uint32_t e = lane * a.w;
        for (uint32_t c = 0; c < a.w; c++, e++) {
            const uint4 q = *reinterpret_cast<const uint4*>(apw + 4 * c);
            const int32_t pw[4] = {(int32_t)q.x, (int32_t)q.y, (int32_t)q.z, (int32_t)q.w};
            rr.add_base_v(tile[e + (e >> 5)], pw);
        }
        if (s < a.m_rows) {
            const ef r = rr.value();
            ef acc = ef_load(a.ro + 4 * (size_t)s);
            acc = bb::ef_add(acc, bb::ef_mul(a.apow0, bb::ef_mul(bb::ef_sub(r, a.ys0), ef_load(a.d0 + 4 * (size_t)s))));
            if (a.d1) acc = bb::ef_add(acc, bb::ef_mul(a.apow1, bb::ef_mul(bb::ef_sub(r, a.ys1), ef_load(a.d1 + 4 * (size_t)s))));
            ef_store(a.ro + 4 * (size_t)s, acc);
        }
        __syncthreads();  // the tile is free for the next one
    }
}

// Quad version of the streaming kernel (17 <= w <= 128): a tile is 16 rows, four lanes share a row (columns c = part, part + 4,
// ...) and meet through two quad-permute additions.  The 64-rows-per-wave kernel above needs 66 * NV words of LDS and NV >= w
// staging registers per lane: seven single-wave workgroups per CU at w = 78, each walking its row through LDS with nothing
// else resident to hide the latency (12 T lane-instr/s, 39 % of the cycles waiting: 2.1 TB/s).  Here a lane stages w / 4 words,
// a tile is a quarter of the LDS, and four times as many waves are resident.
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
template <int NV, bool BIG>
__global__ __launch_bounds__(64) void k_reduce_openings_quad(ReduceArgs a) {
    extern __shared__ uint32_t tile[];
    using idx_t = typename std::conditional<BIG, size_t, uint32_t>::type;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_tiles = (a.m_rows + 15u) / 16u;
    const idx_t total = (idx_t)a.m_rows * (idx_t)a.w;
    uint32_t v[NV];
    auto fetch = [&](uint32_t t) {
        const idx_t base = (idx_t)t * 16u * a.w + lane;
        const idx_t lim = total - 1;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            idx_t e = base + (idx_t)(k * 64);  // words past the tile belong to the next one (or are clamped): read, never used
            e = e < lim ? e : lim;
            v[k] = a.mat[e];
        }
    };
    uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    fetch(t);
    uint32_t* __restrict__ apw = tile + (NV * 66 + 8);
    for (uint32_t e = lane; e < 4u * a.w; e += 64u) apw[e] = a.alpha_pows[8u * (e >> 2) + (e & 3u)];
    const uint32_t wbase = lane + (lane >> 5);
    const uint32_t r = lane >> 2, part = lane & 3u;
    for (; t < n_tiles; t += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NV; k++) tile[wbase + (uint32_t)k * 66u] = v[k];
        __syncthreads();
        const uint32_t t_next = t + gridDim.x < n_tiles ? t + gridDim.x : t;
        fetch(t_next);
        const uint32_t s = t * 16u + r;
        LazyEf rr;
        rr.zero();
        uint32_t e = r * a.w + part;
        for (uint32_t c = part; c < a.w; c += 4, e += 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(apw + 4 * c);
            const int32_t pw[4] = {(int32_t)q.x, (int32_t)q.y, (int32_t)q.z, (int32_t)q.w};
            rr.add_base_v(tile[e + (e >> 5)], pw);
        }
        ef x = rr.value();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            x.c[k] = bb::add(x.c[k], dpp<DPP_QUAD_XOR1>(x.c[k]));
            x.c[k] = bb::add(x.c[k], dpp<DPP_QUAD_XOR2>(x.c[k]));
        }
        // the tail, apow_p * (x - ys_p) * d_p[s]: lane 0 of the quad takes the first point, lane 1 the second, they meet in lane 0
        const bool second = (part & 1u) != 0;
        const size_t sr = s < a.m_rows ? s : a.m_rows - 1;
        const uint32_t* __restrict__ dp = second && a.d1 ? a.d1 : a.d0;
        ef ys, ap;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ys.c[k] = second ? a.ys1.c[k] : a.ys0.c[k];
            ap.c[k] = second ? a.apow1.c[k] : a.apow0.c[k];
        }
        ef term = bb::ef_mul(ap, bb::ef_mul(bb::ef_sub(x, ys), ef_load(dp + 4 * sr)));
        if (second && !a.d1) term = bb::ef_zero();
#pragma unroll
        for (int k = 0; k < 4; k++) term.c[k] = bb::add(term.c[k], dpp<DPP_QUAD_XOR1>(term.c[k]));
        if (part == 0 && s < a.m_rows) ef_store(a.ro + 4 * (size_t)s, bb::ef_add(ef_load(a.ro + 4 * (size_t)s), term));
        __syncthreads();  // the tile is free for the next one
    }
}

// Wide matrices (w > 128: eval_builtin_expr's 148 columns, the hash chips' 493 / 655 / 815): the row is cut into column
// slices of at most 128 words and a workgroup walks (tile, slice) items the way the streaming kernel walks tiles -- the
// next item's words in flight while the current one is reduced --, the words of slice sl of the 64-row tile being
// word (e / sw, e % sw) of a strided block.  With rr_sl = sum_j alpha^j mat[s][c0 + j] a row accumulates
//   G_p += apow_p * alpha^c0 * (rr_sl - ys_p[sl])
// over the slices and touches d0 / d1 / ro once: ro[s] += G_0 d0[s] + G_1 d1[s].
template <int NV, bool BIG>
__global__ __launch_bounds__(64) void k_reduce_openings_wide(WideArgs a) {
    extern __shared__ uint32_t tile[];
    using idx_t = typename std::conditional<BIG, size_t, uint32_t>::type;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_tiles = (a.m_rows + 63u) / 64u;
    const uint32_t n_items = n_tiles * a.n_slices;
    const idx_t last_row = (idx_t)a.m_rows - 1;
    uint32_t v[NV];
    auto fetch = [&](uint32_t item) {
        const uint32_t t = item / a.n_slices, sl = item - t * a.n_slices;
        const uint32_t sw = a.sw[sl], c0 = a.c0[sl], magic = a.magic[sl];
#pragma unroll
        for (int k = 0; k < NV; k++) {
            // words past the slice block re-read a valid word: no branch around a load
            uint32_t e = (uint32_t)k * 64u + lane;
            e = e < 64u * sw ? e : 64u * sw - 1u;
            const uint32_t r = __umulhi(e, magic), j = e - r * sw;
            idx_t row = (idx_t)t * 64u + r;
            row = row < last_row ? row : last_row;
            v[k] = a.mat[row * (idx_t)a.w + (idx_t)(c0 + j)];
        }
    };
    uint32_t item = blockIdx.x * a.n_slices;  // a workgroup takes whole tiles
    if (item >= n_items) return;
    fetch(item);
    uint32_t* __restrict__ apw = tile + (NV * 66 + 64 + 8);
    for (uint32_t e = lane; e < 4u * NV; e += 64u) apw[e] = a.alpha_pows[8u * (e >> 2) + (e & 3u)];
    const uint32_t wbase = lane + (lane >> 5);
    ef g0 = bb::ef_zero(), g1 = bb::ef_zero();
    while (true) {
        const uint32_t t = item / a.n_slices, sl = item - t * a.n_slices;
#pragma unroll
        for (int k = 0; k < NV; k++) tile[wbase + (uint32_t)k * 66u] = v[k];
        __syncthreads();
        uint32_t next = item + 1;
        if (sl + 1 == a.n_slices) next = (t + gridDim.x) * a.n_slices;
        const bool more = next < n_items;
        fetch(more ? next : item);
        const uint32_t sw = a.sw[sl];
        LazyEf rr;
        rr.zero();
        uint32_t e = lane * sw;
        for (uint32_t c = 0; c < sw; c++, e++) {
            const uint4 q = *reinterpret_cast<const uint4*>(apw + 4 * c);
            const int32_t pw[4] = {(int32_t)q.x, (int32_t)q.y, (int32_t)q.z, (int32_t)q.w};
            rr.add_base_v(tile[e + (e >> 5)], pw);
        }
        const ef r = rr.value();
        g0 = bb::ef_add(g0, bb::ef_mul(a.apow0[sl], bb::ef_sub(r, a.ys0[sl])));
        if (a.d1) g1 = bb::ef_add(g1, bb::ef_mul(a.apow1[sl], bb::ef_sub(r, a.ys1[sl])));
        if (sl + 1 == a.n_slices) {
            const uint32_t s = t * 64u + lane;
            if (s < a.m_rows) {
                ef acc = ef_load(a.ro + 4 * (size_t)s);
                acc = bb::ef_add(acc, bb::ef_mul(g0, ef_load(a.d0 + 4 * (size_t)s)));
                if (a.d1) acc = bb::ef_add(acc, bb::ef_mul(g1, ef_load(a.d1 + 4 * (size_t)s)));
                ef_store(a.ro + 4 * (size_t)s, acc);
            }
            g0 = bb::ef_zero();
            g1 = bb::ef_zero();
        }
        __syncthreads();  // the tile is free for the next item
        if (!more) break;
        item = next;
    }
}

// Narrow matrices (w <= NARROW_MAX_W: memory tables, quotient chunks, the callee chip's permutation trace) of one height in one
// launch.  Per row their own words are a few bytes next to the 64 bytes of d0 / d1 / ro traffic and the four extension products of
// the tail, so one launch per matrix is bound by the tail: here a lane owns a row, walks the matrices
//   G_p += apow_p[m] * (sum_c alpha^c mat_m[s][c] - ys_p[m])
// and touches d0, d1, ro once: ro[s] += G_0 d0[s] + G_1 d1[s].
__global__ __launch_bounds__(256) void k_reduce_openings_narrow(NarrowArgs a) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= a.m_rows) return;
    const uint32_t* __restrict__ ap = a.alpha_pows;
    ef g0 = bb::ef_zero(), g1 = bb::ef_zero();
    for (uint32_t m = 0; m < a.n_mats; m++) {
        const NarrowMat& nm = a.m[m];
        const uint32_t w = nm.w;
        const uint32_t* __restrict__ row = nm.mat + (size_t)s * (nm.pitch ? nm.pitch : w);
        LazyEf rr;
        rr.zero();
        for (uint32_t c0 = 0; c0 < w; c0 += 4) {
            // four words per trip go out together; words past the row re-read its last one and are not accumulated
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = row[c0 + k < w ? c0 + k : w - 1];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (c0 + k < w) {
                    int32_t pw[8];
                    load_w8(pw, ap + 8 * (c0 + k));
                    rr.add_base(v[k], pw);
                }
        }
        const ef r = rr.value();
        g0 = bb::ef_add(g0, bb::ef_mul(nm.apow0, bb::ef_sub(r, nm.ys0)));
        if (nm.two) g1 = bb::ef_add(g1, bb::ef_mul(nm.apow1, bb::ef_sub(r, nm.ys1)));
    }
    ef acc = ef_load(a.ro + 4 * (size_t)s);
    acc = bb::ef_add(acc, bb::ef_mul(g0, ef_load(a.d0 + 4 * (size_t)s)));
    if (a.d1) acc = bb::ef_add(acc, bb::ef_mul(g1, ef_load(a.d1 + 4 * (size_t)s)));
    ef_store(a.ro + 4 * (size_t)s, acc);
}

// ---------------------------------------------------------------- FRI fold (p3 fold_even_odd)
// out[j] = (1/2 + beta/2 * ginv^bitrev(j)) e[2j] + (1/2 - beta/2 * ginv^bitrev(j)) e[2j+1]  (+ add[j]),
// ginv = (generator of the size-len subgroup)^-1; len = 2^log_len
// (j_base, n_out: the launch folds the n_out pairs from pair j_base on, its buffers starting there -- a rank's block of a layer when
// several ranks prove one shard together; 0 and half the length otherwise)
__global__ __launch_bounds__(256) void k_fri_fold(const uint32_t* __restrict__ cur, int log_len, const uint32_t* __restrict__ beta_dev,
                                                   uint32_t ginv_m, uint32_t half_m, const uint32_t* __restrict__ add,
                                                   uint32_t* __restrict__ out, uint32_t j_base, uint32_t n_out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    // the layer's challenge was sampled on the device (k_fri_challenge): beta / 2
    const ef half_beta = bb::ef_scale(ef{{beta_dev[0], beta_dev[1], beta_dev[2], beta_dev[3]}}, half_m);
    const ef power = bb::ef_scale(half_beta, bb::pow(ginv_m, brev_bits(j_base + j, log_len - 1)));
    const ef e0 = ef_load(cur + 8 * (size_t)j), e1 = ef_load(cur + 8 * (size_t)j + 4);
    ef r = bb::ef_add(bb::ef_mul(bb::ef_add_base(power, half_m), e0),
                      bb::ef_mul(bb::ef_add_base(bb::ef_sub(bb::ef_zero(), power), half_m), e1));
    if (add) r = bb::ef_add(r, ef_load(add + 4 * (size_t)j));
    ef_store(out + 4 * (size_t)j, r);
}

// ---------------------------------------------------------------- device side of the transcript (FRI commit phase)
// p3's DuplexChallenger (challenger.h) on the device for the one stretch of the proof where the transcript sits on the
// critical path of a latency chain: per FRI layer "observe the layer root, sample beta".  One wave; the width-16 state is held
// by 16 lanes and permuted cooperatively (p16_coop.h).  The host uploads its transcript state before the first layer and
// reads it back after the last one, so the 21 root read-backs and host permutations of a proof leave the chain.
__global__ __launch_bounds__(64) void k_fri_challenge(const P16Params* __restrict__ p, DevChallenger* __restrict__ ch,
                                                      const uint32_t* __restrict__ root, uint32_t* __restrict__ beta_out,
                                                      uint32_t* __restrict__ root_copy) {
    __shared__ DevChallenger c;
    const int lane = threadIdx.x, j = lane & 15;
    if (lane == 0) c = *ch;
    if (root_copy && lane < 8) root_copy[lane] = root[lane];  // the proof's copy of the root, read back once for all layers
    __syncthreads();
    auto duplex = [&]() {
        uint32_t x = (uint32_t)j < c.n_in ? c.input[j & 7] : c.state[j];
        x = coop_perm16(x, p, j);
        __syncthreads();
        if (lane < 16) {
            c.state[lane] = x;
            c.output[lane] = x;
        }
        if (lane == 0) {
            c.n_in = 0;
            c.n_out = c.squeeze;
            c.out_head = 0;
        }
        __syncthreads();
    };
    for (int i = 0; i < 8; i++) {  // observe the digest, one lane value at a time
        if (lane == 0) {
            c.n_out = 0;
            c.input[c.n_in] = root[i];
            c.n_in += 1;
        }
        __syncthreads();
        if (c.n_in == 8) duplex();
    }
    for (int i = 0; i < 4; i++) {  // sample an extension element: pops from the end of the output buffer (or its front)
        if (c.n_in != 0 || c.n_out == 0) duplex();
        if (lane == 0) {
            c.n_out -= 1;
            if (c.pop_front) {
                beta_out[i] = c.output[c.out_head];
                c.out_head += 1;
            } else {
                beta_out[i] = c.output[c.out_head + c.n_out];
            }
        }
        __syncthreads();
    }
    if (lane == 0) *ch = c;
}

// ---------------------------------------------------------------- proof of work (DuplexChallenger::grind)
// witness w is accepted when, after observing it, the next sampled element has `bits` low zero bits: one
// permutation of the state with the pending inputs and w written over its first lanes; the sample is lane squeeze - 1 (7 or 15), or lane 0 when samples pop from the front.
struct PowState {
    uint32_t s[16];  // the transcript state, pending inputs already written over lanes [0, n_pending); a launch argument
};
__global__ __launch_bounds__(256) void k_pow_grind(const P16Params* __restrict__ p, PowState state_in,
                                                    int n_pending, int sample_lane, uint32_t base, uint32_t mask, uint32_t* __restrict__ best) {
    const uint32_t wcan = base + blockIdx.x * blockDim.x + threadIdx.x;
    if (wcan >= bb::P) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = state_in.s[i];
    const uint32_t wm = bb::to_monty(wcan);
    // the witness lands in lane n_pending (compile-time indices only: select per lane)
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (i == n_pending) s[i] = wm;
    p2::NoRecord rec;
    p2::permute_core<16>(s, p->rounds_p, p->ext_rc, p->int_rc, p->diag, p->ext_rc_mp, p->int_rc_mp, p->diag_c, rec, p->sum_mult_c);
    // the first sample after the permutation: lane squeeze - 1 (pop from the end) or lane 0 (pop from the front)
    uint32_t v = s[0];
#pragma unroll
    for (int i = 1; i < 16; i++)
        if (i == sample_lane) v = s[i];
    if ((bb::from_monty(v) & mask) == 0) atomicMin(best, wcan);
}

// ---------------------------------------------------------------- batched Merkle openings (MMCS open_batch)
struct GatherMat {
    const uint32_t* base;
    uint32_t width;
    uint32_t pitch;   // words between rows
    uint32_t log_h;
    uint32_t out_off;  // word offset of this matrix's row inside a query's record
};

// one workgroup per query: record = [rows of every matrix back to back | log_max sibling digests, leaf level first]
__global__ __launch_bounds__(256) void k_gather_openings(const GatherMat* __restrict__ mats, uint32_t n_mats, const uint32_t* __restrict__ digests,
                                                          const uint64_t* __restrict__ level_off, uint32_t log_max, uint32_t rows_words,
                                                          const uint32_t* __restrict__ indices, uint32_t shift, uint32_t* __restrict__ out) {
    const uint32_t q = blockIdx.x;
    const uint32_t index = indices[q] >> shift;
    uint32_t* rec = out + (size_t)q * (rows_words + 8 * log_max);
    for (uint32_t m = 0; m < n_mats; m++) {
        const GatherMat g = mats[m];
        const size_t r = index >> (log_max - g.log_h);
        for (uint32_t c = threadIdx.x; c < g.width; c += blockDim.x) rec[g.out_off + c] = g.base[r * g.pitch + c];
    }
    for (uint32_t t = threadIdx.x; t < 8 * log_max; t += blockDim.x) {
        const uint32_t l = t >> 3, k = t & 7;
        const size_t sib = (index >> l) ^ 1u;
        rec[rows_words + t] = digests[(level_off[l] + sib) * 8 + k];
    }
}

// The same with the matrix table and the level offsets passed by value in the kernel arguments (up to GATHER_INLINE matrices):
// no upload, no host wait for staging buffers -- a proof opens two dozen commitments back to back.
constexpr int GATHER_INLINE = 96;
struct GatherInline {
    GatherMat mats[GATHER_INLINE];
    uint64_t level_off[32];
};
__global__ __launch_bounds__(256) void k_gather_openings_inline(GatherInline t, uint32_t n_mats, const uint32_t* __restrict__ digests,
                                                                 uint32_t log_max, uint32_t rows_words, const uint32_t* __restrict__ indices,
                                                                 uint32_t shift, uint32_t* __restrict__ out) {
    const uint32_t q = blockIdx.x;
    const uint32_t index = indices[q] >> shift;
    uint32_t* rec = out + (size_t)q * (rows_words + 8 * log_max);
    for (uint32_t m = 0; m < n_mats; m++) {
        const GatherMat g = t.mats[m];
        const size_t r = index >> (log_max - g.log_h);
        for (uint32_t c = threadIdx.x; c < g.width; c += blockDim.x) rec[g.out_off + c] = g.base[r * g.pitch + c];
    }
    for (uint32_t e = threadIdx.x; e < 8 * log_max; e += blockDim.x) {
        const uint32_t l = e >> 3, k = e & 7;
        const size_t sib = (index >> l) ^ 1u;
        rec[rows_words + e] = digests[(t.level_off[l] + sib) * 8 + k];
    }
}

}  // namespace

int32_t point_weights(lurkhip_ctx* ctx, int mode, int log_m, const bb::ef& z, uint32_t* out_dev) {
    const uint32_t m = 1u << log_m, threads = (m + PW_BATCH - 1) / PW_BATCH;
    const NttPlan* plan = nullptr;
    LH_TRY(get_ntt_plan(ctx, log_m, &plan));
    // the barycentric weights (mode 0) only feed k_column_dot's lazy accumulators: centred
    hipLaunchKernelGGL(k_point_weights, dim3((threads + 255) / 256), dim3(256), 0, ctx->stream, mode, log_m, bb::to_monty(bb::GEN),
                       (const uint32_t*)plan->tw_fwd, z, out_dev, mode == 0 ? 1 : 0);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t point_weights_batch(lurkhip_ctx* ctx, const std::vector<WeightJob>& jobs) {
    for (size_t at = 0; at < jobs.size(); at += PW_BATCH_MAX) {
        PwBatchArgs a{};
        const size_t n = std::min<size_t>(PW_BATCH_MAX, jobs.size() - at);
        uint32_t max_threads = 1;
        for (size_t k = 0; k < n; k++) {
            const WeightJob& j = jobs[at + k];
            const NttPlan* plan = nullptr;
            LH_TRY(get_ntt_plan(ctx, j.log_m, &plan));
            a.mode[k] = j.mode;
            a.log_m[k] = j.log_m;
            a.tw[k] = (const uint32_t*)plan->tw_fwd;
            a.z[k] = j.z;
            a.out[k] = j.out;
            max_threads = std::max(max_threads, ((1u << j.log_m) + PW_BATCH - 1) / PW_BATCH);
        }
        hipLaunchKernelGGL(k_point_weights_batch, dim3((max_threads + 255) / 256, (unsigned)n), dim3(256), 0, ctx->stream, a, bb::to_monty(bb::GEN));
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

// rows per partial-sum block: 1024, fewer for short matrices on the wave-per-row kernel so that the launch still fills the CUs
constexpr uint32_t DOT_WAVE_MIN_W = 24;
constexpr uint32_t DOT_SLAB_MAX_W = 256;  // k_column_dot_slab: DOT_WAVE_MIN_W .. 256 columns (LURKHIP_DOT_SLAB=0: the wave kernel, A/B)
static bool dot_slab(uint32_t w) {
    static const bool on = getenv("LURKHIP_DOT_SLAB") == nullptr || atoi(getenv("LURKHIP_DOT_SLAB")) != 0;
    return on && w >= DOT_WAVE_MIN_W && w <= DOT_SLAB_MAX_W && getenv("LURKHIP_DOT_OLD") == nullptr;
}
static uint32_t dot_block_rows(uint32_t w, size_t n_rows) {
    uint32_t rb = DOT_ROWS;
    if (w < DOT_WAVE_MIN_W || getenv("LURKHIP_DOT_OLD") != nullptr) return rb;
    if (dot_slab(w)) {  // one workgroup per row block: 512 rows (16 KiB of staged weights), fewer while the launch is short of workgroups
        rb = 512;
        while (rb > 64 && (n_rows + rb - 1) / rb < 1024) rb /= 2;
        return rb;
    }
    const size_t chunks = (w + 63) / 64;
    while (rb > 64 && ((n_rows + rb - 1) / rb) * chunks < 512) rb /= 2;
    return rb;
}
static uint32_t dot_blocks(uint32_t w, size_t n_rows) {
    const uint32_t rb = dot_block_rows(w, n_rows);
    return (uint32_t)((n_rows + rb - 1) / rb);
}

size_t column_dot_partial_words(uint32_t w, size_t n_rows) { return (size_t)dot_blocks(w, n_rows) * 2 * w * 4; }

int32_t column_dot_partial(lurkhip_ctx* ctx, const uint32_t* mat, uint32_t w, uint32_t pitch, size_t n_rows, const uint32_t* u0, const uint32_t* u1,
                           uint32_t* partial_dev) {
    if (pitch == 0) pitch = w;
    const uint32_t n_blocks = dot_blocks(w, n_rows);
    if (dot_slab(w)) {
        const uint32_t rb = dot_block_rows(w, n_rows);
        const int v = w > 128 ? 2 : 1;
        const size_t lds = ((size_t)rb * 8 + (size_t)2 * v * 256 * 4) * 4;
        if (v == 2) hipLaunchKernelGGL(k_column_dot_slab<2>, dim3(n_blocks), dim3(256), lds, ctx->stream, mat, w, pitch, n_rows, rb, u0, u1, partial_dev);
        else hipLaunchKernelGGL(k_column_dot_slab<1>, dim3(n_blocks), dim3(256), lds, ctx->stream, mat, w, pitch, n_rows, rb, u0, u1, partial_dev);
    } else if (w >= DOT_WAVE_MIN_W && getenv("LURKHIP_DOT_OLD") == nullptr) {  // LURKHIP_DOT_OLD: A/B hook, every width on k_column_dot
        const uint32_t n_chunks = (w + 63) / 64;
        hipLaunchKernelGGL(k_column_dot_wave, dim3(n_blocks * n_chunks), dim3(256), 0, ctx->stream, mat, w, pitch, n_rows, dot_block_rows(w, n_rows), n_chunks,
                           u0, u1, partial_dev);
    } else
        hipLaunchKernelGGL(k_column_dot, dim3(n_blocks), dim3(256), 0, ctx->stream, mat, w, pitch, n_rows, u0, u1, partial_dev);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

bool column_dot_is_narrow(uint32_t w) { return w < DOT_WAVE_MIN_W || getenv("LURKHIP_DOT_OLD") != nullptr; }

int32_t column_dot_partial_batch(lurkhip_ctx* ctx, const std::vector<NarrowDot>& items) {
    for (size_t at = 0; at < items.size(); at += NARROW_DOT_MAX) {
        NarrowDotArgs a{};
        const size_t n = std::min<size_t>(NARROW_DOT_MAX, items.size() - at);
        uint32_t max_blocks = 1;
        for (size_t k = 0; k < n; k++) {
            const NarrowDot& d = items[at + k];
            a.mat[k] = d.mat;
            a.u0[k] = d.u0;
            a.u1[k] = d.u1;
            a.partial[k] = d.partial;
            a.w[k] = d.w;
            a.pitch[k] = d.pitch ? d.pitch : d.w;
            a.n_rows[k] = (uint32_t)d.n_rows;
            max_blocks = std::max(max_blocks, dot_blocks(d.w, d.n_rows));
        }
        hipLaunchKernelGGL(k_column_dot_batch, dim3(max_blocks, (unsigned)n), dim3(256), 0, ctx->stream, a);
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t column_dot_finish(lurkhip_ctx* ctx, const std::vector<DotJob>& jobs, uint32_t* out_dev) {
    for (size_t at = 0; at < jobs.size(); at += DOT_FINISH_MAX) {
        DotFinishArgs a{};
        a.n = (uint32_t)std::min<size_t>(DOT_FINISH_MAX, jobs.size() - at);
        a.out = out_dev;
        uint32_t cols = 0;
        for (uint32_t m = 0; m < a.n; m++) {
            const DotJob& j = jobs[at + m];
            a.col_start[m] = cols;
            a.partial[m] = j.partial;
            a.w[m] = j.w;
            a.n_blocks[m] = dot_blocks(j.w, j.n_rows);
            a.out_off[m] = j.out_off;
            cols += j.w * (j.two_points ? 2u : 1u);
        }
        a.col_start[a.n] = cols;
        if (cols) hipLaunchKernelGGL(k_dot_finish_all, dim3(cols), dim3(256), 0, ctx->stream, a);
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t reduce_openings(lurkhip_ctx* ctx, const uint32_t* mat, uint32_t w, uint32_t m_rows, const uint32_t* alpha_pows,
                        const uint32_t* alpha_pows_centred, const uint32_t* d0, const uint32_t* d1, const bb::ef& ys0,
                        const bb::ef& ys1, const bb::ef& apow0, const bb::ef& apow1, uint32_t* ro) {
    if (w > 16 && w <= 128 && alpha_pows_centred && getenv("LURKHIP_OPENINGS_NO_QUAD") == nullptr) {
        // four lanes per row, 16-row tiles (k_reduce_openings_quad)
        ReduceArgs a{mat, w, m_rows, alpha_pows_centred, d0, d1, ys0, ys1, apow0, apow1, ro, 1};
        const uint32_t n_tiles = (m_rows + 15) / 16;
        const bool big = (size_t)m_rows * w >= ((size_t)1 << 30) || getenv("LURKHIP_OPENINGS_FORCE_64BIT") != nullptr;
        auto launch = [&](auto kernel, auto kernel_big, int nv) {
            const size_t lds = ((size_t)nv * 66 + 8 + 4 * (size_t)w) * 4;
            const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(32, (160 * 1024) / (lds + 512)));
            const unsigned blocks = (unsigned)std::min<size_t>(n_tiles, (size_t)per_cu * ctx->num_cus);
            if (big) hipLaunchKernelGGL(kernel_big, dim3(blocks), dim3(64), lds, ctx->stream, a);
            else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), lds, ctx->stream, a);
        };
#define LH_RQ(NV) launch(k_reduce_openings_quad<NV, false>, k_reduce_openings_quad<NV, true>, NV)
        const uint32_t need = (16 * w + 63) / 64;  // words per lane and tile
        if (need <= 8) LH_RQ(8);
        else if (need <= 12) LH_RQ(12);
        else if (need <= 16) LH_RQ(16);
        else if (need <= 20) LH_RQ(20);
        else if (need <= 24) LH_RQ(24);
        else if (need <= 28) LH_RQ(28);
        else LH_RQ(32);
#undef LH_RQ
        LH_HIP(ctx, hipGetLastError());
        return LURKHIP_OK;
    }
    if (w <= 128 && alpha_pows_centred) {
        ReduceArgs a{mat, w, m_rows, alpha_pows_centred, d0, d1, ys0, ys1, apow0, apow1, ro, 1};
        const uint32_t n_tiles = (m_rows + 63) / 64;
        // 4 GiB of words and more: 64-bit word indices; LURKHIP_OPENINGS_FORCE_64BIT (test hook) takes that path at any size
        const bool big = (size_t)m_rows * w >= ((size_t)1 << 30) || getenv("LURKHIP_OPENINGS_FORCE_64BIT") != nullptr;
        auto launch = [&](auto kernel, auto kernel_big, int nv) {
            const size_t lds = ((size_t)nv * 66 + 64 + 8 + 4 * (size_t)nv) * 4;  // tile + the alpha powers of its columns
            const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / (lds + 512)));
            const unsigned blocks = (unsigned)std::min<size_t>(n_tiles, (size_t)per_cu * ctx->num_cus);
            if (big) hipLaunchKernelGGL(kernel_big, dim3(blocks), dim3(64), lds, ctx->stream, a);
            else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), lds, ctx->stream, a);
        };
#define LH_RO(NV) launch(k_reduce_openings_stream<NV, false>, k_reduce_openings_stream<NV, true>, NV)
        if (w <= 8) LH_RO(8);
        else if (w <= 16) LH_RO(16);
        else if (w <= 24) LH_RO(24);
        else if (w <= 32) LH_RO(32);
        else if (w <= 48) LH_RO(48);
        else if (w <= 64) LH_RO(64);
        else if (w <= 80) LH_RO(80);
        else if (w <= 96) LH_RO(96);
        else if (w <= 112) LH_RO(112);
        else LH_RO(128);
#undef LH_RO
        LH_HIP(ctx, hipGetLastError());
        return LURKHIP_OK;
    }
    const size_t lds = (size_t)64 * (w | 1u) * 4;
    const bool staged = lds <= 64 * 1024;
    ReduceArgs a{mat, w, m_rows, alpha_pows, d0, d1, ys0, ys1, apow0, apow1, ro, staged ? 1 : 0};
    hipLaunchKernelGGL(k_reduce_openings, dim3((m_rows + 63) / 64), dim3(64), staged ? lds : 0, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t reduce_openings_wide(lurkhip_ctx* ctx, WideArgs a) {
    LH_ARG(ctx, a.n_slices >= 1 && a.n_slices <= WIDE_MAX_SLICES, "reduce_openings_wide: slice count");
    uint32_t max_sw = 0;
    for (uint32_t i = 0; i < a.n_slices; i++) {
        LH_ARG(ctx, a.sw[i] >= 1 && a.sw[i] <= 128 && a.c0[i] + a.sw[i] <= a.w, "reduce_openings_wide: slice %u", i);
        a.magic[i] = (uint32_t)((((uint64_t)1 << 32) + a.sw[i] - 1) / a.sw[i]);
        max_sw = std::max(max_sw, a.sw[i]);
    }
    const uint32_t n_tiles = (a.m_rows + 63) / 64;
    const bool big = (size_t)a.m_rows * a.w >= ((size_t)1 << 30) || getenv("LURKHIP_OPENINGS_FORCE_64BIT") != nullptr;
    auto launch = [&](auto kernel, auto kernel_big, int nv) {
        const size_t lds = ((size_t)nv * 66 + 64 + 8 + 4 * (size_t)nv) * 4;
        const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / (lds + 512)));
        const unsigned blocks = (unsigned)std::min<size_t>(n_tiles, (size_t)per_cu * ctx->num_cus);
        if (big) hipLaunchKernelGGL(kernel_big, dim3(blocks), dim3(64), lds, ctx->stream, a);
        else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), lds, ctx->stream, a);
    };
#define LH_RW(NV) launch(k_reduce_openings_wide<NV, false>, k_reduce_openings_wide<NV, true>, NV)
    if (max_sw <= 32) LH_RW(32);
    else if (max_sw <= 64) LH_RW(64);
    else if (max_sw <= 80) LH_RW(80);
    else if (max_sw <= 96) LH_RW(96);
    else if (max_sw <= 112) LH_RW(112);
    else LH_RW(128);
#undef LH_RW
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

// Round 4: four lanes to a row, no staging.  The streaming kernels above bring a tile into LDS with coalesced dword loads and
// read it back row-wise: per matrix word a load, an LDS store, an LDS load and their address arithmetic around the four
// multiply-adds that are the work -- about 30 instructions per word, VALU-bound at a quarter of the memory rate (PMC round 3:
// 22 T lane-instr/s, 2-3 TB/s).  Here the lanes of a quad take the 16-byte pieces k = part, part + 4, ... of their row (dword-
// aligned 16-byte loads, which gfx950 allows): a wave reads 16 rows, i.e. one contiguous run of memory when the matrix is dense,
// every line of it within a few instructions.  Per piece: one load, four 16-byte LDS reads of the alpha powers, sixteen
// multiply-adds and eight fold multiply-adds.  The last piece of a row is the four columns w - 4 .. w - 1 with zero weights
// for the columns its neighbour already took (a table of four weights per matrix), so nothing is read past a row and any
// pitch and column offset work: matrices with their own buffer and column ranges of a padded group alike.  All the matrices
// of a height share the launch: ro / d0 / d1 are touched once per row.
typedef const __attribute__((address_space(1))) uint32_t* gwords;
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) u32x4_a4* gquads;
__global__ __launch_bounds__(256) void k_reduce_openings_rows(RowsArgs a) {
    extern __shared__ uint32_t rows_lds[];
    uint32_t* __restrict__ tbl = rows_lds;                      // [max_w][4]: centred alpha^c
    uint32_t* __restrict__ tails = rows_lds + 4 * a.max_w;      // [n_mats][4][4]: the weights of a row's last piece
    uint32_t* __restrict__ zeros = tails + 16 * ROWS_MAX_MATS;  // [4][4]: the weights of a piece past the row
    for (uint32_t e = threadIdx.x; e < 4u * a.max_w; e += 256u) tbl[e] = a.alpha_pows[8u * (e >> 2) + (e & 3u)];
    for (uint32_t e = threadIdx.x; e < 16u * a.n_mats; e += 256u) {
        const uint32_t m = e >> 4, j = (e >> 2) & 3u, w = a.m[m].w;
        const uint32_t c = w - 4u + j, first = 4u * ((w + 3u) / 4u - 1u);  // first column of the last piece proper
        tails[e] = c >= first ? a.alpha_pows[8u * c + (e & 3u)] : 0u;
    }
    if (threadIdx.x < 16u) zeros[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t part = threadIdx.x & 3u, r_in = threadIdx.x >> 2;
    const uint32_t n_tiles = (a.m_rows + 63u) / 64u;
    constexpr int UN = 4;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t s = t * 64u + r_in;
        const bool live = s < a.m_rows;
        const size_t sr = live ? s : a.m_rows - 1;
        ef g = bb::ef_zero();  // lane 0 of the quad: the first point's sum, lane 1: the second point's
        for (uint32_t m = 0; m < a.n_mats; m++) {
            const RowsMat& rm = a.m[m];
            const uint32_t w = rm.w, n_pieces = (w + 3u) / 4u, n_iter = (n_pieces + 3u) / 4u;
            gwords row = (gwords)rm.mat + sr * rm.pitch;
            const uint32_t* __restrict__ tail = tails + 16u * m;
            // Four 64-bit lanes of R * (sum of value x centred weight), two terms between folds on a fixed schedule (lazy_ef.h:
            // a folded lane is below 0.08 p^2, a term below 0.5 p^2, the reduction takes 1.2 p^2)
            int64_t acc[4] = {0, 0, 0, 0};
            for (uint32_t kb = 0; kb < n_iter; kb += UN) {
                u32x4_a4 v[UN];
                const uint32_t* wp[UN];
#pragma unroll
                for (int j = 0; j < UN; j++) {
                    const uint32_t k = (kb + (uint32_t)j) * 4u + part;
                    const bool valid = k < n_pieces, last = k + 1u == n_pieces;
                    v[j] = *(gquads)(row + (!valid ? 0u : last ? w - 4u : 4u * k));  // (a piece past the row re-reads its start: zero weights)
                    wp[j] = !valid ? zeros : last ? tail : tbl + 16u * k;
                }
#pragma unroll
                for (int j = 0; j < UN; j++) {
                    if (kb + (uint32_t)j >= n_iter) break;  // (uniform)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const uint4 q0 = *reinterpret_cast<const uint4*>(wp[j] + 4 * e), q1 = *reinterpret_cast<const uint4*>(wp[j] + 4 * e + 4);
                        const int32_t v0 = (int32_t)v[j][e], v1 = (int32_t)v[j][e + 1];
                        acc[0] = bb::mad_i64(v1, (int32_t)q1.x, bb::mad_i64(v0, (int32_t)q0.x, acc[0]));
                        acc[1] = bb::mad_i64(v1, (int32_t)q1.y, bb::mad_i64(v0, (int32_t)q0.y, acc[1]));
                        acc[2] = bb::mad_i64(v1, (int32_t)q1.z, bb::mad_i64(v0, (int32_t)q0.z, acc[2]));
                        acc[3] = bb::mad_i64(v1, (int32_t)q1.w, bb::mad_i64(v0, (int32_t)q0.w, acc[3]));
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[c] = bb::mad_i64((int32_t)(acc[c] >> 32), (int32_t)bb::R1, (int64_t)(uint64_t)(uint32_t)acc[c]);
                    }
                }
            }
            ef x;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x.c[k] = bb::canon(bb::sred(acc[k]));
                x.c[k] = bb::add(x.c[k], dpp<DPP_QUAD_XOR1>(x.c[k]));
                x.c[k] = bb::add(x.c[k], dpp<DPP_QUAD_XOR2>(x.c[k]));
            }
            const bool second = part == 1u;
            ef ys, ap;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ys.c[k] = second ? rm.ys1.c[k] : rm.ys0.c[k];
                ap.c[k] = second ? rm.apow1.c[k] : rm.apow0.c[k];
            }
            const ef term = bb::ef_mul(ap, bb::ef_sub(x, ys));
            if (!second || rm.two) g = bb::ef_add(g, term);
        }
        const uint32_t* __restrict__ dp = part == 1u && a.d1 ? a.d1 : a.d0;
        ef term = bb::ef_mul(g, ef_load(dp + 4 * sr));
        if (part >= 2u || (part == 1u && !a.d1)) term = bb::ef_zero();
#pragma unroll
        for (int k = 0; k < 4; k++) term.c[k] = bb::add(term.c[k], dpp<DPP_QUAD_XOR1>(term.c[k]));
        if (part == 0u && live) ef_store(a.ro + 4 * (size_t)s, bb::ef_add(ef_load(a.ro + 4 * (size_t)s), term));
    }
}

int32_t reduce_openings_rows(lurkhip_ctx* ctx, RowsArgs a) {
    if (a.n_mats == 0) return LURKHIP_OK;
    LH_ARG(ctx, a.n_mats <= ROWS_MAX_MATS, "reduce_openings_rows: %u matrices", a.n_mats);
    a.max_w = 0;
    for (uint32_t m = 0; m < a.n_mats; m++) {
        LH_ARG(ctx, a.m[m].w >= 4 && a.m[m].w <= ROWS_MAX_W && a.m[m].pitch >= a.m[m].w, "reduce_openings_rows: matrix %u", m);
        a.max_w = std::max(a.max_w, a.m[m].w);
    }
    const size_t lds = ((size_t)4 * a.max_w + 16 * ROWS_MAX_MATS + 16) * 4;
    const uint32_t n_tiles = (a.m_rows + 63) / 64;
    static const int per_cu = getenv("LURKHIP_ROWS_PER_CU") ? std::max(1, atoi(getenv("LURKHIP_ROWS_PER_CU"))) : 8;
    const unsigned blocks = (unsigned)std::min<size_t>(n_tiles, (size_t)ctx->num_cus * per_cu);
    hipLaunchKernelGGL(k_reduce_openings_rows, dim3(blocks), dim3(256), lds, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t reduce_openings_narrow(lurkhip_ctx* ctx, const NarrowArgs& a) {
    if (a.n_mats == 0) return LURKHIP_OK;
    hipLaunchKernelGGL(k_reduce_openings_narrow, dim3((a.m_rows + 255) / 256), dim3(256), 0, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t fri_fold(lurkhip_ctx* ctx, const uint32_t* cur, int log_len, const uint32_t* beta_dev, const uint32_t* add, uint32_t* out, uint32_t pair_base,
                 uint32_t n_pairs) {
    const uint32_t half_m = bb::pow(bb::to_monty(2), bb::P - 2);
    const uint32_t ginv = bb::pow(two_adic_generator_monty(log_len), bb::P - 2);
    const uint32_t half_len = n_pairs ? n_pairs : 1u << (log_len - 1);
    hipLaunchKernelGGL(k_fri_fold, dim3((half_len + 255) / 256), dim3(256), 0, ctx->stream, cur, log_len, beta_dev, ginv, half_m, add, out, pair_base, half_len);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t fri_challenge(lurkhip_ctx* ctx, DevChallenger* ch_dev, const uint32_t* root_dev, uint32_t* beta_dev, uint32_t* root_copy_dev) {
    const P16Params* params = nullptr;
    LH_TRY(get_merkle_params(ctx, &params));
    hipLaunchKernelGGL(k_fri_challenge, dim3(1), dim3(64), 0, ctx->stream, params, ch_dev, root_dev, beta_dev, root_copy_dev);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t pow_grind(lurkhip_ctx* ctx, const uint32_t state_with_pending_m[16], int n_pending, int bits, int sample_lane, uint32_t* witness) {
    const P16Params* params = nullptr;
    LH_TRY(get_merkle_params(ctx, &params));
    void* scratch = nullptr;
    LH_TRY(pool_alloc(ctx, 16, &scratch));
    PowState st;
    for (int i = 0; i < 16; i++) st.s[i] = state_with_pending_m[i];
    int32_t s = LURKHIP_OK;
    hipError_t e = hipMemsetAsync(scratch, 0xff, 4, ctx->stream);  // "no witness yet"
    const uint32_t mask = bits >= 31 ? 0x7fffffffu : ((1u << bits) - 1u);
    // Candidates in increasing ranges, the smallest accepted one of the first range that has any: a range of 2^(bits + 1)
    // candidates holds a witness with probability 1 - e^-2, so the first launch is that size (2^17 permutations for 16 bits,
    // where a fixed 2^20 cost 0.15 ms of every proof) and the ranges double from there up to 2^20.
    uint64_t batch = (uint64_t)1 << std::min(20, std::max(12, bits + 1));
    uint32_t best = 0xffffffffu;
    volatile uint32_t* best_pin = nullptr;
    {
        void* pin = nullptr;
        const int32_t ps = pinned_small(ctx, &pin);
        if (ps != LURKHIP_OK) {
            pool_release(ctx, scratch);
            return ps;
        }
        best_pin = (volatile uint32_t*)((uint8_t*)pin + 64);
    }
    for (uint64_t base = 0; e == hipSuccess && base < bb::P && best == 0xffffffffu; base += batch, batch = std::min<uint64_t>(batch * 2, (uint64_t)1 << 20)) {
        // (base advanced by the range just searched, then the next range's size)
        hipLaunchKernelGGL(k_pow_grind, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx->stream, params, st, n_pending, sample_lane,
                           (uint32_t)base, mask, (uint32_t*)scratch);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(const_cast<uint32_t*>(best_pin), scratch, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = stream_wait(ctx);
        if (e == hipSuccess) best = *best_pin;
    }
    pool_release(ctx, scratch);
    if (e != hipSuccess) s = set_error(ctx, LURKHIP_ERR_HIP, "pow_grind failed: %s", hipGetErrorString(e));
    else if (best == 0xffffffffu) s = set_error(ctx, LURKHIP_ERR_EXEC, "no proof-of-work witness found");
    *witness = best;
    return s;
}

int32_t gather_openings(lurkhip_ctx* ctx, const std::vector<OpenMat>& mats, const uint32_t* digests, const std::vector<size_t>& level_off,
                        uint32_t log_max, const uint32_t* indices_dev, uint32_t n_queries, uint32_t shift, uint32_t* out_dev,
                        uint32_t* record_words) {
    std::vector<GatherMat> g(mats.size());
    uint32_t off = 0;
    for (size_t i = 0; i < mats.size(); i++) {
        g[i] = GatherMat{mats[i].base, mats[i].width, mats[i].pitch ? mats[i].pitch : mats[i].width, mats[i].log_h, off};
        off += mats[i].width;
    }
    *record_words = off + 8 * log_max;
    if (!out_dev) return LURKHIP_OK;  // size query
    if (mats.size() <= (size_t)GATHER_INLINE && level_off.size() <= 32) {
        GatherInline t{};
        for (size_t i = 0; i < g.size(); i++) t.mats[i] = g[i];
        for (size_t i = 0; i < level_off.size(); i++) t.level_off[i] = level_off[i];
        hipLaunchKernelGGL(k_gather_openings_inline, dim3(n_queries), dim3(256), 0, ctx->stream, t, (uint32_t)g.size(), digests, log_max, off,
                           indices_dev, shift, out_dev);
        LH_HIP(ctx, hipGetLastError());
        return LURKHIP_OK;
    }
    std::vector<uint64_t> lo(level_off.begin(), level_off.end());
    void* scratch = nullptr;
    const size_t b_g = g.size() * sizeof(GatherMat), b_lo = lo.size() * 8, o_lo = (b_g + 15) & ~(size_t)15;
    LH_TRY(pool_alloc(ctx, o_lo + b_lo + 16, &scratch));
    hipError_t e = hipMemcpyAsync(scratch, g.data(), b_g, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((uint8_t*)scratch + o_lo, lo.data(), b_lo, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = stream_wait(ctx);  // the staging vectors die at scope exit
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_openings, dim3(n_queries), dim3(256), 0, ctx->stream, (const GatherMat*)scratch, (uint32_t)g.size(), digests,
                           (const uint64_t*)((uint8_t*)scratch + o_lo), log_max, off, indices_dev, shift, out_dev);
        e = hipGetLastError();
    }
    pool_release(ctx, scratch);
    if (e != hipSuccess) return set_error(ctx, LURKHIP_ERR_HIP, "gather_openings failed: %s", hipGetErrorString(e));
    return LURKHIP_OK;
}

}  // namespace lurkhip
