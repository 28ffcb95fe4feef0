// Batched Poseidon2 kernels (permute / hash8 / wide witness) and their C-ABI launchers.
//
// Roofline note (DESIGN.md "Poseidon2"): a width-24 permutation is ~1.36 k modular
// products for 128 algorithmic bytes, so these kernels are int32-VALU bound by an
// order of magnitude; the memory side only has to stay out of the way.  Inputs are
// read as 16-byte vectors per lane (row = one lane's contiguous W*4 bytes), the
// 32-byte digest is written as two 16-byte stores.  The wide-witness kernel writes
// ~2-3 KB per row: it stages each group of W columns for the 64 rows of a wave in
// an LDS tile and writes the tile out with lanes running along the row, so global
// stores are contiguous runs instead of 64 scattered dwords.
#include <string.h>

#include "ctx.h"
#include "poseidon2_dev.h"

namespace {

using namespace p2;

constexpr int BLOCK = 256;

template <int W>
__device__ __forceinline__ void load_state(const uint32_t* __restrict__ in, size_t row, uint32_t (&s)[W], bool canonical) {
    const uint4* src = reinterpret_cast<const uint4*>(in + row * W);
#pragma unroll
    for (int i = 0; i < W / 4; i++) {
        uint4 v = src[i];
        s[4 * i + 0] = v.x;
        s[4 * i + 1] = v.y;
        s[4 * i + 2] = v.z;
        s[4 * i + 3] = v.w;
    }
    if (canonical) {
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = bb::to_monty(s[i]);
    }
}

// OUT = number of leading lanes written (W for permute, 8 for hash8)
template <int W, int OUT>
__global__ __launch_bounds__(BLOCK) void k_permute(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                                                    int canonical) {
    size_t row = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (row >= n) return;
    uint32_t s[W];
    load_state<W>(in, row, s, canonical != 0);
    permute<W>(s);
    if (canonical) {
#pragma unroll
        for (int i = 0; i < OUT; i++) s[i] = bb::from_monty(s[i]);
    }
    uint4* dst = reinterpret_cast<uint4*>(out + row * OUT);
#pragma unroll
    for (int i = 0; i < OUT / 4; i++) dst[i] = make_uint4(s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
}

// ---- wide witness -----------------------------------------------------------
// LDS tile per wave: 64 rows x TS dwords, TS odd so lane-strided ds_write_b32 is conflict free.
template <int W, int RP>
struct TileRec {
    static constexpr int INT_COLS = 2 * RP - 1;
    static constexpr int TS = ((W > INT_COLS ? W : INT_COLS) | 1);
    uint32_t* tile;       // this wave's tile base
    uint32_t* out;        // output matrix
    size_t row0;          // first row of this wave
    size_t n;             // total rows
    int lane;
    bool canonical;
    static constexpr int STRIDE = 8 + 16 * W + W + INT_COLS;

    __device__ __forceinline__ void put(int col, uint32_t v) { tile[lane * TS + col] = v; }
    // write `cols` staged columns of all 64 rows to out[:, base : base + cols]
    __device__ __forceinline__ void flush(int base, int cols) {
        __syncthreads();
        for (int e = lane; e < 64 * cols; e += 64) {
            int r = e / cols, c = e - r * cols;
            size_t row = row0 + r;
            if (row < n) {
                uint32_t v = tile[r * TS + c];
                if (canonical) v = bb::from_monty(v);
                out[row * STRIDE + base + c] = v;
            }
        }
        __syncthreads();
    }
    __device__ __forceinline__ void ext_state(int, int i, uint32_t v) { put(i, v); }
    __device__ __forceinline__ void end_ext_state(int r) { flush(8 + r * W, W); }
    __device__ __forceinline__ void ext_sbox(int, int i, uint32_t v) { put(i, v); }
    __device__ __forceinline__ void end_ext_sbox(int r) { flush(8 + 8 * W + r * W, W); }
    __device__ __forceinline__ void int_init(int i, uint32_t v) { put(i, v); }
    __device__ __forceinline__ void end_int_init() { flush(8 + 16 * W, W); }
    __device__ __forceinline__ void int_state0(int r, uint32_t v) { put(r, v); }
    __device__ __forceinline__ void int_sbox(int r, uint32_t v) { put(RP - 1 + r, v); }
    __device__ __forceinline__ void end_internal() { flush(8 + 17 * W, INT_COLS); }
};

template <int W>
__global__ __launch_bounds__(BLOCK) void k_wide_witness(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                         size_t n, int canonical) {
    constexpr int RP = Cfg<W>::RP;
    using Rec = TileRec<W, RP>;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    size_t row = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t src_row = row < n ? row : n - 1;  // keep every lane alive for the tile barriers
    uint32_t s[W];
    load_state<W>(in, src_row, s, canonical != 0);
    Rec rec;
    rec.tile = smem + wave * 64 * Rec::TS;
    rec.out = out;
    rec.row0 = (size_t)blockIdx.x * BLOCK + wave * 64;
    rec.n = n;
    rec.lane = lane;
    rec.canonical = canonical != 0;
    const auto& p = Cfg<W>::params();
    permute_core<W>(s, RP, p.ext_rc, p.int_rc, p.diag, p.ext_rc_mp, p.int_rc_mp, p.diag_c, rec);
    // the 8 output lanes lead the row (core/poseidon.rs:66-71)
#pragma unroll
    for (int i = 0; i < 8; i++) rec.put(i, s[i]);
    rec.flush(0, 8);
}

// ---- narrow chip trace (one row per round) -----------------------------------
// One permutation per lane; at every round the 64 permutations of a wave emit 64 rows that lie (R + 1) rows apart in the
// output, each a contiguous run of NC words.  A group of W columns is staged in the wave's LDS tile and written with lanes
// running along the row, like the wide witness above.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_narrow_trace(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                                                         int canonical) {
    constexpr int RP = Cfg<W>::RP, R = 8 + RP, NC = 5 * W + 1 + R, TS = W | 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* tile = smem + wave * 64 * TS;
    const size_t perm0 = (size_t)blockIdx.x * BLOCK + wave * 64;
    const size_t perm = perm0 + lane;
    const auto& p = Cfg<W>::params();
    const uint32_t one = canonical ? 1u : bb::to_monty(1u);
    uint32_t s[W], x[W], t[W];
    load_state<W>(in, perm < n ? perm : n - 1, s, canonical != 0);

    // tile columns [0, W) of the wave's 64 permutations -> out[(perm * (R + 1) + row), base : base + W]
    auto flush = [&](const uint32_t (&v)[W], int row, int base) {
#pragma unroll
        for (int i = 0; i < W; i++) tile[lane * TS + i] = v[i];
        __syncthreads();
        for (int e = lane; e < 64 * W; e += 64) {
            const int r = e / W, c = e - r * W;
            if (perm0 + r < n) {
                uint32_t val = tile[r * TS + c];
                if (canonical) val = bb::from_monty(val);
                out[((perm0 + r) * (size_t)(R + 1) + row) * NC + base + c] = val;
            }
        }
        __syncthreads();
    };
    // is_init | rounds[R]: one-hot at `row`
    auto flags = [&](int row) {
        for (int e = lane; e < 64 * (R + 1); e += 64) {
            const int r = e / (R + 1), c = e - r * (R + 1);
            if (perm0 + r < n) out[((perm0 + r) * (size_t)(R + 1) + row) * NC + W + c] = c == row ? one : 0u;
        }
    };
#pragma unroll 1
    for (int row = 0; row <= R; row++) {
        const int round = row - 1;
        const bool internal = round >= 4 && round < 4 + RP;
        flush(s, row, 0);
        flags(row);
#pragma unroll
        for (int i = 0; i < W; i++) x[i] = s[i];
        if (row > 0) {
            if (internal) {
                x[0] = bb::add(x[0], p.int_rc[round - 4]);
            } else {
                const uint32_t* rc = p.ext_rc + (round < 4 ? round : round - RP) * W;
#pragma unroll
                for (int i = 0; i < W; i++) x[i] = bb::add(x[i], rc[i]);
            }
        }
        flush(x, row, W + 1 + R);
#pragma unroll
        for (int i = 0; i < W; i++) t[i] = bb::cube(x[i]);
        flush(t, row, 2 * W + 1 + R);
#pragma unroll
        for (int i = 0; i < W; i++) t[i] = bb::pow7_from_cube(x[i], t[i]);
        flush(t, row, 3 * W + 1 + R);
        if (row > 0) {
            s[0] = t[0];
#pragma unroll
            for (int i = 1; i < W; i++) s[i] = internal ? x[i] : t[i];
        }
        if (internal) internal_layer<W>(s, p.diag);
        else external_layer<W>(s);
        flush(s, row, 4 * W + 1 + R);
    }
}

template <int W>
int32_t launch_narrow(lurkhip_ctx* ctx, size_t n, const uint32_t* in, uint32_t* out, int canonical) {
    constexpr int R = 8 + Cfg<W>::RP, NC = 5 * W + 1 + R;
    const size_t rows = n * (size_t)(R + 1);
    size_t height = 1;
    while (height < rows) height <<= 1;
    if (height > rows) LH_HIP(ctx, hipMemsetAsync(out + rows * NC, 0, (height - rows) * NC * sizeof(uint32_t), ctx->stream));
    if (n == 0) return LURKHIP_OK;
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    LH_ARG(ctx, blocks <= 0x7fffffffu, "n too large for one launch");
    size_t lds = (size_t)(BLOCK / 64) * 64 * (W | 1) * sizeof(uint32_t);
    hipLaunchKernelGGL((k_narrow_trace<W>), dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, in, out, n, canonical);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

template <int W, int OUT>
int32_t launch_permute(lurkhip_ctx* ctx, size_t n, const uint32_t* in, uint32_t* out, int canonical) {
    if (n == 0) return LURKHIP_OK;
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    LH_ARG(ctx, blocks <= 0x7fffffffu, "n too large for one launch");
    hipLaunchKernelGGL((k_permute<W, OUT>), dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream, in, out, n, canonical);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

template <int W>
int32_t launch_wide(lurkhip_ctx* ctx, size_t n, const uint32_t* in, uint32_t* out, int canonical) {
    if (n == 0) return LURKHIP_OK;
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    LH_ARG(ctx, blocks <= 0x7fffffffu, "n too large for one launch");
    using Rec = TileRec<W, Cfg<W>::RP>;
    size_t lds = (size_t)(BLOCK / 64) * 64 * Rec::TS * sizeof(uint32_t);
    hipLaunchKernelGGL((k_wide_witness<W>), dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, in, out, n, canonical);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

enum class Op { Permute, Hash8, Wide, Narrow };

int32_t dispatch(lurkhip_ctx* ctx, Op op, int32_t width, size_t n, const uint32_t* in, uint32_t* out, int canonical) {
    switch (width) {
#define CASE(W_, RP_)                                                                \
    case W_:                                                                         \
        if (op == Op::Narrow) return launch_narrow<W_>(ctx, n, in, out, canonical);   \
        if (op == Op::Permute) return launch_permute<W_, W_>(ctx, n, in, out, canonical); \
        if (op == Op::Hash8) {                                                       \
            if constexpr (W_ >= 8) return launch_permute<W_, 8>(ctx, n, in, out, canonical); \
            else return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "hash8 needs width >= 8"); \
        }                                                                            \
        if constexpr (W_ >= 8) return launch_wide<W_>(ctx, n, in, out, canonical);   \
        else return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "wide witness needs width >= 8");
        LURK_P2_WIDTHS(CASE)
#undef CASE
        default:
            return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "unsupported Poseidon2 width %d", width);
    }
}

size_t narrow_words(int32_t width, size_t n) {
    uint32_t rw = 0;
    uint64_t h = 0;
    lurkhip_poseidon2_trace_shape(width, n, &rw, &h);
    return (size_t)h * rw;
}

int32_t out_lanes(Op op, int32_t width) {
    if (op == Op::Permute) return width;
    if (op == Op::Hash8) return 8;
    return 8 + lurkhip_poseidon2_num_cols(width);
}

int32_t check_common(lurkhip_ctx* ctx, int32_t width, size_t n, const void* in, const void* out, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, lurkhip_poseidon2_num_cols(width) > 0, "unsupported Poseidon2 width %d", width);
    LH_ARG(ctx, repr == LURKHIP_REPR_CANONICAL || repr == LURKHIP_REPR_MONTY, "bad repr %d", repr);
    LH_ARG(ctx, n == 0 || (in && out), "null buffer with n = %zu", n);
    return LURKHIP_OK;
}

int32_t run_dev(lurkhip_ctx* ctx, Op op, int32_t width, size_t n, const uint32_t* in, uint32_t* out, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_TRY(check_common(ctx, width, n, in, out, repr));
    LH_ARG(ctx, ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "device buffers must be 16-byte aligned");
    LH_ARG(ctx, op != Op::Narrow || out, "null trace buffer");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    return dispatch(ctx, op, width, n, in, out, repr == LURKHIP_REPR_CANONICAL);
}

int32_t run_host(lurkhip_ctx* ctx, Op op, int32_t width, size_t n, const uint32_t* in, uint32_t* out, int32_t repr) {
    LH_CHECK_CTX(ctx);  // held for the whole call: the host variants share the context's scratch arenas
    LH_TRY(check_common(ctx, width, n, in, out, repr));
    if (n == 0) {
        // an empty batch still has a trace: one zero row (0.next_power_of_two() == 1, poseidon/trace.rs:20-23)
        if (op == Op::Narrow && out) memset(out, 0, narrow_words(width, 0) * 4);
        return LURKHIP_OK;
    }
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = n * (size_t)width * 4;
    const size_t out_bytes = (op == Op::Narrow ? narrow_words(width, n) : n * (size_t)out_lanes(op, width)) * 4;
    void *din = nullptr, *dout = nullptr;
    LH_TRY(lurkhip::arena_get(ctx, 0, in_bytes, &din));
    LH_TRY(lurkhip::arena_get(ctx, 1, out_bytes, &dout));
    LH_HIP(ctx, hipMemcpyAsync(din, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    LH_TRY(dispatch(ctx, op, width, n, (const uint32_t*)din, (uint32_t*)dout, repr == LURKHIP_REPR_CANONICAL));
    LH_HIP(ctx, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LURKHIP_OK;
}

}  // namespace

extern "C" {

int32_t lurkhip_poseidon2_num_cols(int32_t width) {
    for (int i = 0; i < LURK_P2_NUM_WIDTHS; i++)
        if (LURK_P2_PARAMS[i].width == width) {
            int rp = LURK_P2_PARAMS[i].rounds_p;
            return 16 * width + width + (rp - 1) + rp;
        }
    return LURKHIP_ERR_INVALID_ARG;
}

int32_t lurkhip_poseidon2_trace_shape(int32_t width, size_t n, uint32_t* row_width, uint64_t* height) {
    for (int i = 0; i < LURK_P2_NUM_WIDTHS; i++)
        if (LURK_P2_PARAMS[i].width == width) {
            const uint64_t per = 8 + (uint64_t)LURK_P2_PARAMS[i].rounds_p + 1, rows = (uint64_t)n * per;
            uint64_t h = 1;
            while (h < rows) h <<= 1;
            if (row_width) *row_width = (uint32_t)(5 * width + per);
            if (height) *height = h;
            return LURKHIP_OK;
        }
    return LURKHIP_ERR_INVALID_ARG;
}
int32_t lurkhip_poseidon2_trace(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out, int32_t repr) {
    return run_host(ctx, Op::Narrow, width, n, in, out, repr);
}
int32_t lurkhip_poseidon2_trace_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                    int32_t repr) {
    return run_dev(ctx, Op::Narrow, width, n, in, out, repr);
}

int32_t lurkhip_poseidon2_permute(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                  int32_t repr) {
    return run_host(ctx, Op::Permute, width, n, in, out, repr);
}
int32_t lurkhip_poseidon2_permute_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                      int32_t repr) {
    return run_dev(ctx, Op::Permute, width, n, in, out, repr);
}
int32_t lurkhip_poseidon2_hash8(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                int32_t repr) {
    return run_host(ctx, Op::Hash8, width, n, in, out, repr);
}
int32_t lurkhip_poseidon2_hash8_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                    int32_t repr) {
    return run_dev(ctx, Op::Hash8, width, n, in, out, repr);
}
int32_t lurkhip_poseidon2_wide_witness(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in,
                                       uint32_t* out, int32_t repr) {
    return run_host(ctx, Op::Wide, width, n, in, out, repr);
}
int32_t lurkhip_poseidon2_wide_witness_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in,
                                           uint32_t* out, int32_t repr) {
    return run_dev(ctx, Op::Wide, width, n, in, out, repr);
}

}  // extern "C"
