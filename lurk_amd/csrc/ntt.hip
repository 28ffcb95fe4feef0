// Coset low-degree extension of row-major trace matrices (radix-2 NTT over BabyBear).
//
// Replaces (S1 commit in SURVEY.md 8a; third-party, source absent from /root/reference):
//   p3 TwoAdicFriPcs::commit -> Radix2DitParallel::coset_lde_batch(evals, log_blowup, shift = g)
//   followed by .bit_reverse_rows()   [UPSTREAM-RECALL, Plonky3 @ a0b92870]
// i.e. for an N x w matrix of evaluations over H = <w_N> (natural order) produce the (N << b) x w
// matrix whose row bitrev(j) holds the column polynomials evaluated at g * w_{N<<b}^j, g = 31.
//
// Layout decision: everything stays row-major.  A butterfly couples two *rows*; the w columns of a
// row are independent and contiguous, so lanes run along a row (coalesced w*4-byte segments) and no
// transpose is ever needed between trace generation (row-major), LDE and Merkle leaf hashing
// (row-major rows).  Each pass stages a tile of 2^LOG_R rows x C columns in LDS, runs LOG_R
// decimation-in-frequency stages there, and writes the tile back: 3 passes for N = 2^20.
//
// LDE with blow-up 2^b is done as 2^b size-N transforms: block q of the bit-reversed output equals
// DIF_N(c_i * s_q^i) with s_q = g * w_{N<<b}^{bitrev_b(q)} (the first b stages of the size-(N<<b)
// DIF on zero-padded coefficients are trivial), so zero padding is never materialised.
//
// HBM traffic (algorithmic): iNTT 3 passes r+w over N*w*4 B, forward 2^b * 3 passes
// r+w over N*w*4 B.
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "babybear.h"
#include "commit.h"
#include "ctx.h"

#ifndef LURK_NTT_WAVES
#define LURK_NTT_WAVES 5
#endif

namespace lurkhip {

namespace {


__host__ __device__ inline uint32_t bitrev32(uint32_t x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

// tw[i] = root^i for i < count (Montgomery), root given in Montgomery form
__global__ void k_powers(uint32_t* __restrict__ out, uint32_t root_m, uint32_t scale_m, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = bb::mul(bb::pow(root_m, (uint32_t)i), scale_m);
}

struct PassArgs {
    // A launch transforms up to NTT_MAX_BATCH matrices of one shape: blockIdx.y picks the matrix, the x dimension is the
    // persistent grid of one matrix.  (The narrow matrices of a commitment -- quotient chunks, memory tables -- are a few
    // tens of microseconds per pass each: one launch per matrix leaves the launch boundaries and ramps to dominate.)
    const uint32_t* in[NTT_MAX_BATCH];    // N x w
    uint32_t* out[NTT_MAX_BATCH];         // N x w (may alias in when !bitrev_store and in-place is wanted)
    const uint32_t* tw;    // powers of the size-N root (or inverse root), N/2 entries, Montgomery
    const uint32_t* row_scale[NTT_MAX_BATCH];  // optional per-row multiplier applied on load (N entries) or nullptr
    int log_n;
    int w;
    int bit_lo;        // lowest row-index bit handled by this pass
    int col0;          // first column of this launch's first chunk
    int col_chunk;     // columns per tile (every tile of a launch has the same shape)
    int n_chunks;      // column chunks in this launch
    int in_canonical;  // convert on load
    int out_canonical; // convert on store
    int bitrev_store;  // store row r at bitrev(r, log_n)
    uint32_t magic_cv; // ceil(2^32 / column items): tid / Cv == umulhi(tid, magic)
    // Row grouping for narrow matrices: a tile takes 2^log_l ADJACENT rows (consecutive values of the low row bits)
    // for each of its 2^log_r strided rows, so that every global access is a contiguous run of (w << log_l) words
    // instead of w.  Then col_chunk = w << log_l (one chunk) and magic_w divides a tile column by w.
    int log_l;
    uint32_t magic_w;
    uint32_t n_tiles;  // (row tile, column chunk) pairs of the launch; a workgroup takes several
    uint32_t xcd_run;  // tiles per XCD when the tile count is a multiple of 8 (XCD-contiguous tile order), else 0
    // Chunk-tiled intermediates (round 3): the matrix between two passes of a transform is not the caller's row-major [N][w] but
    // [column chunk][N][NTT_TILE_PITCH words] -- a tile's row segment is then one aligned 128-byte line instead of 104..128
    // bytes at an arbitrary offset of a (w * 4)-byte row (78 columns: 312-byte rows; tools/ubench_fetch.hip: an unaligned
    // segment fetches two lines, and tools/lde_throughput.py: widths that are multiples of 32 run 20 % faster per element).
    int in_tiled, out_tiled;
    int chunk0;        // index of this launch's first column chunk in the matrix (the ragged last chunk has its own launch)
};
constexpr int NTT_TILE_PITCH = 32;

__device__ __forceinline__ int fast_div(uint32_t e, uint32_t magic, int c) {
    return c == 1 ? (int)e : (int)__umulhi(e, magic);
}

// x <- x + y, y <- (x - y) * tw  (decimation in frequency).  The difference is taken signed, in (-p, p), and goes
// through the 4-instruction signed Montgomery product; one correction brings the result back to [0, p).
__device__ __forceinline__ void dif_butterfly(uint32_t& x, uint32_t& y, uint32_t tw) {
    const uint32_t sum = bb::add(x, y);
    const uint32_t r = (uint32_t)bb::smul((int32_t)(x - y), (int32_t)tw);
    y = bb::umin(r, r + bb::P);
    x = sum;
}
__device__ __forceinline__ void dif_butterfly(uint2& x, uint2& y, uint32_t tw) {
    dif_butterfly(x.x, y.x, tw);
    dif_butterfly(x.y, y.y, tw);
}

// Row t of a tile column lives at element swz(t) of the column: t with its low five bits XORed by a linear function of bits
// 5..7.  A column is walked with strides of 2^S_BOT elements by the lanes of a wave (stage groups below), and 32 elements of
// 8 bytes are one sweep of the 64 LDS banks: unswizzled, the 16-element runs of the second group of a 1024-row tile start 128
// elements apart (the same banks: 4-way conflicts), the third and fourth groups collide 8-way -- the PMC counted 2.5 bank-conflict
// cycles per active LDS cycle, a fifth of the pass.  With A = {bit5 -> 0b00101, bit6 -> 0b01010, bit7 -> 0b10100} the low five
// bits are a bijection of every group's 32-lane index set ({0-4}, {0-3,7}, {0,1,4,5,6}, {2-6} for 2^10 rows; {0-5}, {0-2,6,7},
// {3-7} for 2^9), and the map is GF(2)-linear, so an item's rows t0 ^ (b << S_BOT) sit at swz(t0) ^ (a compile-time constant).
// Tiles of 2^10 rows keep the plain layout: their kernel stages sixteen rows per thread at 118-128 VGPRs (the cap of a
// 1024-thread workgroup), and the XOR-ed addresses -- no longer compile-time offsets from one base register -- pushed 60-77
// registers into scratch there (LDE of 2^20 x 78: 1.73 -> 2.30 ms although the bank-conflict cycles fell by 72 %).
#ifndef LURK_NTT_SWZ_MAX_LOG_R
#define LURK_NTT_SWZ_MAX_LOG_R 9
#endif
// row slots per column of a 1024-row tile (A/B: 32 = sixteen column items x 32 slots = 512 threads of up to 256 VGPRs)
#ifndef LURK_NTT_TALL_LOG_SLOTS
#define LURK_NTT_TALL_LOG_SLOTS 6
#endif
#define LURK_NTT_TALL_SLOTS (1 << LURK_NTT_TALL_LOG_SLOTS)
#ifndef LURK_NTT_WIDE_SCALAR
#define LURK_NTT_WIDE_SCALAR 0
#endif
// -DLURK_NTT_R9_TWO_PER_CU=1 (round 3, measured and rejected): two workgroups per CU for the 2^9-row tiles -- sixteen rows per
// thread like the 2^10-row tiles, i.e. 32 row slots x 16 column items = 512 threads and 66 KiB of LDS per workgroup, registers
// capped at the 128 of four waves per SIMD: two tiles of a CU in different phases of the load / stages / store chain instead of
// one 1024-thread workgroup using half the CU's LDS.  2^18 x 114: 0.857 against 0.527 ms, lde 12.9 against 10.4 ms per step (a
// column's 32 row slots are half a wave: twice the stage-group trips per thread, half the lanes per LDS access run).
#ifndef LURK_NTT_R9_TWO_PER_CU
#define LURK_NTT_R9_TWO_PER_CU 0
#endif
__host__ __device__ constexpr int ntt_rows_per_thread(int log_r) {  // of a column-pair item (the kernel's U)
    return log_r > 9 ? (1 << log_r) / LURK_NTT_TALL_SLOTS : (LURK_NTT_R9_TWO_PER_CU && log_r == 9 ? 16 : ((1 << log_r) < 8 ? (1 << log_r) : 8));
}
__host__ __device__ constexpr int ntt_log_rows_per_thread(int log_r) {
    return log_r > 9 ? log_r - LURK_NTT_TALL_LOG_SLOTS : (LURK_NTT_R9_TWO_PER_CU && log_r == 9 ? 4 : (log_r < 3 ? log_r : 3));
}
__host__ __device__ constexpr int ntt_max_threads(int log_r) {  // the kernels' launch bounds
    return log_r > 9 ? 16 * LURK_NTT_TALL_SLOTS : (LURK_NTT_R9_TWO_PER_CU && log_r == 9 ? 512 : 1024);
}
template <int LOG_R>
__host__ __device__ constexpr int swz(int t) {
    if (LOG_R > LURK_NTT_SWZ_MAX_LOG_R) return t;
    const int x = (t >> 5) & 7;
    return t ^ x ^ (x << 2);
}

// G consecutive DIF stages S_TOP .. S_TOP-G+1 of one column (T = one or two matrix columns) of the LDS tile.  The tile is
// stored column-major, col[t] = tile row t, so the 2^G rows t0 | b << S_BOT of an item sit at compile-time offsets from
// one address, and so do the item's twiddles tw[(1 << st) + t_lo]: no per-element index arithmetic.  The thread's slot
// walks the items q = slot, slot + slots, ... of its own column: no division either.
template <int LOG_R, int S_TOP, int G, int SLOTS, class T>
__device__ __forceinline__ void stage_group(T* __restrict__ col, const uint32_t* __restrict__ tw_l, int slot) {
    constexpr int M = 1 << G;
    constexpr int S_BOT = S_TOP - G + 1;
    constexpr int ITEMS = (1 << LOG_R) >> G;
    static_assert(ITEMS % SLOTS == 0, "every slot takes the same number of items");
#pragma unroll
    for (int q = slot; q < ITEMS; q += SLOTS) {
        const int low = q & ((1 << S_BOT) - 1);
        const int t0 = ((q >> S_BOT) << (S_TOP + 1)) | low;
        const int s0 = swz<LOG_R>(t0);
        const uint32_t* __restrict__ twp = tw_l + low;
        T x[M];
#pragma unroll
        for (int b = 0; b < M; b++) x[b] = col[s0 ^ swz<LOG_R>(b << S_BOT)];
#pragma unroll
        for (int g = G - 1; g >= 0; g--) {
            uint32_t tw[1 << (G - 1)];
#pragma unroll
            for (int j = 0; j < (1 << g); j++) tw[j] = twp[(1 << (S_BOT + g)) + (j << S_BOT)];
#pragma unroll
            for (int b = 0; b < M; b++) {
                if (b & (1 << g)) continue;
                dif_butterfly(x[b], x[b | (1 << g)], tw[b & ((1 << g) - 1)]);
            }
        }
#pragma unroll
        for (int b = 0; b < M; b++) col[s0 ^ swz<LOG_R>(b << S_BOT)] = x[b];
    }
}

// stage groups of a pass, top stage first: radix 8 while at least three stages remain (four are split 2 + 2)
// During the stages a column belongs to one wave (all its row slots are lanes of that wave), and a wave's LDS operations
// execute in order: between stage groups only the compiler has to be kept from reordering them -- no workgroup barrier.
// `after_first` runs once, after the first group -- the radix-8 one, which needs the most registers: the prefetch of the next
// tile is issued there, so its staging registers are not live while sixteen data registers and seven twiddles are.
template <int LOG_R, int S_TOP, int SLOTS, class T, class F>
__device__ __forceinline__ void run_stages(T* __restrict__ col, const uint32_t* __restrict__ tw_l, int slot, bool active, F&& after_first) {
    if constexpr (S_TOP < 0) {
        if constexpr (LOG_R == 0) after_first();
    } else {
        constexpr int REM = S_TOP + 1;
        constexpr int G = REM == 4 ? 2 : (REM >= 3 ? 3 : REM);
        if (active) stage_group<LOG_R, S_TOP, G, SLOTS, T>(col, tw_l, slot);
        if constexpr (S_TOP == LOG_R - 1) after_first();
        if constexpr (SLOTS <= 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
        run_stages<LOG_R, S_TOP - G, SLOTS, T>(col, tw_l, slot, active, after_first);
    }
}

__device__ __forceinline__ uint32_t to_monty_elem(uint32_t v) { return bb::to_monty(v); }
__device__ __forceinline__ uint2 to_monty_elem(uint2 v) { return make_uint2(bb::to_monty(v.x), bb::to_monty(v.y)); }
__device__ __forceinline__ uint32_t from_monty_elem(uint32_t v) { return bb::from_monty(v); }
__device__ __forceinline__ uint2 from_monty_elem(uint2 v) { return make_uint2(bb::from_monty(v.x), bb::from_monty(v.y)); }
__device__ __forceinline__ uint32_t scale_elem(uint32_t v, uint32_t s) { return bb::mul(v, s); }
__device__ __forceinline__ uint2 scale_elem(uint2 v, uint32_t s) { return make_uint2(bb::mul(v.x, s), bb::mul(v.y, s)); }

// One pass: tile = rows { hi << (bit_lo+LOG_R) | t << bit_lo | lo : t < 2^LOG_R } x col_chunk columns.
// LDS: the tile column-major with one element of padding per column ([Cv][R + 1] elements of T: consecutive lanes hold
// consecutive columns, the odd column stride keeps them on distinct banks), then the pass's twiddles
// tw_lds[l][(1 << s) + t_lo] for stage s.
//
// Workgroups are persistent and software-pipelined: while the stages of one tile run out of LDS, the rows (and twiddles) of
// the workgroup's next tile are already in flight into registers, so a workgroup overlaps its own HBM latency with its own
// butterflies instead of relying on neighbours being in a different phase (they start in lockstep and stay there).
// BIG: matrices of 4 GiB and more take 64-bit byte offsets; the others address rows as base (SGPR pair) + 32-bit offset.
// SCALE: a per-row multiplier (the coset shift powers) is applied on load.
template <int LOG_R, class T, bool BIG, bool SCALE, int TWN>
__device__ __forceinline__ void ntt_pass_body(const PassArgs& a) {
    using boff_t = typename std::conditional<BIG, size_t, uint32_t>::type;
    constexpr int R = 1 << LOG_R;
    constexpr int RP = R + 1;
    constexpr int EW = (int)(sizeof(T) / 4);  // matrix columns per element
    // rows a thread stages per tile: 8, and 16 for the 1024-row tiles so that the row slots of a column stay the 64 lanes of
    // one wave (the stage groups rely on it: wave-local ordering instead of workgroup barriers)
    // -DLURK_NTT_WIDE_SCALAR=1 (round 3, measured and rejected): one-column items (odd widths: rows are not 8-byte aligned, no
    // column pairs) stage twice the rows per thread, i.e. the same bytes per thread as a pair item, so that a tile is 32 columns
    // wide either way -- the 107- and 53-column chips take 7 and 4 tiles per row block where 4 and 2 would do.  The 32 staged rows
    // plus 32 row scales put 144 .. 416 bytes of the 128-VGPR kernels into scratch: lde 11.2 -> 11.9 ms per fib-mix step.
    constexpr int U_PAIR = ntt_rows_per_thread(LOG_R);
    constexpr int U = (LURK_NTT_WIDE_SCALAR && EW == 1 && LOG_R >= 4) ? 2 * U_PAIR : U_PAIR;
    constexpr int SLOTS = R / U;              // row slots per workgroup: thread = (slot, column item), slot < SLOTS
    constexpr int LOG_U_PAIR = ntt_log_rows_per_thread(LOG_R);
    constexpr int LOG_U = (LURK_NTT_WIDE_SCALAR && EW == 1 && LOG_R >= 4) ? LOG_U_PAIR + 1 : LOG_U_PAIR;
    constexpr int LOG_SLOTS = LOG_R - LOG_U;
    static_assert(SLOTS <= 64, "a column's row slots are lanes of one wave");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    T* tile = reinterpret_cast<T*>(smem);
    const int Cv = a.col_chunk / EW;
    uint32_t* tw_lds = smem + (size_t)Cv * RP * EW;
    const uint32_t lo_bits = (uint32_t)(a.bit_lo - a.log_l);
    // thread = (slot, cv): column item cv of the tile for the whole pass, tile rows / items slot, slot + SLOTS, ...
    // Threads past the last slot shadow a real thread's loads (no branch around the loads) and write nothing.
    const int slot_raw = fast_div(threadIdx.x, a.magic_cv, Cv), cv = (int)threadIdx.x - slot_raw * Cv;
    const bool active = slot_raw < SLOTS;
    const int slot = active ? slot_raw : SLOTS - 1;
    const int tc = cv * EW;                                     // first tile column of the item
    const int l_of = a.log_l ? fast_div((uint32_t)tc, a.magic_w, a.w) : 0;  // which of the L adjacent rows
    const int col_in_chunk = tc - l_of * a.w;
    T* __restrict__ my_col = tile + cv * RP;
    // the stages' map: slot fastest, so the SLOTS row slots of a column item are lanes of one wave
    const int st_cv = (int)threadIdx.x >> LOG_SLOTS, st_slot = (int)threadIdx.x & (SLOTS - 1);
    const bool st_active = st_cv < Cv;
    const int st_cvc = st_active ? st_cv : 0;
    const int st_l = a.log_l ? fast_div((uint32_t)(st_cvc * EW), a.magic_w, a.w) : 0;
    T* __restrict__ st_col = tile + st_cvc * RP;
    const int n_tw = R << a.log_l;

    // Tiles of this workgroup.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and tiles with
    // consecutive ids hold adjacent matrix rows (w * 4 bytes, not a multiple of the 128-byte line): every XCD takes a
    // contiguous run of tiles and walks it with all its workgroups abreast, so shared lines meet in one L2.
    const uint32_t wg = a.xcd_run ? (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t wgs = a.xcd_run ? (gridDim.x >> 3) : gridDim.x;
    const uint32_t run = a.xcd_run ? a.xcd_run : a.n_tiles;
    const uint32_t run0 = a.xcd_run ? (blockIdx.x & 7u) * a.xcd_run : 0u;
    if (wg >= run) return;
    const uint32_t my_tiles = (run - wg + wgs - 1) / wgs;
    // word offset of this thread's column inside a row (row-major), or inside the chunk's [N][pitch] slab plus the slab's start
    // (tiled): the element of matrix row `row` is at (row * stride + off) words
    const boff_t in_stride = a.in_tiled ? (boff_t)NTT_TILE_PITCH : (boff_t)a.w;
    const boff_t out_stride = a.out_tiled ? (boff_t)NTT_TILE_PITCH : (boff_t)a.w;
    auto locate = [&](uint32_t it, uint32_t& row0, boff_t& off_in, boff_t& off_out, uint32_t& lo) {
        const uint32_t bid = run0 + wg + it * wgs;
        const uint32_t tile_id = bid / (uint32_t)a.n_chunks;
        const int chunk = (int)(bid - tile_id * (uint32_t)a.n_chunks);
        lo = (tile_id & ((1u << lo_bits) - 1u)) << a.log_l;  // first of the tile's L adjacent low-bit values
        const uint32_t hi = tile_id >> lo_bits;
        row0 = ((hi << (a.bit_lo + LOG_R)) | lo) + (uint32_t)l_of;
        const boff_t col = (boff_t)(a.col0 + chunk * a.col_chunk + col_in_chunk);
        const boff_t slab = (((boff_t)(a.chunk0 + chunk)) << a.log_n) * (boff_t)NTT_TILE_PITCH + (boff_t)col_in_chunk;
        off_in = a.in_tiled ? slab : col;
        off_out = a.out_tiled ? slab : col;
    };
    // this thread's twiddle slots (TWN of them: 1, or 4 when grouped rows multiply the table): entry idx = [l][k] of the
    // table, k = (1 << s) + t_lo; slots past the table (and k = 0) read entry 0 and are not written
    auto tw_slot = [&](int i, uint32_t lo, size_t& src_idx) -> bool {
        const int idx = (int)threadIdx.x + i * (int)blockDim.x;
        const int l = idx >> LOG_R, k = idx & (R - 1);
        const bool ok = idx < n_tw && k != 0;
        const int s_ = 31 - __clz(k | 1);
        const uint32_t j = ((((uint32_t)k - (1u << s_)) << a.bit_lo) | (lo + (uint32_t)l));
        src_idx = ok ? ((size_t)j << (a.log_n - a.bit_lo - s_ - 1)) : (size_t)0;
        return ok;
    };
    T v[U];
    uint32_t sc[U], twv[TWN];
    const char* __restrict__ src = reinterpret_cast<const char*>(a.in[blockIdx.y]);
    char* __restrict__ dst = reinterpret_cast<char*>(a.out[blockIdx.y]);
    const uint32_t* __restrict__ row_scale = a.row_scale[blockIdx.y];
    auto fetch = [&](uint32_t row0, boff_t off_in, uint32_t lo) {
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int t = slot + (k << LOG_SLOTS);
            const uint32_t row = row0 | ((uint32_t)t << a.bit_lo);
            v[k] = *reinterpret_cast<const T*>(src + ((boff_t)row * in_stride + off_in) * 4);
        }
        // twiddles: w_{2h}^j, h = 2^(bit_lo+s), j = (t_lo << bit_lo) | (lo + l)
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            size_t si;
            tw_slot(i, lo, si);
            twv[i] = a.tw[si];
        }
    };

    // The row scales of a tile are requested when the tile is about to enter LDS, not with its rows one tile ahead (round 3):
    // prefetched, their U registers were live through the stage groups of the previous tile, on top of the radix-8 group's sixteen
    // data registers and seven twiddles, and pushed the 128-VGPR kernels 36 .. 124 bytes into scratch -- the scaled pass (the first
    // of every forward transform) took 345 us for 2^20 x 78 against 234 us for the same access pattern unscaled, its PMC fetch
    // 1.92 x the matrix against 1.51 x (the spills' traffic).  The table is L2-resident: one exposed L2 latency per tile.
    auto fetch_scales = [&](uint32_t row0) {
        if constexpr (SCALE) {
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int t = slot + (k << LOG_SLOTS);
                sc[k] = row_scale[row0 | ((uint32_t)t << a.bit_lo)];
            }
        }
    };
    uint32_t row0, lo;
    boff_t off_in, off_out;
    locate(0, row0, off_in, off_out, lo);
    fetch(row0, off_in, lo);
    for (uint32_t it = 0; it < my_tiles; it++) {
        fetch_scales(row0);
        // staged registers -> LDS
        if (active) {
#pragma unroll
            for (int k = 0; k < U; k++) {
                T x = v[k];
                if (a.in_canonical) x = to_monty_elem(x);
                if constexpr (SCALE) x = scale_elem(x, sc[k]);
                my_col[swz<LOG_R>(slot + (k << LOG_SLOTS))] = x;
            }
        }
#pragma unroll
        for (int i = 0; i < TWN; i++) {
            const int idx = (int)threadIdx.x + i * (int)blockDim.x;
            if (idx < n_tw && (idx & (R - 1)) != 0) tw_lds[idx] = twv[i];
        }
        const uint32_t cur_row0 = row0;
        const boff_t cur_off_out = off_out;
        __syncthreads();
        // the next tile's rows fly while this one's remaining stages run (the last iteration re-reads its own tile: no branch)
        run_stages<LOG_R, LOG_R - 1, SLOTS, T>(st_col, tw_lds + (st_l << LOG_R), st_slot, st_active, [&]() {
            locate(it + 1 < my_tiles ? it + 1 : it, row0, off_in, off_out, lo);
            fetch(row0, off_in, lo);
        });
        __syncthreads();  // the stages ran under the column-per-wave map, the write-back uses the row-contiguous one
        if (active) {
            constexpr int UB = U < 4 ? U : 4;  // rows per batch of the write-back (LDS reads first, then their global stores)
#pragma unroll
            for (int k0 = 0; k0 < U; k0 += UB) {
                T o[UB];
#pragma unroll
                for (int k = 0; k < UB; k++) o[k] = my_col[swz<LOG_R>(slot + ((k0 + k) << LOG_SLOTS))];
#pragma unroll
                for (int k = 0; k < UB; k++) {
                    const int t = slot + ((k0 + k) << LOG_SLOTS);
                    uint32_t row = cur_row0 | ((uint32_t)t << a.bit_lo);
                    if (a.bitrev_store) row = bitrev32(row, a.log_n);
                    T x = o[k];
                    if (a.out_canonical) x = from_monty_elem(x);
                    *reinterpret_cast<T*>(dst + ((boff_t)row * out_stride + cur_off_out) * 4) = x;
                }
            }
        }
        __syncthreads();  // the tile is free for the next one
    }
}

// ---------------------------------------------------------------- fused pass of the LDE (round 3)
// The LAST pass of the inverse transform and the FIRST pass of every coset's forward transform in one kernel, the coefficients
// never leaving the chip: a tile of the inverse's last pass (2^LOG_R consecutive storage rows s = hi << LOG_R | t) holds, after
// its stages, the coefficients k = bitrev_r(t) << (log_n - LOG_R) | bitrev(hi) -- every value of the top LOG_R bits of k for one
// value of the rest, i.e. exactly a tile of a forward first pass (rows t' << bit_lo_f | lo with lo = bitrev(hi)), its rows in
// bit-reversed order.  Per tile: rows -> LDS, inverse stages, tile back into the staging registers, then per coset: registers x
// that coset's scale (s_q^k / N) -> LDS at the bit-reversed local row, forward stages, store as the forward transform's
// first-pass output.  Of the twelve matrix transfers of an LDE (three transforms x two passes x read + write) three go: the
// inverse's coefficient write and both cosets' coefficient reads; the coefficient buffer itself is never written.
struct FusedArgs {
    const uint32_t* in[NTT_MAX_BATCH];       // output of the inverse transform's earlier passes (N x w, bit-reversed order so far)
    uint32_t* out[NTT_MAX_BATCH][2];         // per coset: first-pass output of its forward transform (N x w)
    const uint32_t* scale[NTT_MAX_BATCH][2];  // per coset: s_q^k / N, k < N
    const uint32_t *tw_inv, *tw_fwd;          // N/2 powers of the inverse / forward root
    int log_n, w, n_cosets;
    int col0, col_chunk, n_chunks;
    uint32_t magic_cv, n_tiles, xcd_run;
};

template <int LOG_R, class T, bool BIG>
__device__ __forceinline__ void ntt_fused_body(const FusedArgs& a) {
    using boff_t = typename std::conditional<BIG, size_t, uint32_t>::type;
    constexpr int R = 1 << LOG_R;
    constexpr int RP = R + 1;
    constexpr int EW = (int)(sizeof(T) / 4);
    constexpr int U = ntt_rows_per_thread(LOG_R);
    constexpr int SLOTS = R / U;
    constexpr int LOG_U = ntt_log_rows_per_thread(LOG_R);
    constexpr int LOG_SLOTS = LOG_R - LOG_U;
    static_assert(SLOTS <= 64, "a column's row slots are lanes of one wave");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    T* tile = reinterpret_cast<T*>(smem);
    const int Cv = a.col_chunk / EW;
    uint32_t* tw_i = smem + (size_t)Cv * RP * EW;  // inverse twiddles of the pass: the same for every tile (bit_lo = 0)
    uint32_t* tw_f = tw_i + R;                     // forward twiddles of the tile
    const int hi_bits = a.log_n - LOG_R;           // = bit_lo of the forward first pass
    const int slot_raw = fast_div(threadIdx.x, a.magic_cv, Cv), cv = (int)threadIdx.x - slot_raw * Cv;
    const bool active = slot_raw < SLOTS;
    const int slot = active ? slot_raw : SLOTS - 1;
    T* __restrict__ my_col = tile + cv * RP;
    const int st_cv = (int)threadIdx.x >> LOG_SLOTS, st_slot = (int)threadIdx.x & (SLOTS - 1);
    const bool st_active = st_cv < Cv;
    T* __restrict__ st_col = tile + (st_active ? st_cv : 0) * RP;

    const uint32_t wg = a.xcd_run ? (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t wgs = a.xcd_run ? (gridDim.x >> 3) : gridDim.x;
    const uint32_t run = a.xcd_run ? a.xcd_run : a.n_tiles;
    const uint32_t run0 = a.xcd_run ? (blockIdx.x & 7u) * a.xcd_run : 0u;
    if (wg >= run) return;
    const uint32_t my_tiles = (run - wg + wgs - 1) / wgs;
    // one twiddle per thread (the launch has at least R threads): entry k = (1 << s) + t_lo of stage s
    const int twk = (int)threadIdx.x;
    const bool tw_ok = twk < R && twk != 0;
    const int tw_s = 31 - __clz(twk | 1);
    const uint32_t tw_tlo = (uint32_t)twk - (1u << tw_s);
    if (tw_ok) tw_i[twk] = a.tw_inv[(size_t)tw_tlo << (a.log_n - tw_s - 1)];  // w_{2h}^j, h = 2^s, j = t_lo
    auto locate = [&](uint32_t it, uint32_t& hi, boff_t& col) {
        const uint32_t bid = run0 + wg + it * wgs;
        hi = bid / (uint32_t)a.n_chunks;
        const int chunk = (int)(bid - hi * (uint32_t)a.n_chunks);
        col = (boff_t)(a.col0 + chunk * a.col_chunk + cv * EW);
    };
    const char* __restrict__ src = reinterpret_cast<const char*>(a.in[blockIdx.y]);
    T v[U];
    auto fetch = [&](uint32_t hi, boff_t col) {
#pragma unroll
        for (int k = 0; k < U; k++) {
            const uint32_t row = (hi << LOG_R) | (uint32_t)(slot + (k << LOG_SLOTS));
            v[k] = *reinterpret_cast<const T*>(src + ((boff_t)row * (boff_t)a.w + col) * 4);
        }
    };
    uint32_t hi;
    boff_t col;
    locate(0, hi, col);
    fetch(hi, col);
    for (uint32_t it = 0; it < my_tiles; it++) {
        const uint32_t lo = hi_bits ? bitrev32(hi, hi_bits) : 0u;  // the forward tile's fixed low bits
        const boff_t cur_col = col;
        // forward twiddles of this tile: w_{2h}^j, h = 2^(hi_bits + s), j = t_lo << hi_bits | lo
        uint32_t twf = 0;
        if (tw_ok) twf = a.tw_fwd[(size_t)((tw_tlo << hi_bits) | lo) << (a.log_n - hi_bits - tw_s - 1)];
        if (active) {
#pragma unroll
            for (int k = 0; k < U; k++) my_col[swz<LOG_R>(slot + (k << LOG_SLOTS))] = v[k];
        }
        __syncthreads();
        run_stages<LOG_R, LOG_R - 1, SLOTS, T>(st_col, tw_i, st_slot, st_active, [&]() {});
        __syncthreads();
        // the tile (coefficients, local row t = coefficient top bits bitrev_r(t)) back into the staging registers
        if (active) {
#pragma unroll
            for (int k = 0; k < U; k++) v[k] = my_col[swz<LOG_R>(slot + (k << LOG_SLOTS))];
        }
        if (tw_ok) tw_f[twk] = twf;
        __syncthreads();
        for (int q = 0; q < a.n_cosets; q++) {
            const uint32_t* __restrict__ scale = a.scale[blockIdx.y][q];
            if (active) {
#pragma unroll
                for (int k = 0; k < U; k++) {
                    const uint32_t tp = bitrev32((uint32_t)(slot + (k << LOG_SLOTS)), LOG_R);  // forward local row
                    my_col[swz<LOG_R>((int)tp)] = scale_elem(v[k], scale[(tp << hi_bits) | lo]);
                }
            }
            const bool last_q = q + 1 == a.n_cosets;
            if (last_q) {  // the staging registers are free: the next tile's rows fly under this coset's stages
                locate(it + 1 < my_tiles ? it + 1 : it, hi, col);
                fetch(hi, col);
            }
            __syncthreads();
            run_stages<LOG_R, LOG_R - 1, SLOTS, T>(st_col, tw_f, st_slot, st_active, [&]() {});
            __syncthreads();
            if (active) {
                char* __restrict__ dst = reinterpret_cast<char*>(a.out[blockIdx.y][q]);
                constexpr int UB = U < 4 ? U : 4;
#pragma unroll
                for (int k0 = 0; k0 < U; k0 += UB) {
                    T o[UB];
#pragma unroll
                    for (int k = 0; k < UB; k++) o[k] = my_col[swz<LOG_R>(slot + ((k0 + k) << LOG_SLOTS))];
#pragma unroll
                    for (int k = 0; k < UB; k++) {
                        const uint32_t row = ((uint32_t)(slot + ((k0 + k) << LOG_SLOTS)) << hi_bits) | lo;
                        *reinterpret_cast<T*>(dst + ((boff_t)row * (boff_t)a.w + cur_col) * 4) = o[k];
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <int LOG_R, class T, bool BIG>
__global__ __launch_bounds__(ntt_max_threads(LOG_R)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ntt_fused(FusedArgs a) {
    ntt_fused_body<LOG_R, T, BIG>(a);
}

// tiles of up to 128 rows: two or three workgroups per CU, registers capped for five waves per SIMD
template <int LOG_R, class T, bool BIG, bool SCALE, int TWN>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(LURK_NTT_WAVES, 8))) void k_ntt_pass(PassArgs a) {
    ntt_pass_body<LOG_R, T, BIG, SCALE, TWN>(a);
}
// tiles of 256 .. 1024 rows: one workgroup of up to 16 waves per CU (its LDS tile is most of the CU's 160 KiB), so each wave may
// use the 128 VGPRs of a 4-waves-per-SIMD kernel -- the 1024-row tiles stage sixteen rows per thread
template <int LOG_R, class T, bool BIG, bool SCALE, int TWN>
__global__ __launch_bounds__(ntt_max_threads(LOG_R)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ntt_pass_tall(PassArgs a) {
    ntt_pass_body<LOG_R, T, BIG, SCALE, TWN>(a);
}

template <class T, bool BIG, bool SCALE, int TWN>
void launch_pass2(int log_r, dim3 blocks, int threads, size_t lds, hipStream_t stream, const PassArgs& a) {
    switch (log_r) {
#define LH_NTT_CASE(LR, KERNEL)                                                                                              \
    case LR: {                                                                                                               \
        auto kern = KERNEL<LR, T, BIG, SCALE, TWN>;                                                                          \
        if (lds > 64 * 1024) {                                                                                               \
            static bool big_lds_ok = false; /* per instantiation: tiles above 64 KiB need the opt-in once */               \
            if (!big_lds_ok) {                                                                                               \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
                big_lds_ok = true;                                                                                           \
            }                                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, blocks, dim3(threads), lds, stream, a);                                                     \
        break;                                                                                                               \
    }
        LH_NTT_CASE(0, k_ntt_pass) LH_NTT_CASE(1, k_ntt_pass) LH_NTT_CASE(2, k_ntt_pass) LH_NTT_CASE(3, k_ntt_pass)
        LH_NTT_CASE(4, k_ntt_pass) LH_NTT_CASE(5, k_ntt_pass) LH_NTT_CASE(6, k_ntt_pass) LH_NTT_CASE(7, k_ntt_pass)
        LH_NTT_CASE(8, k_ntt_pass_tall) LH_NTT_CASE(9, k_ntt_pass_tall) LH_NTT_CASE(10, k_ntt_pass_tall)
#undef LH_NTT_CASE
    }
}

static uint32_t magic_for(int c) { return c <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + c - 1) / c); }

// canonical primitive 2^27-th root of unity used by p3 BabyBear: 0x1a427a41 [UPSTREAM-RECALL];
// any generator of the 2-Sylow subgroup gives the same subgroup H, but the *order of rows* in the
// LDE depends on which generator is used, so it is a parameter pinned here in one place.
constexpr uint32_t TWO_ADIC_ROOT_27 = 0x1a427a41u;

uint32_t host_pow(uint32_t a_m, uint64_t e) {
    uint32_t r = bb::R1;
    while (e) {
        if (e & 1) r = bb::mul(r, a_m);
        a_m = bb::mul(a_m, a_m);
        e >>= 1;
    }
    return r;
}

}  // namespace

uint32_t two_adic_generator_monty(int bits) {
    // w_{2^bits} = root27^(2^(27-bits)).  BabyBear has no subgroup of order 2^28 and more: the callers bound their heights
    // (commit_impl, the verifier's decoders), and a request outside [0, 27] gets 0 -- not a root of anything -- instead of
    // silently the 2^27 root
    if (bits < 0 || bits > bb::TWO_ADICITY) return 0;
    uint32_t r = bb::to_monty(TWO_ADIC_ROOT_27);
    for (int i = bits; i < bb::TWO_ADICITY; i++) r = bb::mul(r, r);
    return r;
}

int32_t NttPlan::init(lurkhip_ctx* ctx, int log_n_) {
    log_n = log_n_;
    size_t half = log_n > 0 ? ((size_t)1 << (log_n - 1)) : 1;
    LH_HIP(ctx, hipMalloc(&tw_fwd, half * 4));
    LH_HIP(ctx, hipMalloc(&tw_inv, half * 4));
    uint32_t root = two_adic_generator_monty(log_n);
    uint32_t root_inv = host_pow(root, bb::P - 2);
    unsigned blocks = (unsigned)((half + 255) / 256);
    hipLaunchKernelGGL(k_powers, dim3(blocks), dim3(256), 0, ctx->stream, tw_fwd, root, bb::R1, half);
    hipLaunchKernelGGL(k_powers, dim3(blocks), dim3(256), 0, ctx->stream, tw_inv, root_inv, bb::R1, half);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

void NttPlan::destroy() {
    if (tw_fwd) (void)hipFree(tw_fwd);
    if (tw_inv) (void)hipFree(tw_inv);
    tw_fwd = tw_inv = nullptr;
}

int32_t fill_powers(lurkhip_ctx* ctx, uint32_t* out, uint32_t root_m, uint32_t scale_m, size_t count) {
    unsigned blocks = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(k_powers, dim3(blocks), dim3(256), 0, ctx->stream, out, root_m, scale_m, count);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

template <class T, bool BIG>
void launch_pass(int log_r, dim3 blocks, int threads, size_t lds, hipStream_t stream, const PassArgs& a) {
    const bool one = ((size_t)1 << (log_r + a.log_l)) <= (size_t)threads;  // one twiddle per thread covers the table
    if (a.row_scale[0] && one) launch_pass2<T, BIG, true, 1>(log_r, blocks, threads, lds, stream, a);
    else if (a.row_scale[0]) launch_pass2<T, BIG, true, 4>(log_r, blocks, threads, lds, stream, a);
    else if (one) launch_pass2<T, BIG, false, 1>(log_r, blocks, threads, lds, stream, a);
    else launch_pass2<T, BIG, false, 4>(log_r, blocks, threads, lds, stream, a);
}

// Pass schedule: split log_n stage bits (from the top) into chunks of at most max_log_r.
static void schedule(int log_n, int max_log_r, std::vector<std::pair<int, int>>& passes /* (bit_lo, log_r) */) {
    passes.clear();
    int remaining = log_n;
    int n_pass = (log_n + max_log_r - 1) / max_log_r;
    if (n_pass == 0) n_pass = 1;
    int base = log_n / n_pass, extra = log_n % n_pass;
    for (int p = 0; p < n_pass; p++) {
        int lr = base + (p < extra ? 1 : 0);
        remaining -= lr;
        passes.push_back({remaining, lr});
    }
}

// The column-chunk plan of a transform: tile height and chunk width, decided once per (log_n, w) so that every pass of a transform
// -- and the inverse / forward transforms that hand a chunk-tiled matrix to each other -- cut the columns the same way.
struct ChunkPlan {
    int max_log_r, col_chunk, n_pass;
    size_t lds_cap;
};
static ChunkPlan plan_chunks(int log_n, int w, bool aligned8) {
    auto items_of = [&](int cols) { return (aligned8 && cols % 2 == 0) ? cols / 2 : cols; };
    auto is_pair = [&](int cols) { return aligned8 && cols % 2 == 0; };
    auto rows_per_thread = [](int log_r, bool pair) {  // the kernel's U
        const int u = ntt_rows_per_thread(log_r);
        return (LURK_NTT_WIDE_SCALAR && !pair && log_r >= 4) ? 2 * u : u;
    };
    auto max_threads = [](int log_r) { return ntt_max_threads(log_r); };  // the kernels' launch bounds
    auto threads_of = [&](int log_r, int cols) { return std::max(1, (1 << log_r) / rows_per_thread(log_r, is_pair(cols))) * items_of(cols); };
    auto lds_bytes = [](int log_r, int cols, int log_l) {
        return ((size_t)cols * (((size_t)1 << log_r) + 1) + ((size_t)1 << (log_r + log_l))) * 4;
    };
    // Every pass reads and writes the whole matrix, so the number of passes is what the LDE costs in HBM traffic: tiles of up to
    // 2^10 rows (132 KiB of the CU's 160 KiB of LDS at 32 columns) take a 2^20-row transform in two passes instead of three.
    // LURKHIP_NTT_MAX_LOG_R caps the tile height (7 = the 64 KiB tiles of round 1; for A/B measurements).
    int cap_log_r = 10;
    if (const char* e = getenv("LURKHIP_NTT_MAX_LOG_R")) cap_log_r = std::max(1, std::min(10, atoi(e)));
    const int n_pass = std::max(1, (log_n + cap_log_r - 1) / cap_log_r);
    int max_log_r = std::max(1, (log_n + n_pass - 1) / n_pass);  // tallest tile of the schedule
    size_t lds_cap = max_log_r > 7 ? (size_t)140 * 1024 : (size_t)64 * 1024;
    if (const char* e = getenv("LURKHIP_NTT_LDS_CAP_KB")) lds_cap = (size_t)std::max(8, atoi(e)) * 1024;  // A/B hook: tile bytes per workgroup
    // column chunk: the widest even divisor of w that fits (no ragged chunk), else the widest even width that fits (the ragged
    // remainder gets its own launch); narrow matrices are one chunk
    auto fits = [&](int cols) { return lds_bytes(max_log_r, cols, 0) <= lds_cap && threads_of(max_log_r, cols) <= max_threads(max_log_r); };
    int col_chunk = w;
    if (!fits(w)) {
        int widest = 2;
        for (int c = std::min(w, 112); c >= 2; c--)
            if (c % 2 == 0 && fits(c)) {
                widest = c;
                break;
            }
        col_chunk = widest;
        for (int c = widest; c >= std::max(2, widest * 3 / 4); c--)
            if (c % 2 == 0 && w % c == 0) {
                col_chunk = c;
                break;
            }
    }
    while ((lds_bytes(max_log_r, col_chunk, 0) > lds_cap || threads_of(max_log_r, col_chunk) > max_threads(max_log_r)) && max_log_r > 1)
        max_log_r--;
    return ChunkPlan{max_log_r, col_chunk, std::max(1, (log_n + max_log_r - 1) / max_log_r), lds_cap};
}

size_t ntt_tiled_words(int log_n, int w) {
    // off by default: measured on the fib-mix step (round 3) the tiled intermediates are 5 % SLOWER (lde 11.8 against 11.2 ms)
    // -- see DESIGN.md 3.3; LURKHIP_NTT_TILED=1 turns them on for A/B measurements
    static const bool enabled = getenv("LURKHIP_NTT_TILED") != nullptr && atoi(getenv("LURKHIP_NTT_TILED")) != 0;
    if (!enabled || w % 2 != 0) return 0;
    const ChunkPlan cp = plan_chunks(log_n, w, true);
    const int n_chunks = (w + cp.col_chunk - 1) / cp.col_chunk;
    if (n_chunks < 2 || cp.col_chunk > NTT_TILE_PITCH || cp.n_pass < 2) return 0;
    return ((size_t)n_chunks << log_n) * NTT_TILE_PITCH;
}

// Full size-N DIF transform of an N x w row-major matrix.
//   src -> dst, natural-order input; output bit-reversed, or natural when bitrev_store (the
//   permutation is fused into the last pass's store).  `scratch` (N x w) is needed when more than one
//   pass runs and the last pass scatters (it must not alias dst).  row_scale multiplies row i on load.
int32_t ntt_dif(lurkhip_ctx* ctx, const NttPlan& plan, bool inverse, const uint32_t* src, uint32_t* dst,
                uint32_t* scratch, int w, const uint32_t* row_scale, bool in_canonical, bool out_canonical,
                bool bitrev_store) {
    NttBatch b{};
    b.n = 1;
    b.src[0] = src;
    b.dst[0] = dst;
    b.scratch[0] = scratch;
    b.row_scale[0] = row_scale;
    return ntt_dif_batch(ctx, plan, inverse, b, w, in_canonical, out_canonical, bitrev_store);
}

// The same transform for b.n matrices of one shape, one launch per pass.  Either every matrix has a row_scale or none.
int32_t ntt_dif_batch(lurkhip_ctx* ctx, const NttPlan& plan, bool inverse, const NttBatch& b, int w, bool in_canonical,
                      bool out_canonical, bool bitrev_store) {
    LH_ARG(ctx, b.n >= 1 && b.n <= NTT_MAX_BATCH, "NTT batch size");
    const int log_n = plan.log_n;
    // two adjacent columns per lane (8-byte accesses) when rows and chunks start on 8-byte boundaries
    uintptr_t all_ptrs = 0;
    for (int m = 0; m < b.n; m++) {
        all_ptrs |= (uintptr_t)b.src[m] | (uintptr_t)b.dst[m] | (uintptr_t)b.scratch[m];
        LH_ARG(ctx, (b.row_scale[m] != nullptr) == (b.row_scale[0] != nullptr), "NTT batch mixes scaled and unscaled matrices");
    }
    const bool aligned8 = (all_ptrs & 7u) == 0 && w % 2 == 0;
    auto items_of = [&](int cols) { return (aligned8 && cols % 2 == 0) ? cols / 2 : cols; };
    auto is_pair = [&](int cols) { return aligned8 && cols % 2 == 0; };
    auto rows_per_thread = [](int log_r, bool pair) {  // the kernel's U
        const int u = ntt_rows_per_thread(log_r);
        return (LURK_NTT_WIDE_SCALAR && !pair && log_r >= 4) ? 2 * u : u;
    };
    auto max_threads = [](int log_r) { return ntt_max_threads(log_r); };  // the kernels' launch bounds
    auto threads_of = [&](int log_r, int cols) { return std::max(1, (1 << log_r) / rows_per_thread(log_r, is_pair(cols))) * items_of(cols); };
    auto lds_bytes = [](int log_r, int cols, int log_l) {
        return ((size_t)cols * (((size_t)1 << log_r) + 1) + ((size_t)1 << (log_r + log_l))) * 4;
    };
    // Every pass reads and writes the whole matrix, so the number of passes is what the LDE costs in HBM traffic: tiles of up to
    // 2^10 rows (132 KiB of the CU's 160 KiB of LDS at 32 columns) take a 2^20-row transform in two passes instead of three.
    // LURKHIP_NTT_MAX_LOG_R caps the tile height (7 = the 64 KiB tiles of round 1; for A/B measurements).
    ChunkPlan cp = plan_chunks(log_n, w, aligned8);
    int max_log_r = cp.max_log_r;
    const int col_chunk = cp.col_chunk;
    const size_t lds_cap = cp.lds_cap;
    // The tile's twiddles are staged by at most four loads per thread, so a launch has at least 2^(log_r + log_l) / 4 threads:
    // a narrow (ragged) chunk of a tall tile gets idle threads -- they shadow a real thread's loads and write nothing.
    auto launch_threads = [&](int log_r, int cols, int log_l) {
        int need = (int)((((size_t)1 << (log_r + log_l)) + 3) / 4);
        // tall tiles: enough threads for ONE twiddle load each whenever the launch bounds allow it (the four-load variants of the
        // 128-VGPR kernels spill 12 .. 48 bytes; the 26-column chunks of a 78-column matrix launched 832 threads and took them)
        static const bool one_tw = getenv("LURKHIP_NTT_ONE_TW") == nullptr || atoi(getenv("LURKHIP_NTT_ONE_TW")) != 0;
        if (one_tw && log_r >= 8 && ((size_t)1 << (log_r + log_l)) <= (size_t)max_threads(log_r)) need = 1 << (log_r + log_l);
        return std::min(max_threads(log_r), (std::max(threads_of(log_r, cols), need) + 63) / 64 * 64);
    };
    std::vector<std::pair<int, int>> passes;
    schedule(log_n, max_log_r, passes);
    if (b.reversed_schedule) {  // the same pass sizes, smallest first (top bits first all the same: DIF order)
        std::vector<int> sizes;
        for (auto& pr : passes) sizes.push_back(pr.second);
        std::reverse(sizes.begin(), sizes.end());
        int remaining = log_n;
        for (size_t i = 0; i < passes.size(); i++) {
            remaining -= sizes[i];
            passes[i] = {remaining, sizes[i]};
        }
    }
    const size_t p_begin = b.skip_first_pass ? 1 : 0, p_end = passes.size() - (b.skip_last_pass ? 1 : 0);
    LH_ARG(ctx, p_begin <= p_end && !(b.skip_first_pass && b.skip_last_pass && passes.size() < 2), "NTT pass range");
    const int n_full = w / col_chunk, last_w = w % col_chunk;
    // chunk-tiled layouts: only for shapes ntt_tiled_words() admits, cut exactly as it assumes
    const size_t tiled_words = (b.src_tiled || b.dst_tiled || b.scratch_tiled) ? ntt_tiled_words(log_n, w) : 0;
    if (b.src_tiled || b.dst_tiled || b.scratch_tiled)
        LH_ARG(ctx, tiled_words != 0 && aligned8 && col_chunk <= NTT_TILE_PITCH, "this NTT shape has no chunk-tiled form");
    const bool inter_tiled = b.scratch_tiled && passes.size() >= 2;
    if (inter_tiled)
        for (int m = 0; m < b.n; m++) LH_ARG(ctx, b.scratch[m] != nullptr, "tiled intermediates need a scratch buffer");
    const uint32_t* cur_in[NTT_MAX_BATCH] = {};
    for (int m = 0; m < b.n; m++) cur_in[m] = b.src[m];
    for (size_t p = p_begin; p < p_end; p++) {
        bool last = p + 1 == p_end;
        PassArgs a{};
        for (int m = 0; m < b.n; m++) {
            uint32_t* cur_out;
            if (last) cur_out = b.dst[m];
            else if (inter_tiled) cur_out = b.scratch[m];   // chunk-tiled intermediates (in place for the middle passes)
            else if (bitrev_store) cur_out = b.scratch[m];  // keep dst free for the final scatter
            else cur_out = b.dst[m];                         // in-place chain inside dst
            a.in[m] = cur_in[m];
            a.out[m] = cur_out;
            a.row_scale[m] = p == 0 ? b.row_scale[m] : nullptr;
        }
        a.in_tiled = (p == 0 ? b.src_tiled : inter_tiled) ? 1 : 0;
        a.out_tiled = (last ? b.dst_tiled : inter_tiled) ? 1 : 0;
        a.tw = inverse ? plan.tw_inv : plan.tw_fwd;
        a.log_n = log_n;
        a.w = w;
        a.bit_lo = passes[p].first;
        const int log_r = passes[p].second;
        a.in_canonical = (p == 0 && in_canonical) ? 1 : 0;
        a.out_canonical = (last && out_canonical && !b.skip_last_pass) ? 1 : 0;
        a.bitrev_store = (last && bitrev_store && !b.skip_last_pass) ? 1 : 0;
        a.magic_w = magic_for(w);
        // narrow matrices: group adjacent rows in the strided passes so that global runs are >= ~256 bytes
        int log_l = 0;
        if (n_full == 1 && last_w == 0 && a.bit_lo > 0 && !a.bitrev_store) {
            while (log_l < 4 && log_l < a.bit_lo && (w << (log_l + 1)) <= 128 &&
                   lds_bytes(log_r, w << (log_l + 1), log_l + 1) <= lds_cap && threads_of(log_r, w << (log_l + 1)) <= max_threads(log_r) &&
                   ((size_t)1 << (log_r + log_l + 1)) <= (size_t)4 * launch_threads(log_r, w << (log_l + 1), log_l + 1))
                log_l++;
        }
        a.log_l = log_l;
        // one launch for the full chunks, one more for a ragged last chunk: every tile of a launch has the same shape
        for (int part = 0; part < 2; part++) {
            if (part == 0 && n_full == 0) continue;
            if (part == 1 && last_w == 0) continue;
            a.col0 = part == 0 ? 0 : n_full * col_chunk;
            a.chunk0 = part == 0 ? 0 : n_full;
            a.col_chunk = part == 0 ? (col_chunk << log_l) : last_w;
            a.n_chunks = part == 0 ? n_full : 1;
            const bool pair = aligned8 && a.col_chunk % 2 == 0 && a.col0 % 2 == 0;
            const int cv = a.col_chunk / (pair ? 2 : 1);
            const int slots = std::max(1, (1 << log_r) / rows_per_thread(log_r, pair));  // the kernel's SLOTS
            LH_ARG(ctx, slots * cv <= max_threads(log_r), "NTT tile shape");
            a.magic_cv = magic_for(cv);
            const int threads = launch_threads(log_r, a.col_chunk, log_l);
            const size_t tiles = ((size_t)1 << (log_n - log_r - log_l)) * a.n_chunks;
            LH_ARG(ctx, tiles <= 0x7fffffffu, "NTT grid too large");
            LH_ARG(ctx, ((size_t)1 << (log_r + log_l)) <= (size_t)4 * threads, "NTT twiddle staging");
            a.n_tiles = (uint32_t)tiles;
            a.xcd_run = (tiles % 8 == 0 && tiles >= 64) ? (uint32_t)(tiles / 8) : 0u;
            // persistent workgroups: as many per CU as its registers (5 waves per SIMD at <= 96 VGPRs) and LDS hold, all
            // resident at once; a multiple of 8 so that every XCD gets the same number
            const size_t lds = lds_bytes(log_r, a.col_chunk, log_l);
            const int per_cu = std::max(1, std::min(20 / (threads / 64), (int)((160 * 1024) / (lds + 256))));
            // (a batch shares the CUs: every matrix gets its part of the resident grid)
            size_t blocks = std::min<size_t>(tiles, std::max<size_t>(8, (size_t)per_cu * ctx->num_cus / b.n));
            if (a.xcd_run) blocks = blocks / 8 * 8;
            // matrices of 4 GiB and more need 64-bit offsets; LURKHIP_NTT_FORCE_64BIT (test hook) takes that path at any size
            const size_t widest_row = std::max<size_t>((size_t)w, tiled_words >> log_n);  // words per row of the larger layout in use
            const bool big = (widest_row << (log_n + 2)) >= ((size_t)1 << 32) || getenv("LURKHIP_NTT_FORCE_64BIT") != nullptr;
            const dim3 grid((unsigned)blocks, (unsigned)b.n);
            if (pair && big) launch_pass<uint2, true>(log_r, grid, threads, lds, ctx->stream, a);
            else if (pair) launch_pass<uint2, false>(log_r, grid, threads, lds, ctx->stream, a);
            else if (big) launch_pass<uint32_t, true>(log_r, grid, threads, lds, ctx->stream, a);
            else launch_pass<uint32_t, false>(log_r, grid, threads, lds, ctx->stream, a);
            LH_HIP(ctx, hipGetLastError());
        }
        for (int m = 0; m < b.n; m++) cur_in[m] = a.out[m];
    }
    return LURKHIP_OK;
}

template <class T, bool BIG>
static void launch_fused(int log_r, dim3 blocks, int threads, size_t lds, hipStream_t stream, const FusedArgs& a) {
    switch (log_r) {
#define LH_FUSED_CASE(LR)                                                                                                    \
    case LR: {                                                                                                               \
        auto kern = k_ntt_fused<LR, T, BIG>;                                                                                 \
        if (lds > 64 * 1024) {                                                                                               \
            static bool big_lds_ok = false;                                                                                  \
            if (!big_lds_ok) {                                                                                               \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
                big_lds_ok = true;                                                                                           \
            }                                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, blocks, dim3(threads), lds, stream, a);                                                     \
        break;                                                                                                               \
    }
        LH_FUSED_CASE(6) LH_FUSED_CASE(7) LH_FUSED_CASE(8) LH_FUSED_CASE(9) LH_FUSED_CASE(10)
#undef LH_FUSED_CASE
    }
}

// shapes the fused pass takes: at least two passes per transform, first-pass tiles of 2^6 .. 2^10 rows, even widths in more
// than one column chunk or at least a line wide (narrow matrices keep the row-grouping passes)
bool ntt_lde_fused_eligible(int log_n, int w) {
    // off by default (round 3, measured): correct -- the commitment and prover parity tests pass under LURKHIP_NTT_FUSED=1 -- and
    // slower: the tile's coefficients must survive the first coset's forward stages in the staging registers, on top of the
    // radix-8 group's sixteen data registers and seven twiddles, and the 128-VGPR kernels spill 164 .. 572 bytes per thread
    // (2^10-row tiles: 572): lde 13.65 against 10.46 ms per fib-mix step, same box.  With 512-thread workgroups (256 VGPRs,
    // -DLURK_NTT_TALL_LOG_SLOTS=5) nothing spills but the plain passes are 9 % slower already (11.5 ms).  DESIGN.md 3.3.
    static const bool enabled = getenv("LURKHIP_NTT_FUSED") != nullptr && atoi(getenv("LURKHIP_NTT_FUSED")) != 0;
    if (!enabled || w % 2 != 0 || w < 32) return false;
    const ChunkPlan cp = plan_chunks(log_n, w, true);
    return cp.n_pass >= 2 && cp.max_log_r >= 6 && cp.max_log_r <= 10;
}

int32_t ntt_lde_fused(lurkhip_ctx* ctx, const NttPlan& plan, int n, const uint32_t* const* inter, uint32_t* const (*out)[2],
                      const uint32_t* const (*scale)[2], int w, bool* done) {
    *done = false;
    const int log_n = plan.log_n;
    if (!ntt_lde_fused_eligible(log_n, w)) return LURKHIP_OK;
    LH_ARG(ctx, n >= 1 && n <= NTT_MAX_BATCH, "NTT batch size");
    uintptr_t all_ptrs = 0;
    for (int m = 0; m < n; m++) all_ptrs |= (uintptr_t)inter[m] | (uintptr_t)out[m][0] | (uintptr_t)out[m][1];
    if ((all_ptrs & 7u) != 0) return LURKHIP_OK;
    const ChunkPlan cp = plan_chunks(log_n, w, true);
    const int log_r = cp.max_log_r, col_chunk = cp.col_chunk;
    std::vector<std::pair<int, int>> passes;
    schedule(log_n, log_r, passes);
    if (passes[0].second != log_r) return LURKHIP_OK;  // the forward first pass is the tallest by construction
    const int n_full = w / col_chunk, last_w = w % col_chunk;
    const int U = ntt_rows_per_thread(log_r), slots = (1 << log_r) / U;
    const int max_thr = ntt_max_threads(log_r);
    FusedArgs a{};
    for (int m = 0; m < n; m++) {
        a.in[m] = inter[m];
        for (int q = 0; q < 2; q++) {
            a.out[m][q] = out[m][q];
            a.scale[m][q] = scale[m][q];
        }
    }
    a.tw_inv = plan.tw_inv;
    a.tw_fwd = plan.tw_fwd;
    a.log_n = log_n;
    a.w = w;
    a.n_cosets = 2;
    for (int part = 0; part < 2; part++) {
        if (part == 0 && n_full == 0) continue;
        if (part == 1 && last_w == 0) continue;
        a.col0 = part == 0 ? 0 : n_full * col_chunk;
        a.col_chunk = part == 0 ? col_chunk : last_w;
        a.n_chunks = part == 0 ? n_full : 1;
        const bool pair = a.col_chunk % 2 == 0 && a.col0 % 2 == 0;
        const int cv = a.col_chunk / (pair ? 2 : 1);
        if (slots * cv > max_thr || (1 << log_r) > max_thr) return set_error(ctx, LURKHIP_ERR_INVALID_ARG, "fused NTT tile shape");
        a.magic_cv = magic_for(cv);
        const int threads = std::min(max_thr, (std::max(slots * cv, 1 << log_r) + 63) / 64 * 64);
        const size_t tiles = ((size_t)1 << (log_n - log_r)) * a.n_chunks;
        a.n_tiles = (uint32_t)tiles;
        a.xcd_run = (tiles % 8 == 0 && tiles >= 64) ? (uint32_t)(tiles / 8) : 0u;
        const size_t lds = ((size_t)a.col_chunk * (((size_t)1 << log_r) + 1) + ((size_t)2 << log_r)) * 4;
        const int per_cu = std::max(1, std::min(20 / std::max(1, threads / 64), (int)((160 * 1024) / (lds + 256))));
        size_t blocks = std::min<size_t>(tiles, std::max<size_t>(8, (size_t)per_cu * ctx->num_cus / n));
        if (a.xcd_run) blocks = blocks / 8 * 8;
        const bool big = (((size_t)w) << (log_n + 2)) >= ((size_t)1 << 32) || getenv("LURKHIP_NTT_FORCE_64BIT") != nullptr;
        const dim3 grid((unsigned)blocks, (unsigned)n);
        if (pair && big) launch_fused<uint2, true>(log_r, grid, threads, lds, ctx->stream, a);
        else if (pair) launch_fused<uint2, false>(log_r, grid, threads, lds, ctx->stream, a);
        else if (big) launch_fused<uint32_t, true>(log_r, grid, threads, lds, ctx->stream, a);
        else launch_fused<uint32_t, false>(log_r, grid, threads, lds, ctx->stream, a);
        LH_HIP(ctx, hipGetLastError());
    }
    *done = true;
    return LURKHIP_OK;
}

}  // namespace lurkhip
