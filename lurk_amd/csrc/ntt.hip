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
#include <vector>

#include "babybear.h"
#include "commit.h"
#include "ctx.h"

namespace lurkhip {

namespace {

constexpr int NTT_BLOCK = 256;

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
    const uint32_t* in;    // N x w
    uint32_t* out;         // N x w (may alias in when !bitrev_store and in-place is wanted)
    const uint32_t* tw;    // powers of the size-N root (or inverse root), N/2 entries, Montgomery
    const uint32_t* row_scale;  // optional per-row multiplier applied on load (N entries) or nullptr
    int log_n;
    int w;
    int bit_lo;        // lowest row-index bit handled by this pass
    int log_r;         // number of stages (tile rows = 1 << log_r)
    int col_chunk;     // columns per tile
    int in_canonical;  // convert on load
    int out_canonical; // convert on store
    int bitrev_store;  // store row r at bitrev(r, log_n)
    uint32_t magic_c;  // ceil(2^32 / col_chunk): e / C == umulhi(e, magic) for e < 2^17, C < 2^10
    uint32_t magic_c2; // same for C / 2 (two-column butterflies), 0 when C is odd
    uint32_t magic_last;   // the same pair for the ragged last column chunk (w % col_chunk columns)
    uint32_t magic_last2;
    // Row grouping for narrow matrices: a tile takes 2^log_l ADJACENT rows (consecutive values of the low row bits)
    // for each of its 2^log_r strided rows, so that every global access is a contiguous run of (w << log_l) words
    // instead of w.  Then col_chunk = w << log_l (one chunk) and magic_w divides a tile column by w.
    int log_l;
    uint32_t magic_w;
};

__device__ __forceinline__ int fast_div(uint32_t e, uint32_t magic, int c) {
    return c == 1 ? (int)e : (int)__umulhi(e, magic);
}


// x <- x + y, y <- (x - y) * tw  (decimation in frequency); (x - y + P) < 2P is a valid Montgomery operand next to a
// reduced twiddle
__device__ __forceinline__ void dif_butterfly(uint32_t& x, uint32_t& y, uint32_t tw) {
    const uint32_t sum = bb::add(x, y);
    y = bb::mul(x + bb::P - y, tw);
    x = sum;
}
__device__ __forceinline__ void dif_butterfly(uint2& x, uint2& y, uint32_t tw) {
    dif_butterfly(x.x, y.x, tw);
    dif_butterfly(x.y, y.y, tw);
}

// G consecutive DIF stages s_top .. s_top-G+1 of the LDS tile [R][Cv] (elements of type T = one or two columns).
// Item (q, c): the 2^G rows t0 | b << s_bot, b < 2^G, of column c, where t0 is q with G zero bits inserted at s_bot.
template <int G, class T>
__device__ __forceinline__ void stage_group(T* __restrict__ tile, const uint32_t* __restrict__ tw_lds, int R, int Cv, int s_top, int log_l,
                                            uint32_t magic_cv, uint32_t magic_w, int w, int cols_per_item, int NT) {
    constexpr int M = 1 << G;
    const int s_bot = s_top - G + 1;
    const int items = (R >> G) * Cv;
    for (int e = threadIdx.x; e < items; e += NT) {
        const int q = fast_div(e, magic_cv, Cv), c = e - q * Cv;
        const int low = q & ((1 << s_bot) - 1);
        const int t0 = ((q >> s_bot) << (s_top + 1)) | low;
        const int l = log_l ? fast_div((uint32_t)(cols_per_item * c), magic_w, w) : 0;
        T x[M];
#pragma unroll
        for (int b = 0; b < M; b++) x[b] = tile[(t0 + (b << s_bot)) * Cv + c];
#pragma unroll
        for (int g = G - 1; g >= 0; g--) {
            const int st = s_bot + g;  // stage: pair distance 2^g in b
#pragma unroll
            for (int b = 0; b < M; b++) {
                if (b & (1 << g)) continue;
                // twiddle index of the butterfly whose upper row is t0 | b << s_bot: its low `st` bits
                const int t_lo = low | ((b & ((1 << g) - 1)) << s_bot);
                const uint32_t tw = tw_lds[(((1 << st) + t_lo) << log_l) + l];
                dif_butterfly(x[b], x[b | (1 << g)], tw);
            }
        }
#pragma unroll
        for (int b = 0; b < M; b++) tile[(t0 + (b << s_bot)) * Cv + c] = x[b];
    }
}

// One pass: tile = rows { hi << (bit_lo+log_r) | t << bit_lo | lo : t < 2^log_r } x col_chunk columns.
// LDS: [R][C] tile followed by the pass's twiddles: tw_lds[(1 << s) + t_lo] for stage s.
__global__ __launch_bounds__(1024) void k_ntt_pass(PassArgs a) {
    const int NT = (int)blockDim.x;  // 256, or 1024 for large tiles: the stages are a latency chain per workgroup
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    const int R = 1 << a.log_r;
    const int n_col_chunks = a.log_l ? 1 : (a.w + a.col_chunk - 1) / a.col_chunk;
    const uint32_t tile_id = blockIdx.x / n_col_chunks;
    const int chunk = blockIdx.x - tile_id * n_col_chunks;
    const int col0 = chunk * a.col_chunk;
    const int C = a.log_l ? a.col_chunk : min(a.col_chunk, a.w - col0);
    const bool is_full = C == a.col_chunk;
    // every chunk takes the multiply-high division: the ragged last chunk has its own magic numbers
    const uint32_t mg = is_full ? a.magic_c : a.magic_last, mg2 = is_full ? a.magic_c2 : a.magic_last2;
    const int L = 1 << a.log_l;
    const uint32_t lo_bits = (uint32_t)(a.bit_lo - a.log_l);
    const uint32_t lo = (tile_id & ((1u << lo_bits) - 1u)) << a.log_l;  // first of the tile's L adjacent low-bit values
    const uint32_t hi = tile_id >> lo_bits;
    const uint32_t row_base = (hi << (a.bit_lo + a.log_r)) | lo;
    uint32_t* tw_lds = tile + R * a.col_chunk;

    // stage twiddles: w_{2h}^j, h = 2^(bit_lo+s), j = (t_lo << bit_lo) | (lo + l); stored at [k * L + l]
    for (int idx = threadIdx.x + L; idx < R * L; idx += NT) {
        int k = idx >> a.log_l, l = idx & (L - 1);
        int s = 31 - __clz(k);
        uint32_t t_lo = (uint32_t)k - (1u << s);
        uint32_t j = (t_lo << a.bit_lo) | (lo + (uint32_t)l);
        tw_lds[idx] = a.tw[(size_t)j << (a.log_n - a.bit_lo - s - 1)];
    }
    // load: tile row t = the L adjacent matrix rows row_base | t << bit_lo .. + L - 1, contiguous in memory.
    // Thread (tr, tc) walks rows tr, tr + RS, ... of column tc: one division per thread, none per element.
    const int RS = NT / C;                       // tile rows covered per sweep (C <= 128 <= NT)
    const int tr = (int)threadIdx.x / C, tc = (int)threadIdx.x - tr * C;
    const bool lane_on = tr < RS;
    const uint32_t l_of_tc = a.log_l ? (uint32_t)fast_div((uint32_t)tc, a.magic_w, a.w) : 0u;
    if (lane_on) {
        const uint32_t* __restrict__ src = a.in + col0 + tc;
        for (int t = tr; t < R; t += RS) {
            const uint32_t row = row_base | ((uint32_t)t << a.bit_lo);
            uint32_t v = src[(size_t)row * a.w];
            if (a.in_canonical) v = bb::to_monty(v);
            if (a.row_scale) v = bb::mul(v, a.row_scale[row + l_of_tc]);
            tile[t * C + tc] = v;
        }
    }
    __syncthreads();
    // DIF stages s = log_r-1 .. 0 (pair distance 2^s tile rows), taken in groups of up to three: a thread holds the
    // 2^g rows of one column (or column pair) in registers for g consecutive stages, so the tile makes one LDS round
    // trip -- and one index computation -- per group instead of per stage (the passes are int32-issue bound).
    const bool pairs = mg2 != 0;  // two adjacent columns per lane (8-byte LDS accesses; C even, rows 8-byte aligned)
    const int Cv = pairs ? (C >> 1) : C;
    const uint32_t mgv = pairs ? mg2 : mg;
    int remaining = a.log_r, s_top = a.log_r - 1;
    while (remaining > 0) {
        const int g = remaining == 4 ? 2 : (remaining >= 3 ? 3 : remaining);
        if (pairs) {
            uint2* t2 = reinterpret_cast<uint2*>(tile);
            if (g == 3) stage_group<3, uint2>(t2, tw_lds, R, Cv, s_top, a.log_l, mgv, a.magic_w, a.w, 2, NT);
            else if (g == 2) stage_group<2, uint2>(t2, tw_lds, R, Cv, s_top, a.log_l, mgv, a.magic_w, a.w, 2, NT);
            else stage_group<1, uint2>(t2, tw_lds, R, Cv, s_top, a.log_l, mgv, a.magic_w, a.w, 2, NT);
        } else {
            if (g == 3) stage_group<3, uint32_t>(tile, tw_lds, R, Cv, s_top, a.log_l, mgv, a.magic_w, a.w, 1, NT);
            else if (g == 2) stage_group<2, uint32_t>(tile, tw_lds, R, Cv, s_top, a.log_l, mgv, a.magic_w, a.w, 1, NT);
            else stage_group<1, uint32_t>(tile, tw_lds, R, Cv, s_top, a.log_l, mgv, a.magic_w, a.w, 1, NT);
        }
        __syncthreads();
        remaining -= g;
        s_top -= g;
    }
    // store (same thread-to-element map as the load)
    if (lane_on) {
        uint32_t* __restrict__ dst = a.out + col0 + tc;
        for (int t = tr; t < R; t += RS) {
            uint32_t row = row_base | ((uint32_t)t << a.bit_lo);
            if (a.bitrev_store) row = bitrev32(row, a.log_n);
            uint32_t v = tile[t * C + tc];
            if (a.out_canonical) v = bb::from_monty(v);
            dst[(size_t)row * a.w] = v;
        }
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
    // w_{2^bits} = root27^(2^(27-bits))
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

// Full size-N DIF transform of an N x w row-major matrix.
//   src -> dst, natural-order input; output bit-reversed, or natural when bitrev_store (the
//   permutation is fused into the last pass's store).  `scratch` (N x w) is needed when more than one
//   pass runs and the last pass scatters (it must not alias dst).  row_scale multiplies row i on load.
int32_t ntt_dif(lurkhip_ctx* ctx, const NttPlan& plan, bool inverse, const uint32_t* src, uint32_t* dst,
                uint32_t* scratch, int w, const uint32_t* row_scale, bool in_canonical, bool out_canonical,
                bool bitrev_store) {
    const int log_n = plan.log_n;
    if (log_n == 0) {
        // 1 x w: identity (times scale)
        PassArgs a{src, dst, plan.tw_fwd, row_scale, 0, w, 0, 0, w < 64 ? w : 64, in_canonical, out_canonical, 0, 0, 0};
        a.magic_c = magic_for(a.col_chunk);
        a.magic_last = (w % a.col_chunk) ? magic_for(w % a.col_chunk) : 0;
        a.magic_last2 = 0;
        a.log_l = 0;
        a.magic_w = 0;
        int chunks = (w + a.col_chunk - 1) / a.col_chunk;
        hipLaunchKernelGGL(k_ntt_pass, dim3(chunks), dim3(NTT_BLOCK), (size_t)(a.col_chunk + 1) * 4, ctx->stream, a);
        LH_HIP(ctx, hipGetLastError());
        return LURKHIP_OK;
    }
    // tile budget: rows * cols * 4 B <= 64 KiB so two blocks fit a CU.  Prefer a chunk width that divides w (no
    // ragged chunk) and is even (two-column butterflies): the largest such divisor in [32, 112], else 64.
    int col_chunk = w;
    if (w > 112) {
        col_chunk = 64;
        for (int c = 112; c >= 32; c--)
            if (w % c == 0 && c % 2 == 0) {
                col_chunk = c;
                break;
            }
    }
    int max_log_r = 7;
    while (((size_t)1 << max_log_r) * col_chunk * 4 > 64 * 1024 && max_log_r > 1) max_log_r--;
    std::vector<std::pair<int, int>> passes;
    schedule(log_n, max_log_r, passes);
    const int n_chunks = (w + col_chunk - 1) / col_chunk;
    const uint32_t* cur_in = src;
    for (size_t p = 0; p < passes.size(); p++) {
        bool last = p + 1 == passes.size();
        uint32_t* cur_out;
        if (last) cur_out = dst;
        else if (bitrev_store) cur_out = scratch;   // keep dst free for the final scatter
        else cur_out = dst;                          // in-place chain inside dst
        PassArgs a;
        a.in = cur_in;
        a.out = cur_out;
        a.tw = inverse ? plan.tw_inv : plan.tw_fwd;
        a.row_scale = p == 0 ? row_scale : nullptr;
        a.log_n = log_n;
        a.w = w;
        a.bit_lo = passes[p].first;
        a.log_r = passes[p].second;
        a.col_chunk = col_chunk;
        a.in_canonical = (p == 0 && in_canonical) ? 1 : 0;
        a.out_canonical = (last && out_canonical) ? 1 : 0;
        a.bitrev_store = (last && bitrev_store) ? 1 : 0;
        a.magic_c = magic_for(col_chunk);
        a.magic_c2 = (col_chunk % 2 == 0 && col_chunk >= 4) ? magic_for(col_chunk / 2) : 0;
        // narrow matrices: group adjacent rows in the strided passes so that global runs are >= ~256 bytes
        int log_l = 0;
        if (n_chunks == 1 && a.bit_lo > 0 && !a.bitrev_store) {
            while (log_l < 4 && log_l < a.bit_lo && (w << (log_l + 1)) <= 128 &&
                   ((size_t)1 << a.log_r) * ((size_t)(w << (log_l + 1)) + (1u << (log_l + 1))) * 4 <= 64 * 1024)
                log_l++;
        }
        a.log_l = log_l;
        a.magic_w = magic_for(w);
        if (log_l > 0) {
            a.col_chunk = w << log_l;
            a.magic_c = magic_for(a.col_chunk);
            a.magic_c2 = (w % 2 == 0 && a.col_chunk >= 4) ? magic_for(a.col_chunk / 2) : 0;
        }
        const int last_w = w % col_chunk;
        a.magic_last = last_w ? magic_for(last_w) : 0;
        a.magic_last2 = (last_w && last_w % 2 == 0 && last_w >= 4) ? magic_for(last_w / 2) : 0;
        size_t tiles = ((size_t)1 << (log_n - a.log_r - a.log_l)) * n_chunks;
        LH_ARG(ctx, tiles <= 0x7fffffffu, "NTT grid too large");
        size_t lds = ((size_t)1 << a.log_r) * ((size_t)a.col_chunk + ((size_t)1 << a.log_l)) * 4;  // tile + per-pass twiddles
        const size_t tile_elems = ((size_t)1 << a.log_r) * a.col_chunk;
        // many tiles: 256-thread workgroups (several per CU overlap each other's load / stage / store phases);
        // few tiles: the per-workgroup latency chain dominates, so spread each tile over up to 1024 threads
        const int threads = tile_elems >= 4096 ? (tiles >= 4096 ? 512 : 1024) : (tile_elems >= 2048 ? 512 : NTT_BLOCK);
        hipLaunchKernelGGL(k_ntt_pass, dim3((unsigned)tiles), dim3(threads), lds, ctx->stream, a);
        LH_HIP(ctx, hipGetLastError());
        cur_in = cur_out;
    }
    return LURKHIP_OK;
}

}  // namespace lurkhip
