// Prover stages that consume a chip's AIR program: debug constraint check, LogUp permutation trace,
// quotient values.
//
// Replaces (third-party, source absent from /root/reference -> parity unpinned):
//   sphinx-core `generate_permutation_trace` / `eval_permutation_constraints` / `quotient_values`
//   / `debug_constraints`   [UPSTREAM-RECALL, sphinx @ 8a39b951 = SP1 v1 lineage]
// driven by the in-tree AIRs (/root/reference/src/lair/air.rs:158-552, src/lair/memory.rs:71-109,
// src/gadgets/bytes/trace.rs:117-143, src/lair/lair_chip.rs:166-191) and the in-tree lookup protocol
// (/root/reference/src/air/builder.rs:34-133).  The dead in-tree LogUp (src/logup/trace.rs:53-151) has the
// same shape (RLC denominators, batched inverses, running sum) and is covered by the same kernels.
//
// Permutation trace row i (sphinx `populate_permutation_row`):
//   for interaction j (sends first, then receives), with tuple v_0..v_{k-1}, multiplicity m, kind K:
//       d_j = alpha + K + sum_t beta^(t+1) v_t          (extension field)
//   batch column c = sum_{j in batch c} (+-m_j) / d_j   (+ for sends, - for receives; batch = 2^log_quotient_degree)
//   last column = inclusive running sum over rows of the row's batch columns.
#include <map>
#include <mutex>
#include <numeric>

#include "air_vm.h"
#include "babybear.h"
#include "commit.h"
#include "ctx.h"
#include "lair/air.h"
#include <string.h>

#include "jit.h"
#include "lazy_ef.h"
#include "stark_kernels.h"
#include "stark.h"

struct lurkhip_air {
    lair::ChipAir air;
    lair::AirPrograms prog;
    std::mutex mu;
    struct DevPrograms {
        uint32_t* cons = nullptr;
        uint32_t* inter = nullptr;
        uint32_t* inter_static = nullptr;  // k_interaction_starts: kinds, offsets, constant terms
        std::vector<uint32_t*> cons_parts;    // constraint program pieces (AirPrograms::constraint_parts)
        std::vector<uint32_t*> parts;         // interaction program pieces (AirPrograms::interaction_parts)
        std::vector<uint32_t*> parts_coarse;  // AirPrograms::interaction_parts_coarse
        lurkhip::JitKernels jit;              // run-time compiled kernels of this chip (lurkhip_air_compile), or empty
    };
    std::map<int, DevPrograms> dev;  // device -> programs
    uint32_t tuple_words = 0;                              // sum over interactions of (1 + #values)
    uint32_t max_tuple = 0;
};

namespace lurkhip {

int32_t air_programs_dev(lurkhip_ctx* ctx, lurkhip_air* a, const uint32_t** cons, const uint32_t** inter,
                         const std::vector<uint32_t*>** parts, bool coarse, const uint32_t** inter_static,
                         const std::vector<uint32_t*>** cons_parts) {
    std::lock_guard<std::mutex> g(a->mu);
    auto it = a->dev.find(ctx->device);
    if (it == a->dev.end()) {
        lurkhip_air::DevPrograms d;
        auto upload = [&](const std::vector<uint32_t>& words, uint32_t** out) -> int32_t {
            LH_HIP(ctx, hipMalloc(out, words.size() * 4));
            LH_HIP(ctx, hipMemcpy(*out, words.data(), words.size() * 4, hipMemcpyHostToDevice));
            return LURKHIP_OK;
        };
        LH_TRY(upload(a->prog.constraints, &d.cons));
        LH_TRY(upload(a->prog.interactions, &d.inter));
        {
            const auto& pr = a->prog;
            const uint32_t n = (uint32_t)pr.interaction_kinds.size();
            std::vector<uint32_t> st{n};
            st.insert(st.end(), pr.interaction_kinds.begin(), pr.interaction_kinds.end());
            std::vector<uint32_t> offs(n + 1, 0);
            for (const auto& t : pr.const_terms) offs[t.interaction + 1]++;
            for (uint32_t j = 0; j < n; j++) offs[j + 1] += offs[j];
            st.insert(st.end(), offs.begin(), offs.end());
            for (const auto& t : pr.const_terms) {  // const_terms are in interaction order
                st.push_back(t.t);
                st.push_back(bb::to_monty(t.value % bb::P));
            }
            LH_TRY(upload(st, &d.inter_static));
        }
        for (const auto& part : a->prog.constraint_parts) {
            uint32_t* dp = nullptr;
            LH_TRY(upload(part, &dp));
            d.cons_parts.push_back(dp);
        }
        for (const auto& part : a->prog.interaction_parts) {
            uint32_t* dp = nullptr;
            LH_TRY(upload(part, &dp));
            d.parts.push_back(dp);
        }
        for (const auto& part : a->prog.interaction_parts_coarse) {
            uint32_t* dp = nullptr;
            LH_TRY(upload(part, &dp));
            d.parts_coarse.push_back(dp);
        }
        it = a->dev.emplace(ctx->device, std::move(d)).first;
    }
    if (cons) *cons = it->second.cons;
    if (inter_static) *inter_static = it->second.inter_static;
    if (inter) *inter = it->second.inter;
    if (parts) *parts = coarse ? &it->second.parts_coarse : &it->second.parts;
    if (cons_parts) *cons_parts = &it->second.cons_parts;
    return LURKHIP_OK;
}

// the chip's compiled kernels on this context's device, if lurkhip_air_compile was called for it
static JitKernels jit_of(lurkhip_ctx* ctx, lurkhip_air* a) {
    std::lock_guard<std::mutex> g(a->mu);
    auto it = a->dev.find(ctx->device);
    return it == a->dev.end() ? JitKernels{} : it->second.jit;
}

const lair::ChipAir& air_of(const lurkhip_air* a) { return a->air; }
const lair::AirPrograms& programs_of(const lurkhip_air* a) { return a->prog; }

// Launch shape of a VM kernel that stages its rows in LDS: 64-lane workgroups whose LDS holds regs[n_regs][64], then
// `tiles` row tiles of `tile_rows` x (w | 1) words (odd stride: a lane's row starts on its own bank), then 64 row
// indices.  Falls back to unstaged global reads (staged = false) when that does not fit.
struct VmShape {
    int block;
    size_t lds;
    bool staged;
    uint32_t wp;  // padded row stride of a tile
};
constexpr size_t VM_LDS_BUDGET = 64 * 1024;
int vm_block(uint32_t n_regs, size_t* lds_bytes);
VmShape vm_shape(uint32_t n_regs, uint32_t w, uint32_t tiles, uint32_t tile_rows, uint32_t extra_words = 0) {
    VmShape sh;
    sh.wp = w | 1u;
    size_t need = ((size_t)n_regs * 64 + (size_t)tiles * tile_rows * sh.wp + 80 + extra_words) * 4;  // + row indices (up to 65)
    if (need <= VM_LDS_BUDGET) {
        sh.block = 64;
        sh.lds = need;
        sh.staged = true;
    } else {
        sh.block = vm_block(n_regs, &sh.lds);
        sh.staged = false;
    }
    return sh;
}

// threads per block for a VM launch whose register file is regs[n_regs][block] in LDS
int vm_block(uint32_t n_regs, size_t* lds_bytes) {
    int block = 256;
    while (block > 64 && (size_t)n_regs * block * 4 > 48 * 1024) block >>= 1;
    *lds_bytes = (size_t)n_regs * block * 4;
    return block;
}

namespace {

using bb::ef;


// ---------------------------------------------------------------- explicit-row evaluation (debug / parity)
struct DumpSink {
    uint32_t* cons_out;   // [K] canonical
    uint32_t* inter_out;  // per interaction: multiplicity, then the tuple (canonical)
    uint32_t k = 0, t = 0;
    __device__ __forceinline__ void assert_zero(uint32_t v) { cons_out[k++] = bb::from_monty(v); }
    __device__ __forceinline__ void ibegin(uint32_t, bool, uint32_t) { t++; /* slot for the multiplicity */ base = t - 1; }
    __device__ __forceinline__ void ival(uint32_t v) { inter_out[t++] = bb::from_monty(v); }
    __device__ __forceinline__ void ival_at(uint32_t, uint32_t) {}  // compact pieces only (prover kernels)
    __device__ __forceinline__ void ival_run(const uint32_t*, uint32_t, uint32_t) {}
    __device__ __forceinline__ void iend(uint32_t m) { inter_out[base] = bb::from_monty(m); }
    uint32_t base = 0;
};

struct EvalRowsArgs {
    const uint32_t* cons_prog;
    const uint32_t* inter_prog;
    const uint32_t* local;  // [n][w] Montgomery
    const uint32_t* next;
    const uint32_t* prep_local;
    const uint32_t* prep_next;
    const uint32_t* pub;
    const uint32_t* sels;  // [n][3] Montgomery
    uint32_t* cons_out;    // [n][K]
    uint32_t* inter_out;   // [n][T]
    uint32_t n, w, pw, K, T;
};

__global__ void k_air_eval_rows(EvalRowsArgs a) {
    extern __shared__ uint32_t regs[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    airvm::Sources s{a.local + (size_t)i * a.w, a.next + (size_t)i * a.w, a.prep_local + (size_t)i * a.pw,
                     a.prep_next + (size_t)i * a.pw, a.pub, {a.sels[3 * i], a.sels[3 * i + 1], a.sels[3 * i + 2]}};
    DumpSink sink{a.cons_out + (size_t)i * a.K, a.inter_out + (size_t)i * a.T};
    airvm::run(a.cons_prog, s, regs + threadIdx.x, blockDim.x, sink);
    airvm::run(a.inter_prog, s, regs + threadIdx.x, blockDim.x, sink);
}

// ---------------------------------------------------------------- debug check of a whole trace
struct CheckSink {
    unsigned long long* first_bad;
    uint32_t row;
    uint32_t k = 0;
    __device__ __forceinline__ void assert_zero(uint32_t v) {
        if (v != 0) atomicMin(first_bad, ((unsigned long long)row << 32) | k);
        k++;
    }
    __device__ __forceinline__ void ibegin(uint32_t, bool, uint32_t) {}
    __device__ __forceinline__ void ival(uint32_t) {}
    __device__ __forceinline__ void ival_at(uint32_t, uint32_t) {}
    __device__ __forceinline__ void ival_run(const uint32_t*, uint32_t, uint32_t) {}
    __device__ __forceinline__ void iend(uint32_t) {}
};

__global__ void k_air_check(const uint32_t* __restrict__ prog, const uint32_t* __restrict__ main, const uint32_t* __restrict__ prep,
                            const uint32_t* __restrict__ pub, uint32_t n, uint32_t w, uint32_t pw, unsigned long long* first_bad,
                            uint32_t n_regs, uint32_t wp, int staged) {
    extern __shared__ uint32_t lds[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nx = i + 1 >= n ? 0 : i + 1;
    const uint32_t* main_l = main + (size_t)(i < n ? i : 0) * w;
    const uint32_t* main_n = main + (size_t)nx * w;
    if (staged) {
        // rows i0 .. i0 + 64 (local of lane l = tile row l, next = tile row l + 1; the wrap row is staged last)
        uint32_t* tile = lds + n_regs * blockDim.x;
        uint32_t* idx = tile + (blockDim.x + 1) * wp;
        idx[threadIdx.x] = i < n ? i : 0;
        if (threadIdx.x == blockDim.x - 1) idx[blockDim.x] = nx;
        __syncthreads();
        stage_rows(tile, wp, main, w, idx, blockDim.x + 1);
        __syncthreads();
        main_l = tile + threadIdx.x * wp;
        main_n = tile + (threadIdx.x + 1) * wp;
        // a lane that is the last real row but not the last lane of the block needs row 0 as its next row
        if (i + 1 == n && threadIdx.x + 1 < blockDim.x) main_n = main;
    }
    if (i >= n) return;
    // the selectors of a row-by-row checker (p3 check_constraints): indicator values on the trace domain
    airvm::Sources s{main_l, main_n, prep + (size_t)i * pw, prep + (size_t)nx * pw, pub,
                     {i == 0 ? bb::R1 : 0u, i + 1 == n ? bb::R1 : 0u, i + 1 == n ? 0u : bb::R1}};
    CheckSink sink{first_bad, i};
    airvm::run(prog, s, lds + threadIdx.x, blockDim.x, sink);
}

// ---------------------------------------------------------------- permutation trace rows
// out[i] = base^i, Montgomery.  centred == 0: 4 canonical words per power.  centred == 1: 8 words per power for the lazy
// 64-bit accumulators below: the coefficients c0..c3 and 11*c1, 11*c2, 11*c3 (the wrap-around factors of x^4 = 11) as
// signed representatives in (-p/2, p/2], then a zero.
// reversed != 0: power i is stored at slot count - 1 - i.
__global__ void k_ef_powers(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t* __restrict__ out_base, uint32_t count,
                            int centred, int reversed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t* out = out_base;
    const uint32_t slot = reversed ? count - 1 - i : i;
    ef base{{b0, b1, b2, b3}}, r = bb::ef_one();
    uint32_t e = i;
    while (e) {
        if (e & 1u) r = bb::ef_mul(r, base);
        base = bb::ef_sqr(base);
        e >>= 1;
    }
    if (!centred) {
        out[4 * slot] = r.c[0];
        out[4 * slot + 1] = r.c[1];
        out[4 * slot + 2] = r.c[2];
        out[4 * slot + 3] = r.c[3];
        return;
    }
    auto centre = [](uint32_t x) -> uint32_t { return x > bb::P / 2 ? x - bb::P : x; };
    for (int c = 0; c < 4; c++) out[8 * slot + c] = centre(r.c[c]);
    for (int c = 1; c < 4; c++) out[8 * slot + 3 + c] = centre(bb::mul(bb::EXT_W_M, r.c[c]));
    out[8 * slot + 7] = 0;
}

}  // namespace

// the same with the base read from device memory (a challenge the device drew: prover.hip, the constraint-folding alpha)
__global__ void k_ef_powers_dev(const uint32_t* __restrict__ base_dev, uint32_t* __restrict__ out, uint32_t count, int centred, int reversed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t slot = reversed ? count - 1 - i : i;
    ef base{{base_dev[0], base_dev[1], base_dev[2], base_dev[3]}}, r = bb::ef_one();
    uint32_t e = i;
    while (e) {
        if (e & 1u) r = bb::ef_mul(r, base);
        base = bb::ef_sqr(base);
        e >>= 1;
    }
    if (!centred) {
        for (int c = 0; c < 4; c++) out[4 * slot + c] = r.c[c];
        return;
    }
    auto centre = [](uint32_t x) -> uint32_t { return x > bb::P / 2 ? x - bb::P : x; };
    for (int c = 0; c < 4; c++) out[8 * slot + c] = centre(r.c[c]);
    for (int c = 1; c < 4; c++) out[8 * slot + 3 + c] = centre(bb::mul(bb::EXT_W_M, r.c[c]));
    out[8 * slot + 7] = 0;
}

namespace {

// Start value of every interaction's denominator: alpha + kind + sum over its constant tuple elements of beta^t * c.
// stat = [n | kinds[n] | offsets[n + 1] | (t, constant in Montgomery form) pairs], uploaded once per chip; beta_pows is the
// centred 8-word table.  One thread per interaction, once per proof and chip.
__global__ void k_interaction_starts(const uint32_t* __restrict__ stat, const uint32_t* __restrict__ beta_pows, ef alpha,
                                     uint32_t* __restrict__ starts) {
    const uint32_t n = stat[0], j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t* kinds = stat + 1;
    const uint32_t* offs = kinds + n;
    const uint32_t* terms = offs + n + 1;
    ef s_ = bb::ef_add_base(alpha, bb::to_monty(kinds[j]));
    for (uint32_t e = offs[j]; e < offs[j + 1]; e++) {
        const uint32_t t = terms[2 * e], c = terms[2 * e + 1];
        ef pw;
        for (int k = 0; k < 4; k++) {
            const uint32_t x = beta_pows[8 * t + k];  // centred -> canonical
            pw.c[k] = (int32_t)x < 0 ? x + bb::P : x;
        }
        s_ = bb::ef_add(s_, bb::ef_scale(pw, c));
    }
    for (int k = 0; k < 4; k++) starts[4 * j + k] = s_.c[k];
}

// the same for several chips in one launch (a small proof is made of launches): block b belongs to the chip whose first_block is the
// last one at or below b
constexpr int STARTS_BATCH = 48;
struct StartsBatch {
    const uint32_t* stat[STARTS_BATCH];
    uint32_t* starts[STARTS_BATCH];
    uint32_t first_block[STARTS_BATCH + 1];
    uint32_t n;
};
__global__ void k_interaction_starts_batch(StartsBatch b, const uint32_t* __restrict__ beta_pows, ef alpha) {
    uint32_t c = 0;
    for (uint32_t i = 1; i < b.n; i++)
        if (blockIdx.x >= b.first_block[i]) c = i;
    const uint32_t* __restrict__ stat = b.stat[c];
    const uint32_t n = stat[0], j = (blockIdx.x - b.first_block[c]) * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t* kinds = stat + 1;
    const uint32_t* offs = kinds + n;
    const uint32_t* terms = offs + n + 1;
    ef s_ = bb::ef_add_base(alpha, bb::to_monty(kinds[j]));
    for (uint32_t e = offs[j]; e < offs[j + 1]; e++) {
        const uint32_t t = terms[2 * e], cst = terms[2 * e + 1];
        ef pw;
        for (int k = 0; k < 4; k++) {
            const uint32_t x = beta_pows[8 * t + k];  // centred -> canonical
            pw.c[k] = (int32_t)x < 0 ? x + bb::P : x;
        }
        s_ = bb::ef_add(s_, bb::ef_scale(pw, cst));
    }
    for (int k = 0; k < 4; k++) b.starts[c][4 * j + k] = s_.c[k];
}

__global__ void k_perm_rows(PermArgs a) { perm_rows_body<InterpreterRunner>(a); }

// ---------------------------------------------------------------- inclusive scan of an EF column
// (the sequential loop of /root/reference/src/logup/trace.rs:142-148 / sphinx's `scan`), three launches:
// chunk-local scans, scan of the chunk totals, offset add.  EF addition is component-wise.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_PER_THREAD = 4;
constexpr int SCAN_CHUNK = SCAN_BLOCK * SCAN_PER_THREAD;

__device__ __forceinline__ uint4 add4(uint4 a, uint4 b) {
    return make_uint4(bb::add(a.x, b.x), bb::add(a.y, b.y), bb::add(a.z, b.z), bb::add(a.w, b.w));
}

// data: element r at data[r * stride_words .. +4]
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_local(uint32_t* __restrict__ data, size_t stride_words, size_t n,
                                                            uint4* __restrict__ totals) {
    __shared__ uint4 sh[SCAN_BLOCK];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    uint4 v[SCAN_PER_THREAD];
    uint4 run = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        size_t r = base + k;
        uint4 x = r < n ? *reinterpret_cast<const uint4*>(data + r * stride_words) : make_uint4(0, 0, 0, 0);
        run = add4(run, x);
        v[k] = run;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    // Hillis-Steele over the per-thread totals
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if ((int)threadIdx.x >= off) t = sh[threadIdx.x - off];
        __syncthreads();
        if ((int)threadIdx.x >= off) sh[threadIdx.x] = add4(sh[threadIdx.x], t);
        __syncthreads();
    }
    uint4 prefix = threadIdx.x ? sh[threadIdx.x - 1] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        size_t r = base + k;
        if (r < n) *reinterpret_cast<uint4*>(data + r * stride_words) = add4(v[k], prefix);
    }
    if (threadIdx.x == SCAN_BLOCK - 1) totals[blockIdx.x] = sh[SCAN_BLOCK - 1];
}

// the same for several one-chunk columns in one launch (block b scans column b): the short chips of a small proof
constexpr int SCAN_BATCH = 48;
struct ScanBatch {
    uint32_t* data[SCAN_BATCH];
    uint32_t stride[SCAN_BATCH], n[SCAN_BATCH];
};
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_local_batch(ScanBatch b) {
    __shared__ uint4 sh[SCAN_BLOCK];
    uint32_t* __restrict__ data = b.data[blockIdx.x];
    const size_t stride_words = b.stride[blockIdx.x], n = b.n[blockIdx.x];
    const size_t base = (size_t)threadIdx.x * SCAN_PER_THREAD;
    uint4 v[SCAN_PER_THREAD];
    uint4 run = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        size_t r = base + k;
        uint4 x = r < n ? *reinterpret_cast<const uint4*>(data + r * stride_words) : make_uint4(0, 0, 0, 0);
        run = add4(run, x);
        v[k] = run;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if ((int)threadIdx.x >= off) t = sh[threadIdx.x - off];
        __syncthreads();
        if ((int)threadIdx.x >= off) sh[threadIdx.x] = add4(sh[threadIdx.x], t);
        __syncthreads();
    }
    uint4 prefix = threadIdx.x ? sh[threadIdx.x - 1] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        size_t r = base + k;
        if (r < n) *reinterpret_cast<uint4*>(data + r * stride_words) = add4(v[k], prefix);
    }
}

// exclusive scan of `count` totals in one workgroup (count <= 2^27 / 1024 fits a loop)
__global__ __launch_bounds__(1024) void k_scan_totals(uint4* __restrict__ totals, size_t count) {
    __shared__ uint4 sh[1024];
    uint4 carry = make_uint4(0, 0, 0, 0);
    for (size_t base = 0; base < count; base += 1024) {
        size_t i = base + threadIdx.x;
        uint4 x = i < count ? totals[i] : make_uint4(0, 0, 0, 0);
        sh[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint4 t = make_uint4(0, 0, 0, 0);
            if ((int)threadIdx.x >= off) t = sh[threadIdx.x - off];
            __syncthreads();
            if ((int)threadIdx.x >= off) sh[threadIdx.x] = add4(sh[threadIdx.x], t);
            __syncthreads();
        }
        uint4 incl = sh[threadIdx.x];
        uint4 excl = threadIdx.x ? sh[threadIdx.x - 1] : make_uint4(0, 0, 0, 0);
        if (i < count) totals[i] = add4(carry, excl);
        uint4 last = sh[1023];
        __syncthreads();
        carry = add4(carry, last);
        (void)incl;
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_add(uint32_t* __restrict__ data, size_t stride_words, size_t n,
                                                          const uint4* __restrict__ offsets) {
    if (blockIdx.x == 0) return;
    const uint4 off = offsets[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_PER_THREAD;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        size_t r = base + k;
        if (r < n) {
            uint4* p = reinterpret_cast<uint4*>(data + r * stride_words);
            *p = add4(*p, off);
        }
    }
}

}  // namespace

namespace {

__global__ void k_quotient(QuotientArgs a) { quotient_body<InterpreterRunner>(a); }

// out[3 s ..] = is_first_row, is_last_row, is_transition (p3 TwoAdicMultiplicativeCoset::selectors_on_coset) at
// x = g w_Q^i, i = bitrev(s): zh_i / (x - 1), zh_i / (x - w_N^-1), x - w_N^-1.  A thread takes four rows and inverts their eight
// denominators with one Fermat ladder (Montgomery's trick); w_Q^i comes from the NTT twiddle table (tw[i], i < Q / 2).
struct SelectorArgs {
    uint32_t log_q, lqd, g_m, wn_inv_m;
    uint32_t zh[4];
    const uint32_t* tw;
    uint32_t* out;
};
__global__ __launch_bounds__(256) void k_selectors(SelectorArgs a) {
    const uint32_t q = 1u << a.log_q, half = q >> 1, qd_mask = (1u << a.lqd) - 1u;
    const uint32_t s0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (s0 >= q) return;
    uint32_t den[8], zh[4], pre[8];
    uint32_t acc = bb::R1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t s = s0 + k < q ? s0 + k : q - 1;
        const uint32_t i = a.log_q ? (__brev(s) >> (32 - a.log_q)) : 0u;
        const uint32_t wi = half == 0 ? bb::R1 : (i < half ? a.tw[i] : bb::neg(a.tw[i - half]));
        const uint32_t x = bb::mul(a.g_m, wi);
        den[2 * k] = bb::sub(x, bb::R1);
        den[2 * k + 1] = bb::sub(x, a.wn_inv_m);
        zh[k] = a.zh[i & qd_mask];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        pre[k] = acc;
        acc = bb::mul(acc, den[k] ? den[k] : bb::R1);
    }
    uint32_t inv = bb::inv(acc);
    uint32_t dinv[8];
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        dinv[k] = den[k] ? bb::mul(inv, pre[k]) : 0u;  // (a zero denominator inverts to zero like bb::inv; the coset never meets H)
        inv = bb::mul(inv, den[k] ? den[k] : bb::R1);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (s0 + k >= q) break;
        uint32_t* o = a.out + 3 * (size_t)(s0 + k);
        o[0] = bb::mul(zh[k], dinv[2 * k]);
        o[1] = bb::mul(zh[k], dinv[2 * k + 1]);
        o[2] = den[2 * k + 1];
    }
}

}  // namespace

// LDS layout of a multi-piece VM launch: the pieces' register files, the staged tile of 64 rows (when it fits), 64 row
// indices, one extension element per piece and lane for the final combination.
struct PartLayout {
    VmParts parts;
    uint32_t regs_words;
    uint32_t wp;
    bool staged;
    size_t lds_bytes;
};
// `compiled`: the launch runs the chip's compiled kernels, whose values live in VGPRs -- no register files in LDS.  (They were
// reserved all the same until the end of round 2: 64 lanes x n_regs words per piece, 30 KB and more beside a 20 KB tile, so a CU
// held two or three workgroups of the compiled kernels and they waited for memory half of their cycles.)
static PartLayout layout_parts(const std::vector<const std::vector<uint32_t>*>& host, const std::vector<uint32_t*>& dev, uint32_t w,
                               bool compiled) {
    PartLayout l{};
    l.parts.n_parts = (uint32_t)host.size();
    uint32_t off = 0;
    for (size_t j = 0; j < host.size(); j++) {
        l.parts.prog[j] = dev[j];
        l.parts.reg_off[j] = off;
        if (!compiled) off += (*host[j])[airp::H_N_REGS] * 64u;
    }
    // the pieces dealt to the waves so that the four SIMDs' loads are level (VmParts::piece_of): longest piece first, each to the
    // least loaded SIMD that still has a wave free (SIMD s takes waves s, s + 4, ..); LURKHIP_BALANCE_WAVES=0: piece order (round 5)
    {
        const uint32_t n = l.parts.n_parts;
        static const bool balance = getenv("LURKHIP_BALANCE_WAVES") == nullptr || atoi(getenv("LURKHIP_BALANCE_WAVES")) != 0;
        for (uint32_t j = 0; j < n; j++) l.parts.piece_of[j] = (uint8_t)j;
        if (balance && n > 4) {
            std::vector<uint32_t> order(n);
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return host[x]->size() > host[y]->size(); });
            uint64_t load[4] = {0, 0, 0, 0};
            uint32_t taken[4] = {0, 0, 0, 0};
            for (uint32_t j : order) {
                int best = -1;
                for (int sd = 0; sd < 4; sd++) {
                    const uint32_t cap = (n + 3u - (uint32_t)sd) / 4u;  // waves sd, sd + 4, .. below n
                    if (taken[sd] < cap && (best < 0 || load[sd] < load[best])) best = sd;
                }
                l.parts.piece_of[(uint32_t)best + 4u * taken[best]] = (uint8_t)j;
                taken[best]++;
                load[best] += host[j]->size();
            }
        }
    }
    l.regs_words = off;
    l.wp = w | 1u;
    const size_t fixed = (size_t)off + 64 + (size_t)host.size() * 512;  // row indices; per piece and lane one fold and one column sum
    l.staged = (fixed + 64u * l.wp) * 4 <= VM_LDS_BUDGET;
    l.lds_bytes = (fixed + (l.staged ? 64u * l.wp : 0u)) * 4;
    return l;
}

int32_t ef_powers(lurkhip_ctx* ctx, const uint32_t base_m[4], uint32_t* out_dev, uint32_t count, bool centred, bool reversed) {
    hipLaunchKernelGGL(k_ef_powers, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, base_m[0], base_m[1], base_m[2], base_m[3],
                       out_dev, count, centred ? 1 : 0, reversed ? 1 : 0);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t ef_powers_dev(lurkhip_ctx* ctx, const uint32_t* base_dev, uint32_t* out_dev, uint32_t count, bool centred, bool reversed) {
    hipLaunchKernelGGL(k_ef_powers_dev, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, base_dev, out_dev, count, centred ? 1 : 0, reversed ? 1 : 0);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t scan_ef_column(lurkhip_ctx* ctx, uint32_t* data, size_t stride_words, size_t n) {
    if (n == 0) return LURKHIP_OK;
    const size_t chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    void* totals = nullptr;
    LH_TRY(pool_alloc(ctx, chunks * sizeof(uint4), &totals));
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)chunks), dim3(SCAN_BLOCK), 0, ctx->stream, data, stride_words, n, (uint4*)totals);
    if (chunks > 1) {
        hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, ctx->stream, (uint4*)totals, chunks);
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)chunks), dim3(SCAN_BLOCK), 0, ctx->stream, data, stride_words, n,
                           (const uint4*)totals);
    }
    pool_release(ctx, totals);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}


bool scan_is_one_chunk(size_t n) { return n <= (size_t)SCAN_CHUNK; }
int32_t scan_ef_columns_one_chunk(lurkhip_ctx* ctx, int n_cols, uint32_t* const* data, const uint32_t* strides, const uint32_t* ns) {
    for (int at = 0; at < n_cols; at += SCAN_BATCH) {
        ScanBatch b{};
        const int m = std::min(SCAN_BATCH, n_cols - at);
        for (int k = 0; k < m; k++) {
            b.data[k] = data[at + k];
            b.stride[k] = strides[at + k];
            b.n[k] = ns[at + k];
        }
        hipLaunchKernelGGL(k_scan_local_batch, dim3((unsigned)m), dim3(SCAN_BLOCK), 0, ctx->stream, b);
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t interaction_starts_batch(lurkhip_ctx* ctx, int n, lurkhip_air* const* airs, const bb::ef& alpha, const uint32_t* beta_pows, uint32_t* const* starts) {
    for (int at = 0; at < n;) {
        StartsBatch b{};
        uint32_t blocks = 0;
        for (; at < n && b.n < (uint32_t)STARTS_BATCH; at++) {
            const uint32_t n_inter = airs[at]->air.num_interactions();
            if (n_inter == 0) continue;
            const uint32_t* istat = nullptr;
            LH_TRY(air_programs_dev(ctx, airs[at], nullptr, nullptr, nullptr, false, &istat));
            b.stat[b.n] = istat;
            b.starts[b.n] = starts[at];
            b.first_block[b.n] = blocks;
            blocks += (n_inter + 63) / 64;
            b.n++;
        }
        b.first_block[b.n] = blocks;
        if (blocks) hipLaunchKernelGGL(k_interaction_starts_batch, dim3(blocks), dim3(64), 0, ctx->stream, b, beta_pows, alpha);
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

uint32_t air_beta_pows(const lurkhip_air* a) { return a->max_tuple + 2; }
uint32_t air_num_interactions(const lurkhip_air* a) { return a->air.num_interactions(); }
uint32_t air_total_constraints(const lurkhip_air* a) { return (uint32_t)a->air.constraints.size() + (a->air.permutation_width() - 1) + 3; }

int32_t permutation_trace_impl(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t height, const uint32_t* main_dev, const uint32_t* prep_dev,
                               const bb::ef& alpha, const bb::ef& beta, uint32_t* out_dev, bb::ef* cumulative_sum_m,
                               const uint32_t* shared_beta_pows, uint32_t* shared_starts, uint32_t main_pitch, uint32_t out_pitch, uint32_t* col_live,
                               bool starts_ready, bool defer_scan) {
    LH_ARG(ctx, a->air.prep_width == 0 || prep_dev, "chip has preprocessed columns: pass them");
    if (main_pitch == 0) main_pitch = a->air.width;
    if (out_pitch == 0) out_pitch = 4 * a->air.permutation_width();
    LH_ARG(ctx, main_pitch >= a->air.width && out_pitch >= 4 * a->air.permutation_width() && out_pitch % 4 == 0, "bad row pitch");
    LH_ARG(ctx, height > 0, "empty trace");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const std::vector<uint32_t*>* dparts = nullptr;
    const uint32_t* istat = nullptr;
    LH_TRY(air_programs_dev(ctx, a, nullptr, nullptr, &dparts, false, &istat));
    const uint32_t perm_w = a->air.permutation_width(), batch = 1u << a->air.log_quotient_degree();
    void* pows = nullptr;  // beta powers | interaction start values
    const uint32_t n_pows = a->max_tuple + 2, n_inter = a->air.num_interactions();
    LH_TRY(pool_alloc(ctx, (size_t)n_pows * 32 + (size_t)std::max(n_inter, 1u) * 16, &pows));
    uint32_t* starts = shared_starts ? shared_starts : (uint32_t*)pows + (size_t)n_pows * 8;
    const uint32_t* beta_pows = shared_beta_pows ? shared_beta_pows : (const uint32_t*)pows;
    span_begin(ctx, "perm_rows", 2);
    int32_t s = shared_beta_pows ? LURKHIP_OK : ef_powers(ctx, beta.c, (uint32_t*)pows, n_pows, true);
    if (s == LURKHIP_OK && n_inter && !(shared_starts && starts_ready))
        hipLaunchKernelGGL(k_interaction_starts, dim3((n_inter + 63) / 64), dim3(64), 0, ctx->stream, istat, beta_pows, alpha, starts);
    if (s == LURKHIP_OK) {
        PermArgs pa{};
        std::vector<const std::vector<uint32_t>*> host_parts;
        for (const auto& part : a->prog.interaction_parts) host_parts.push_back(&part);
        const JitKernels jit = jit_of(ctx, a);
        const PartLayout lay = layout_parts(host_parts, *dparts, a->air.width, jit.perm_rows != nullptr && getenv("LURKHIP_JIT_KEEP_LDS_REGS") == nullptr);
        pa.parts = lay.parts;
        pa.main = main_dev;
        pa.prep = prep_dev ? prep_dev : main_dev;
        pa.beta_pows = beta_pows;
        pa.starts = starts;
        pa.n = height;
        pa.w = a->air.width;
        pa.pw = a->air.prep_width;
        pa.perm_w = perm_w;
        pa.batch = batch;
        pa.out = out_dev;
        pa.regs_words = lay.regs_words;
        pa.wp = lay.wp;
        pa.staged = lay.staged ? 1 : 0;
        pa.main_pitch = main_pitch;
        pa.out_pitch = out_pitch;
        pa.col_live = col_live;
        if (jit.perm_rows) {
            void* params[] = {&pa};
            if (hipModuleLaunchKernel(jit.perm_rows, (height + 63) / 64, 1, 1, 64 * lay.parts.n_parts, 1, 1, (unsigned)lay.lds_bytes, ctx->stream,
                                      params, nullptr) != hipSuccess)
                s = set_error(ctx, LURKHIP_ERR_HIP, "launch of the compiled permutation kernel failed");
        } else {
            hipLaunchKernelGGL(k_perm_rows, dim3((height + 63) / 64), dim3(64 * lay.parts.n_parts), lay.lds_bytes, ctx->stream, pa);
        }
        if (hipGetLastError() != hipSuccess) s = set_error(ctx, LURKHIP_ERR_HIP, "k_perm_rows launch failed");
    }
    span_switch(ctx, "perm_rows", "perm_scan", 2);
    if (s == LURKHIP_OK && !defer_scan) s = scan_ef_column(ctx, out_dev + 4 * (perm_w - 1), (size_t)out_pitch, height);
    span_end(ctx, "perm_scan", 2);
    pool_release(ctx, pows);
    if (s == LURKHIP_OK && cumulative_sum_m) {
        LH_HIP(ctx, hipMemcpyAsync(cumulative_sum_m->c, out_dev + (size_t)(height - 1) * out_pitch + (size_t)(perm_w - 1) * 4, 16, hipMemcpyDeviceToHost, ctx->stream));
        LH_HIP(ctx, stream_wait(ctx));
    }
    return s;
}

// The quotient-domain selectors of a (height, quotient degree): a table of the context, written once by a kernel on the
// context's CURRENT stream -- a caller that spreads chips over several streams (prover.hip's side lanes) asks for the tables
// of all its chips on the main stream first, so that the write is ordered before every reader.  nullptr: no memory for the
// table (or LURKHIP_QUOTIENT_SELECTORS_INLINE): the quotient kernel computes the selectors per row.
uint32_t* selector_table_of(lurkhip_ctx* ctx, uint32_t log_n, uint32_t lqd) {
    const auto key = std::make_pair((int)log_n, (int)lqd);
    auto it = ctx->selector_tables.find(key);
    if (it != ctx->selector_tables.end()) return it->second;
    if (getenv("LURKHIP_QUOTIENT_SELECTORS_INLINE") != nullptr) return nullptr;
    const uint32_t log_q = log_n + lqd;
    const NttPlan* plan = nullptr;
    void* tbl = nullptr;
    if (get_ntt_plan(ctx, (int)log_q, &plan) != LURKHIP_OK || hipMalloc(&tbl, ((size_t)12) << log_q) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    SelectorArgs sa{};
    sa.log_q = log_q;
    sa.lqd = lqd;
    sa.g_m = bb::to_monty(bb::GEN);
    sa.wn_inv_m = bb::pow(two_adic_generator_monty((int)log_n), bb::P - 2);
    // Z_H(x) = x^N - 1 on the coset: g^N * (w_Q^N)^i - 1, i mod 2^lqd
    uint32_t gn = sa.g_m;
    for (uint32_t i = 0; i < log_n; i++) gn = bb::mul(gn, gn);
    const uint32_t w_qd = two_adic_generator_monty((int)lqd);
    uint32_t cur = bb::R1;
    for (uint32_t c = 0; c < (1u << lqd) && c < 4; c++) {
        sa.zh[c] = bb::sub(bb::mul(gn, cur), bb::R1);
        cur = bb::mul(cur, w_qd);
    }
    sa.tw = (const uint32_t*)plan->tw_fwd;
    sa.out = (uint32_t*)tbl;
    const uint32_t threads = ((1u << log_q) + 3) / 4;
    hipLaunchKernelGGL(k_selectors, dim3((threads + 255) / 256), dim3(256), 0, ctx->stream, sa);
    ctx->selector_tables.emplace(key, (uint32_t*)tbl);
    return (uint32_t*)tbl;
}

uint32_t air_next_columns(const lurkhip_air* a) {
    uint32_t n = 0;
    for (const lair::Node& nd : a->air.nodes)
        if (nd.kind == lair::N_MAIN_NEXT) n = std::max(n, nd.a + 1);
    return n;
}
bool air_reads_prep_next(const lurkhip_air* a) {
    for (const lair::Node& nd : a->air.nodes)
        if (nd.kind == lair::N_PREP_NEXT) return true;
    return false;
}

int32_t quotient_impl(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t log_n, const uint32_t* main_lde_dev, const uint32_t* prep_lde_dev,
                      const uint32_t* perm_lde_dev, const bb::ef& perm_alpha, const bb::ef& perm_beta, const bb::ef& alpha_m,
                      const bb::ef& cumsum_m, const uint32_t* public_values, uint32_t* out_dev, const uint32_t* shared_beta_pows,
                      const uint32_t* shared_starts, const uint32_t* pitches, bool honest_running_sum, const uint32_t* shared_alpha_pows,
                      const uint32_t* shared_public_m, const uint32_t* cumsum_dev, const QuotientSplit* split, const uint32_t* col_live) {
    LH_ARG(ctx, !cumsum_dev || shared_alpha_pows, "a cumulative sum on the device goes with a shared table of alpha powers");
    LH_ARG(ctx, !split || (honest_running_sum && getenv("LURKHIP_QUOTIENT_READ_NEXT_SUM") == nullptr), "a quotient on a rank's rows cannot read the next row's sums");
    if (split && split->n_rows == 0) return LURKHIP_OK;  // the rank's rows lie outside this chip's quotient domain
    LH_ARG(ctx, a->air.prep_width == 0 || prep_lde_dev, "chip has preprocessed columns: pass their LDE");
    LH_ARG(ctx, a->air.num_public == 0 || public_values, "chip reads public values: pass them");
    const uint32_t lqd = a->air.log_quotient_degree();
    LH_ARG(ctx, lqd <= 2 && log_n + lqd <= (uint32_t)bb::TWO_ADICITY, "unsupported quotient degree / height");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const std::vector<uint32_t*>* dparts = nullptr;
    const std::vector<uint32_t*>* dcons = nullptr;
    const uint32_t* istat = nullptr;
    LH_TRY(air_programs_dev(ctx, a, nullptr, nullptr, &dparts, /*coarse=*/true, &istat, &dcons));
    const uint32_t perm_w = a->air.permutation_width(), batch = 1u << lqd;
    const uint32_t n_batches = perm_w - 1;
    const uint32_t k_total = (uint32_t)a->air.constraints.size() + n_batches + 3;
    const uint32_t np = a->air.num_public;
    const uint32_t *pa = perm_alpha.c, *pb = perm_beta.c, *al = alpha_m.c, *cs = cumsum_m.c;
    const uint32_t n_bp = a->max_tuple + 2;
    void* scratch = nullptr;  // alpha powers | beta powers | interaction start values | public values
    const uint32_t n_inter = a->air.num_interactions();
    const size_t o_bp = (size_t)k_total * 32, o_st = o_bp + (size_t)n_bp * 32, o_pub = o_st + (size_t)std::max(n_inter, 1u) * 16,
                 total = o_pub + std::max<size_t>(np, 1) * 4;
    LH_TRY(pool_alloc(ctx, total, &scratch));
    uint8_t* d = (uint8_t*)scratch;
    // the sinks weigh constraint k with table[K - 1 - k]: powers in natural order give sphinx's Horner folding (first constraint,
    // highest power); the table reversed gives constraint k the power alpha^k (lurkhip_protocol_profile::constraint_alpha_ascending)
    // (a proof's chips share ONE table of alpha powers in natural order -- a chip reads table[K - 1 - k] with its own K -- and one copy
    // of the public values: two launches per chip fewer, which is what a small proof is made of)
    int32_t s = shared_alpha_pows ? LURKHIP_OK : ef_powers(ctx, al, (uint32_t*)d, k_total, true, profile_of(ctx).constraint_alpha_ascending != 0);
    const uint32_t* beta_pows = shared_beta_pows ? shared_beta_pows : (const uint32_t*)(d + o_bp);
    if (s == LURKHIP_OK && !shared_beta_pows) s = ef_powers(ctx, pb, (uint32_t*)(d + o_bp), n_bp, true);
    if (s == LURKHIP_OK && n_inter && !shared_starts)
        hipLaunchKernelGGL(k_interaction_starts, dim3((n_inter + 63) / 64), dim3(64), 0, ctx->stream, istat, beta_pows,
                           bb::ef{{pa[0], pa[1], pa[2], pa[3]}}, (uint32_t*)(d + o_st));
    std::vector<uint32_t> pubm(np);
    if (s == LURKHIP_OK && np && !shared_public_m) {
        for (uint32_t i = 0; i < np; i++) pubm[i] = bb::to_monty(public_values[i] % bb::P);
        s = upload_words(ctx, (uint32_t*)(d + o_pub), pubm.data(), np);  // launch arguments: no host wait in the middle of the stage
    }
    if (s == LURKHIP_OK) {
        QuotientArgs q{};
        std::vector<const std::vector<uint32_t>*> host_parts;
        std::vector<uint32_t*> dev_parts;
        for (size_t j = 0; j < a->prog.constraint_parts.size(); j++) {
            host_parts.push_back(&a->prog.constraint_parts[j]);
            dev_parts.push_back((*dcons)[j]);
        }
        for (size_t j = 0; j < a->prog.interaction_parts_coarse.size(); j++) {
            host_parts.push_back(&a->prog.interaction_parts_coarse[j]);
            dev_parts.push_back((*dparts)[j]);
        }
        const JitKernels jit = jit_of(ctx, a);
        const PartLayout lay = layout_parts(host_parts, dev_parts, a->air.width, jit.quotient != nullptr && getenv("LURKHIP_JIT_KEEP_LDS_REGS") == nullptr);
        q.parts = lay.parts;
        q.n_cons_parts = (uint32_t)a->prog.constraint_parts.size();
        q.n_cons = (uint32_t)a->air.constraints.size();
        q.main = main_lde_dev;
        q.prep = prep_lde_dev ? prep_lde_dev : main_lde_dev;
        q.perm = perm_lde_dev;
        q.pub = shared_public_m ? shared_public_m : (const uint32_t*)(d + o_pub);
        q.alpha_pows = shared_alpha_pows ? shared_alpha_pows : (const uint32_t*)d;
        q.beta_pows = beta_pows;
        q.starts = shared_starts ? shared_starts : (const uint32_t*)(d + o_st);
        q.cumulative_sum = bb::ef{{cs[0], cs[1], cs[2], cs[3]}};
        q.log_n = log_n;
        q.log_q = log_n + lqd;
        q.w = a->air.width;
        q.pw = a->air.prep_width;
        q.perm_w = perm_w;
        q.main_pitch = pitches && pitches[0] ? pitches[0] : q.w;
        q.prep_pitch = prep_lde_dev ? (pitches && pitches[1] ? pitches[1] : q.pw) : q.main_pitch;  // (no preprocessed columns: the pointer aliases the main matrix)
        q.perm_pitch = pitches && pitches[2] ? pitches[2] : perm_w * 4;
        q.batch = batch;
        q.k_total = k_total;
        q.g_m = bb::to_monty(bb::GEN);
        q.wq_m = two_adic_generator_monty((int)q.log_q);
        const uint32_t wn = two_adic_generator_monty((int)log_n);
        q.wn_inv_m = bb::pow(wn, bb::P - 2);
        // Z_H(x) = x^N - 1 on the coset: g^N * (w_Q^N)^i - 1, i mod 2^lqd
        uint32_t gn = q.g_m;
        for (uint32_t i = 0; i < log_n; i++) gn = bb::mul(gn, gn);
        const uint32_t w_qd = two_adic_generator_monty((int)lqd);
        uint32_t cur = bb::R1;
        for (uint32_t c = 0; c < (1u << lqd); c++) {
            q.zh[c] = bb::sub(bb::mul(gn, cur), bb::R1);
            q.zh_inv[c] = bb::pow(q.zh[c], bb::P - 2);
            cur = bb::mul(cur, w_qd);
        }
        // the transition constraint of the running sum on an honest trace: -cumulative_sum / (N w_N) times Z_H(x) (stark_kernels.h)
        q.honest_running_sum = honest_running_sum && getenv("LURKHIP_QUOTIENT_READ_NEXT_SUM") == nullptr ? 1 : 0;
        {
            const uint32_t n_w = bb::mul(bb::to_monty((uint32_t)(((uint64_t)1 << log_n) % bb::P)), wn);
            const uint32_t neg_inv = bb::sub(0u, bb::pow(n_w, bb::P - 2));
            for (int c = 0; c < 4; c++) q.trans_const.c[c] = bb::mul(cs[c], neg_inv);
            q.cumsum_dev = cumsum_dev;
            q.trans_scale = neg_inv;
        }
        q.out = out_dev;
        q.col_live = getenv("LURKHIP_QUOT_COL_LIVE") != nullptr && atoi(getenv("LURKHIP_QUOT_COL_LIVE")) == 0 ? nullptr : col_live;
        q.sel = selector_table_of(ctx, log_n, lqd);
        q.regs_words = lay.regs_words;
        q.wp = lay.wp;
        q.staged = lay.staged ? 1 : 0;
        if (split) {
            q.split = 1;
            q.s_base = split->s_base;
            q.n_rows = split->n_rows;
            q.next_off = split->next_off;
            q.log_rows = split->log_rows;
        }
        const uint32_t rows = split ? split->n_rows : 1u << q.log_q;
        span_begin(ctx, "quotient", 2);
        if (jit.quotient) {
            void* params[] = {&q};
            if (hipModuleLaunchKernel(jit.quotient, (rows + 63) / 64, 1, 1, 64 * lay.parts.n_parts, 1, 1, (unsigned)lay.lds_bytes, ctx->stream, params,
                                      nullptr) != hipSuccess)
                s = set_error(ctx, LURKHIP_ERR_HIP, "launch of the compiled quotient kernel failed");
        } else {
            hipLaunchKernelGGL(k_quotient, dim3((rows + 63) / 64), dim3(64 * lay.parts.n_parts), lay.lds_bytes, ctx->stream, q);
        }
        span_end(ctx, "quotient", 2);
        if (hipGetLastError() != hipSuccess) s = set_error(ctx, LURKHIP_ERR_HIP, "k_quotient launch failed");
    }
    pool_release(ctx, scratch);
    return s;
}


}  // namespace lurkhip

using namespace lurkhip;

namespace {

thread_local std::string g_air_err;

template <class F>
int32_t air_guard(lurkhip_air** out, F&& f) {
    if (!out) return LURKHIP_ERR_INVALID_ARG;
    *out = nullptr;
    try {
        auto* a = new lurkhip_air();
        try {
            a->air = f();
            a->prog = lair::lower_air(a->air);
        } catch (...) {
            delete a;
            throw;
        }
        for (const auto* v : {&a->air.sends, &a->air.receives})
            for (const auto& it : *v) {
                a->tuple_words += 1 + (uint32_t)it.values.size();
                a->max_tuple = std::max<uint32_t>(a->max_tuple, (uint32_t)it.values.size());
            }
        *out = a;
        return LURKHIP_OK;
    } catch (const std::exception& e) {
        g_air_err = e.what();
        return lurkhip::set_error(nullptr, LURKHIP_ERR_UNSUPPORTED, "%s", e.what());
    }
}

}  // namespace

extern "C" {

int32_t lurkhip_air_mem(uint32_t len, lurkhip_air** out) {
    return air_guard(out, [&] {
        lair::mem_index_from_len(len);
        return lair::build_mem_air(len);
    });
}
int32_t lurkhip_air_bytes(lurkhip_air** out) {
    return air_guard(out, [&] { return lair::build_bytes_air(); });
}
int32_t lurkhip_air_entrypoint(uint32_t func_idx, uint32_t num_public_values, lurkhip_air** out) {
    return air_guard(out, [&] { return lair::build_entrypoint_air(func_idx, num_public_values); });
}
int32_t lurkhip_air_poseidon2(int32_t width, lurkhip_air** out) {
    return air_guard(out, [&] { return lair::build_poseidon2_air((uint32_t)width); });
}
int32_t lurkhip_air_from_chip(lair::ChipAir&& air, lurkhip_air** out) {
    return air_guard(out, [&] { return std::move(air); });
}

int32_t lurkhip_air_free(lurkhip_air* a) {
    if (!a) return LURKHIP_OK;
    for (auto& kv : a->dev) {
        (void)hipFree(kv.second.cons);
        (void)hipFree(kv.second.inter_static);
        (void)hipFree(kv.second.inter);
        for (auto* part : kv.second.parts) (void)hipFree(part);
        for (auto* part : kv.second.parts_coarse) (void)hipFree(part);
        jit_release(&kv.second.jit);
    }
    delete a;
    return LURKHIP_OK;
}

int32_t lurkhip_air_info(const lurkhip_air* a, uint32_t* info) {
    if (!a || !info) return LURKHIP_ERR_INVALID_ARG;
    info[0] = a->air.width;
    info[1] = a->air.prep_width;
    info[2] = (uint32_t)a->air.constraints.size();
    info[3] = (uint32_t)a->air.sends.size();
    info[4] = (uint32_t)a->air.receives.size();
    info[5] = a->air.max_constraint_degree();
    info[6] = a->air.log_quotient_degree();
    info[7] = a->air.permutation_width();
    info[8] = a->tuple_words;
    info[9] = a->air.num_public;
    info[10] = a->prog.constraints[airp::H_N_REGS];
    info[11] = a->prog.constraints[airp::H_N_INSTR];
    info[12] = a->prog.interactions[airp::H_N_REGS];
    info[13] = a->prog.interactions[airp::H_N_INSTR];
    info[14] = (uint32_t)a->prog.interaction_parts.size();
    {
        uint32_t sum = 0;
        for (const auto& part : a->prog.interaction_parts) sum += part[airp::H_N_INSTR];
        info[15] = sum;  // instructions over all pieces (common subexpressions are recomputed per piece)
    }
    return LURKHIP_OK;
}

const char* lurkhip_air_name(const lurkhip_air* a) { return a ? a->air.name.c_str() : ""; }

// sizes of each interaction's tuple, sends first then receives; returns the number written
int32_t lurkhip_air_interaction_sizes(const lurkhip_air* a, uint32_t* sizes, uint32_t cap) {
    if (!a || !sizes) return LURKHIP_ERR_INVALID_ARG;
    uint32_t k = 0;
    for (const auto* v : {&a->air.sends, &a->air.receives})
        for (const auto& it : *v) {
            if (k < cap) sizes[k] = (uint32_t)it.values.size();
            k++;
        }
    return (int32_t)k;
}

// Compiles the chip's program pieces to straight-line device code (hiprtc, gfx950) and uses the compiled kernels for its
// permutation traces and quotients on this context's device from now on.  Seconds to tens of seconds of host time for a big
// chip: worth it for traces of 2^17 rows and more.  On failure the chip keeps running on the interpreter and the error says why.
int32_t lurkhip_air_compile(lurkhip_ctx* ctx, lurkhip_air* a) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, a != nullptr, "null air");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_TRY(air_programs_dev(ctx, a, nullptr, nullptr));  // creates the per-device entry
    {
        std::lock_guard<std::mutex> g(a->mu);
        if (a->dev.at(ctx->device).jit.module) return LURKHIP_OK;
    }
    JitKernels k;
    std::string log;
    if (!jit_compile(a->prog, 1u << a->air.log_quotient_degree(), &k, &log)) return set_error(ctx, LURKHIP_ERR_EXEC, "compiling %s failed: %s", a->air.name.c_str(), log.c_str());
    std::lock_guard<std::mutex> g(a->mu);
    a->dev.at(ctx->device).jit = k;
    return LURKHIP_OK;
}

// Generates and compiles the chip's kernels without loading them: no device needed (build / CPU test of the run-time compiler).
// Returns the code object size in bytes, or a negative error with the compiler's message in `log`.
int32_t lurkhip_air_compile_check(const lurkhip_air* a, char* log, uint32_t log_cap) {
    if (!a) return LURKHIP_ERR_INVALID_ARG;
    std::string l;
    const size_t n = jit_compile_only(a->prog, 1u << a->air.log_quotient_degree(), &l);
    if (log && log_cap) {
        const size_t k = std::min<size_t>(l.size(), log_cap - 1);
        memcpy(log, l.data(), k);
        log[k] = 0;
    }
    return n ? (int32_t)std::min<size_t>(n, 0x7fffffff) : LURKHIP_ERR_EXEC;
}

// The lowered register programs (air_program.h), for inspection and tests.  which: 0 constraints, 1 interactions (whole),
// 2 interaction pieces of the permutation-trace kernel, 3 pieces of the quotient kernel, 4 constraint pieces of the quotient
// kernel.  Returns the program's word count
// (copying at most `cap` words), or a negative error for an unknown program.
int32_t lurkhip_air_program(const lurkhip_air* a, int32_t which, uint32_t index, uint32_t* out, uint32_t cap) {
    if (!a) return LURKHIP_ERR_INVALID_ARG;
    const std::vector<uint32_t>* p = nullptr;
    if (which == 0) p = &a->prog.constraints;
    else if (which == 1) p = &a->prog.interactions;
    else if (which == 2 && index < a->prog.interaction_parts.size()) p = &a->prog.interaction_parts[index];
    else if (which == 3 && index < a->prog.interaction_parts_coarse.size()) p = &a->prog.interaction_parts_coarse[index];
    else if (which == 4 && index < a->prog.constraint_parts.size()) p = &a->prog.constraint_parts[index];
    if (!p) return LURKHIP_ERR_INVALID_ARG;
    for (size_t i = 0; i < p->size() && i < cap && out; i++) out[i] = (*p)[i];
    return (int32_t)p->size();
}

int32_t lurkhip_air_eval_rows(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t n_rows, const uint32_t* local, const uint32_t* next,
                              const uint32_t* prep_local, const uint32_t* prep_next, const uint32_t* public_values,
                              const uint32_t* selectors, uint32_t* constraints_out, uint32_t* interactions_out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, a && local && next && selectors && constraints_out && interactions_out, "null argument");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t w = a->air.width, pw = a->air.prep_width, K = (uint32_t)a->air.constraints.size(), T = a->tuple_words;
    const uint32_t np = a->air.num_public;
    LH_ARG(ctx, pw == 0 || (prep_local && prep_next), "chip has preprocessed columns: pass them");
    LH_ARG(ctx, np == 0 || public_values, "chip reads public values: pass them");
    const uint32_t *cp = nullptr, *ip = nullptr;
    LH_TRY(air_programs_dev(ctx, a, &cp, &ip));
    auto words = [&](size_t x) { return std::max<size_t>(x, 4) * 4; };
    size_t o_local = 0, o_next = o_local + words((size_t)n_rows * w), o_pl = o_next + words((size_t)n_rows * w),
           o_pn = o_pl + words((size_t)n_rows * pw), o_pub = o_pn + words((size_t)n_rows * pw), o_sel = o_pub + words(np),
           o_co = o_sel + words((size_t)n_rows * 3), o_io = o_co + words((size_t)n_rows * K), total = o_io + words((size_t)n_rows * T);
    void* dev = nullptr;
    LH_TRY(pool_alloc(ctx, total, &dev));
    uint8_t* d = (uint8_t*)dev;
    std::vector<uint32_t> host(total / 4, 0);
    auto put = [&](size_t off, const uint32_t* src, size_t count) {
        for (size_t i = 0; i < count; i++) host[off / 4 + i] = bb::to_monty(src[i] % bb::P);
    };
    put(o_local, local, (size_t)n_rows * w);
    put(o_next, next, (size_t)n_rows * w);
    if (pw) {
        put(o_pl, prep_local, (size_t)n_rows * pw);
        put(o_pn, prep_next, (size_t)n_rows * pw);
    }
    if (np) put(o_pub, public_values, np);
    put(o_sel, selectors, (size_t)n_rows * 3);
    int32_t s = LURKHIP_OK;
    hipError_t e = hipMemcpyAsync(dev, host.data(), o_co, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = stream_wait(ctx);
    if (e == hipSuccess && n_rows) {
        EvalRowsArgs args{cp, ip, (const uint32_t*)(d + o_local), (const uint32_t*)(d + o_next), (const uint32_t*)(d + o_pl),
                          (const uint32_t*)(d + o_pn), (const uint32_t*)(d + o_pub), (const uint32_t*)(d + o_sel), (uint32_t*)(d + o_co),
                          (uint32_t*)(d + o_io), n_rows, w, pw, K, T};
        const uint32_t n_regs = std::max(a->prog.constraints[airp::H_N_REGS], a->prog.interactions[airp::H_N_REGS]);
        size_t lds = 0;
        int block = vm_block(n_regs, &lds);
        hipLaunchKernelGGL(k_air_eval_rows, dim3((n_rows + block - 1) / block), dim3(block), lds, ctx->stream, args);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(constraints_out, d + o_co, (size_t)n_rows * K * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(interactions_out, d + o_io, (size_t)n_rows * T * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = stream_wait(ctx);
    }
    if (e != hipSuccess) s = set_error(ctx, LURKHIP_ERR_HIP, "air_eval_rows failed: %s", hipGetErrorString(e));
    pool_release(ctx, dev);
    return s;
}

int32_t lurkhip_air_check_trace_dev(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t height, const uint32_t* main_dev,
                                    const uint32_t* prep_dev, const uint32_t* public_values, int64_t* first_bad_row,
                                    int32_t* first_bad_constraint) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, a && main_dev && first_bad_row && first_bad_constraint, "null argument");
    LH_ARG(ctx, a->air.prep_width == 0 || prep_dev, "chip has preprocessed columns: pass them");
    LH_ARG(ctx, a->air.num_public == 0 || public_values, "chip reads public values: pass them");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t* cp = nullptr;
    LH_TRY(air_programs_dev(ctx, a, &cp, nullptr));
    const uint32_t np = a->air.num_public;
    void* scratch = nullptr;
    LH_TRY(pool_alloc(ctx, 16 + (size_t)np * 4, &scratch));
    unsigned long long init = ~0ull;
    std::vector<uint32_t> pubm(np);
    for (uint32_t i = 0; i < np; i++) pubm[i] = bb::to_monty(public_values[i] % bb::P);
    hipError_t e = hipMemcpyAsync(scratch, &init, 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && np) e = hipMemcpyAsync((uint8_t*)scratch + 16, pubm.data(), np * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = stream_wait(ctx);
    if (e == hipSuccess && height) {
        const uint32_t n_regs = a->prog.constraints[airp::H_N_REGS];
        const VmShape shp = vm_shape(n_regs, a->air.width, 1, 65);
        hipLaunchKernelGGL(k_air_check, dim3((height + shp.block - 1) / shp.block), dim3(shp.block), shp.lds, ctx->stream, cp, main_dev,
                           prep_dev ? prep_dev : main_dev, (const uint32_t*)((uint8_t*)scratch + 16), height, a->air.width,
                           a->air.prep_width, (unsigned long long*)scratch, n_regs, shp.wp, shp.staged ? 1 : 0);
        e = hipGetLastError();
    }
    unsigned long long res = ~0ull;
    if (e == hipSuccess) e = hipMemcpyAsync(&res, scratch, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = stream_wait(ctx);
    pool_release(ctx, scratch);
    if (e != hipSuccess) return set_error(ctx, LURKHIP_ERR_HIP, "air_check_trace failed: %s", hipGetErrorString(e));
    if (res == ~0ull) {
        *first_bad_row = -1;
        *first_bad_constraint = -1;
    } else {
        *first_bad_row = (int64_t)(res >> 32);
        *first_bad_constraint = (int32_t)(res & 0xffffffffu);
    }
    return LURKHIP_OK;
}

int32_t lurkhip_permutation_trace_dev(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t height, const uint32_t* main_dev,
                                      const uint32_t* prep_dev, const uint32_t* challenges, uint32_t* out_dev,
                                      uint32_t* cumulative_sum) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, a && main_dev && challenges && out_dev, "null argument");
    bb::ef alpha, beta, cs;
    for (int i = 0; i < 4; i++) {
        alpha.c[i] = bb::to_monty(challenges[i] % bb::P);
        beta.c[i] = bb::to_monty(challenges[4 + i] % bb::P);
    }
    LH_TRY(permutation_trace_impl(ctx, a, height, main_dev, prep_dev, alpha, beta, out_dev, cumulative_sum ? &cs : nullptr));
    if (cumulative_sum)
        for (int i = 0; i < 4; i++) cumulative_sum[i] = bb::from_monty(cs.c[i]);
    return LURKHIP_OK;
}

int32_t lurkhip_quotient_dev(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t log_n, const uint32_t* main_lde_dev,
                             const uint32_t* prep_lde_dev, const uint32_t* perm_lde_dev, const uint32_t* perm_challenges,
                             const uint32_t* alpha, const uint32_t* cumulative_sum, const uint32_t* public_values,
                             uint32_t* out_dev) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, a && main_lde_dev && perm_lde_dev && perm_challenges && alpha && cumulative_sum && out_dev, "null argument");
    bb::ef pa, pb, al, cs;
    for (int i = 0; i < 4; i++) {
        pa.c[i] = bb::to_monty(perm_challenges[i] % bb::P);
        pb.c[i] = bb::to_monty(perm_challenges[4 + i] % bb::P);
        al.c[i] = bb::to_monty(alpha[i] % bb::P);
        cs.c[i] = bb::to_monty(cumulative_sum[i] % bb::P);
    }
    return quotient_impl(ctx, a, log_n, main_lde_dev, prep_lde_dev, perm_lde_dev, pa, pb, al, cs, public_values, out_dev);
}

}  // extern "C"
