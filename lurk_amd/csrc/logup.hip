// Twin of the reference's in-tree LogUp module, /root/reference/src/logup/ (SURVEY.md 8a row L1), on the device.
//
// Replaces: generate_multiplicities_trace (logup/trace.rs:10-50), generate_permutation_trace (logup/trace.rs:53-151),
//           eval_logup_constraints + Interaction::apply (logup/air.rs:11-109).
// The module is dead upstream (`// pub mod logup;`, src/lib.rs:6: never compiled, never called, no tests); the permutation
// argument that actually runs is sphinx's (stark.hip).  This twin states what the code says -- DESIGN.md 3.5 lists the places
// where it disagrees with itself (a powers table one entry short, inverse columns where the comment promises m / d, an inclusive
// running sum under constraints that want an exclusive one, provides-then-requires in the trace against requires-then-provides
// in the AIR) -- and offers the self-consistent variants behind flags.  Parity: bit-exact against the checker's restatement
// (tests/test_logup_gpu.py); nothing upstream can pin it.
//
// Interactions arrive as a flat u32 program (PairColLC, src/air/symbolic/virtual_col.rs:8-13):
//   blob   = n_provides, n_requires, interaction*          (provides first)
//   interaction = has_is_real, [lc], n_values, lc*
//   lc     = n_terms, (kind, index, weight)*, constant     kind 0 identity (row index column), 1 preprocessed, 2 main
// One trace row per lane; the program words come through the scalar cache (uniform addresses).
#include <vector>

#include "babybear.h"
#include "ctx.h"
#include "stark.h"

namespace lurkhip {
namespace {

using bb::ef;
constexpr int LOGUP_BLOCK = 256;

__device__ __forceinline__ uint32_t lc_apply(const uint32_t* __restrict__& pc, uint32_t identity, const uint32_t* __restrict__ prep,
                                             const uint32_t* __restrict__ main) {
    const uint32_t n_terms = *pc++;
    uint32_t acc = 0;
    for (uint32_t t = 0; t < n_terms; t++) {
        const uint32_t kind = pc[0], idx = pc[1], w = pc[2];
        pc += 3;
        const uint32_t v = kind == 0 ? identity : (kind == 1 ? prep[idx] : main[idx]);
        acc = bb::add(acc, bb::mul(v, w));
    }
    return bb::add(acc, *pc++);
}
// skips an lc
__device__ __forceinline__ void lc_skip(const uint32_t* __restrict__& pc) { pc += 3 * pc[0] + 2; }

// d = r + sum_j gamma^j v_j, gamma^0 = 1 (logup/air.rs:79-109); gp = powers of gamma (gp[0] = 1)
__device__ __forceinline__ ef denominator(const uint32_t* __restrict__& pc, uint32_t identity, const uint32_t* __restrict__ prep,
                                          const uint32_t* __restrict__ main, const ef& r, const uint32_t* __restrict__ gp) {
    const uint32_t n_values = *pc++;
    ef d = r;
    for (uint32_t j = 0; j < n_values; j++) {
        const uint32_t v = lc_apply(pc, identity, prep, main);
        const ef g = ef{{gp[4 * j], gp[4 * j + 1], gp[4 * j + 2], gp[4 * j + 3]}};
        d = j == 0 ? bb::ef_add_base(d, v) : bb::ef_add(d, bb::ef_scale(g, v));
    }
    return d;
}

struct LogupArgs {
    const uint32_t* blob;      // device, Montgomery weights / constants
    const uint32_t* identity;  // [h]
    const uint32_t* prep;      // [h][prep_w] or null
    const uint32_t* main;      // [h][main_w]
    const uint32_t* mult;      // [h][n_provides][4]
    const uint32_t* gp;        // gamma powers [max values][4]
    uint32_t* out;             // [h][1 + n_int][4]
    uint32_t height, prep_w, main_w;
    ef z, r;
};

// logup/trace.rs:97-139: cells 1 + k = 1 / d_k (0 where is_real is 0), cell 0 = sum_k m_k / d_k
__global__ __launch_bounds__(LOGUP_BLOCK) void k_logup_rows(LogupArgs a) {
    const uint32_t row = blockIdx.x * LOGUP_BLOCK + threadIdx.x;
    if (row >= a.height) return;
    const uint32_t* __restrict__ pc = a.blob;
    const uint32_t n_prov = pc[0], n_req = pc[1];
    pc += 2;
    const uint32_t n_int = n_prov + n_req;
    const uint32_t identity = a.identity[row];
    const uint32_t* prep = a.prep ? a.prep + (size_t)row * a.prep_w : nullptr;
    const uint32_t* main = a.main + (size_t)row * a.main_w;
    uint32_t* out = a.out + (size_t)row * (1 + n_int) * 4;
    const ef neg_z = bb::ef_sub(ef{{0, 0, 0, 0}}, a.z);
    ef total{{0, 0, 0, 0}};
    for (uint32_t k = 0; k < n_int; k++) {
        const uint32_t has_real = *pc++;
        bool real = true;
        if (has_real) real = lc_apply(pc, identity, prep, main) != 0;
        const ef d = denominator(pc, identity, prep, main, a.r, a.gp);
        ef cell{{0, 0, 0, 0}};
        const bool zero = (d.c[0] | d.c[1] | d.c[2] | d.c[3]) == 0;  // `if inverse.is_zero() { continue; }` (trace.rs:129-131)
        if (real && !zero) {
            cell = bb::ef_inv(d);
            const uint32_t* m = a.mult + ((size_t)row * n_prov + k) * 4;
            const ef mk = k < n_prov ? ef{{m[0], m[1], m[2], m[3]}} : neg_z;
            total = bb::ef_add(total, bb::ef_mul(cell, mk));
        }
        for (int e = 0; e < 4; e++) out[4 * (1 + k) + e] = cell.c[e];
    }
    for (int e = 0; e < 4; e++) out[e] = total.c[e];
}

// inclusive -> exclusive running sum: s'_i = s_i - t_i needs t; simpler: shift the scanned column down by one row
__global__ __launch_bounds__(LOGUP_BLOCK) void k_logup_shift(const uint32_t* __restrict__ scanned, uint32_t* __restrict__ out, uint32_t height,
                                                             uint32_t stride_words) {
    const uint32_t row = blockIdx.x * LOGUP_BLOCK + threadIdx.x;
    if (row >= height) return;
    for (int e = 0; e < 4; e++) out[(size_t)row * stride_words + e] = row == 0 ? 0u : scanned[(size_t)(row - 1) * 4 + e];
}
__global__ __launch_bounds__(LOGUP_BLOCK) void k_logup_gather_col0(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t height,
                                                                   uint32_t stride_words) {
    const uint32_t row = blockIdx.x * LOGUP_BLOCK + threadIdx.x;
    if (row >= height) return;
    for (int e = 0; e < 4; e++) out[(size_t)row * 4 + e] = in[(size_t)row * stride_words + e];
}

// logup/trace.rs:10-50: out[row][col] = sum_j zp[j] * counts[row][j]   (zp[j] = z^(traces[j] + 1), computed by the host)
struct MultArgs {
    const uint32_t* counts;  // [h][n_traces], plain u32 counts
    const uint32_t* zp;      // [n_traces][4]
    uint32_t* out;
    uint32_t height, n_traces, out_stride_words, col;
};
__global__ __launch_bounds__(LOGUP_BLOCK) void k_logup_mult(MultArgs a) {
    const uint32_t row = blockIdx.x * LOGUP_BLOCK + threadIdx.x;
    if (row >= a.height) return;
    ef acc{{0, 0, 0, 0}};
    for (uint32_t j = 0; j < a.n_traces; j++) {
        const uint32_t m = bb::to_monty(a.counts[(size_t)row * a.n_traces + j] % bb::P);
        acc = bb::ef_add(acc, bb::ef_scale(ef{{a.zp[4 * j], a.zp[4 * j + 1], a.zp[4 * j + 2], a.zp[4 * j + 3]}}, m));
    }
    uint32_t* o = a.out + (size_t)row * a.out_stride_words + 4 * a.col;
    for (int e = 0; e < 4; e++) o[e] = acc.c[e];
}

// logup/air.rs:11-77 on row pairs: out[row] = [c_0 .. c_{n_int-1}, first, transition, last] (EF)
struct ConsArgs {
    const uint32_t* blob;
    const uint32_t *perm_local, *perm_next;  // [n][1 + n_int][4]
    const uint32_t* mult;                    // [n][n_prov][4]
    const uint32_t *identity, *prep, *main;
    const uint32_t* gp;
    const uint32_t* sels;  // [n][3]: is_first_row, is_last_row, is_transition
    uint32_t* out;         // [n][n_int + 3][4]
    uint32_t n, prep_w, main_w, air_order;
    ef z, r, final_sum;
};
__global__ __launch_bounds__(LOGUP_BLOCK) void k_logup_constraints(ConsArgs a) {
    const uint32_t row = blockIdx.x * LOGUP_BLOCK + threadIdx.x;
    if (row >= a.n) return;
    const uint32_t n_prov = a.blob[0], n_req = a.blob[1], n_int = n_prov + n_req;
    const uint32_t identity = a.identity[row];
    const uint32_t* prep = a.prep ? a.prep + (size_t)row * a.prep_w : nullptr;
    const uint32_t* main = a.main + (size_t)row * a.main_w;
    const uint32_t* pl = a.perm_local + (size_t)row * (1 + n_int) * 4;
    const uint32_t* pn = a.perm_next + (size_t)row * (1 + n_int) * 4;
    uint32_t* out = a.out + (size_t)row * (n_int + 3) * 4;
    const ef neg_z = bb::ef_sub(ef{{0, 0, 0, 0}}, a.z);
    ef running{{0, 0, 0, 0}};
    // the AIR walks chain(requires, provides) (air.rs:37) but pairs them with chain(provide multiplicities, -z ...) (air.rs:39-43)
    // and with the inverse columns in order; `air_order == 0` walks provides first like the trace generator
    for (uint32_t slot = 0; slot < n_int; slot++) {
        const uint32_t k = a.air_order ? (slot < n_req ? n_prov + slot : slot - n_req) : slot;  // interaction index in the blob
        const uint32_t* __restrict__ pc = a.blob + 2;
        for (uint32_t i = 0; i < k; i++) {  // seek interaction k
            if (*pc++) lc_skip(pc);
            const uint32_t nv = *pc++;
            for (uint32_t j = 0; j < nv; j++) lc_skip(pc);
        }
        const uint32_t has_real = *pc++;
        uint32_t real = bb::R1;
        if (has_real) real = lc_apply(pc, identity, prep, main);
        const ef d = denominator(pc, identity, prep, main, a.r, a.gp);
        const ef inv{{pl[4 * (1 + slot)], pl[4 * (1 + slot) + 1], pl[4 * (1 + slot) + 2], pl[4 * (1 + slot) + 3]}};
        const uint32_t* m = a.mult + ((size_t)row * n_prov + slot) * 4;
        const ef mk = slot < n_prov ? ef{{m[0], m[1], m[2], m[3]}} : neg_z;
        ef c = bb::ef_sub(bb::ef_mul(d, inv), bb::ef_one());
        ef term = bb::ef_mul(mk, inv);
        if (has_real) {
            c = bb::ef_scale(c, real);
            term = bb::ef_scale(term, real);
        }
        running = bb::ef_add(running, term);
        for (int e = 0; e < 4; e++) out[4 * slot + e] = c.c[e];
    }
    const ef partial{{pl[0], pl[1], pl[2], pl[3]}}, partial_next{{pn[0], pn[1], pn[2], pn[3]}};
    const uint32_t* s = a.sels + (size_t)row * 3;
    const ef first = bb::ef_scale(partial, s[0]);
    const ef trans = bb::ef_scale(bb::ef_sub(bb::ef_add(running, partial), partial_next), s[2]);
    const ef last = bb::ef_scale(bb::ef_sub(running, a.final_sum), s[1]);
    for (int e = 0; e < 4; e++) {
        out[4 * n_int + e] = first.c[e];
        out[4 * (n_int + 1) + e] = trans.c[e];
        out[4 * (n_int + 2) + e] = last.c[e];
    }
}

// canonical host words -> Montgomery device copy (pooled); blob weights / constants are converted word by word by the caller
struct Staged {
    lurkhip_ctx* ctx;
    std::vector<void*> blocks;
    ~Staged() {
        for (void* p : blocks) pool_release(ctx, p);
    }
    int32_t up(const uint32_t* host, size_t words, bool to_monty, uint32_t** out) {
        *out = nullptr;
        if (!host || !words) return LURKHIP_OK;
        std::vector<uint32_t> tmp(host, host + words);
        if (to_monty)
            for (auto& v : tmp) v = bb::to_monty(v % bb::P);
        void* d = nullptr;
        LH_TRY(pool_alloc(ctx, words * 4, &d));
        blocks.push_back(d);
        LH_HIP(ctx, hipMemcpyAsync(d, tmp.data(), words * 4, hipMemcpyHostToDevice, ctx->stream));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // tmp goes out of scope
        *out = (uint32_t*)d;
        return LURKHIP_OK;
    }
    int32_t alloc(size_t words, uint32_t** out) {
        void* d = nullptr;
        LH_TRY(pool_alloc(ctx, std::max<size_t>(words, 4) * 4, &d));
        blocks.push_back(d);
        *out = (uint32_t*)d;
        return LURKHIP_OK;
    }
};

// validates the blob, converts weights / constants to Montgomery, reports the counts
int32_t parse_blob(lurkhip_ctx* ctx, const uint32_t* blob, uint64_t words, uint32_t prep_w, uint32_t main_w, std::vector<uint32_t>& out,
                   uint32_t* n_prov, uint32_t* n_req, uint32_t* max_values) {
    LH_ARG(ctx, blob && words >= 2, "logup program too short");
    uint64_t at = 2;
    *n_prov = blob[0], *n_req = blob[1], *max_values = 1;
    LH_ARG(ctx, (uint64_t)*n_prov + *n_req >= 1 && (uint64_t)*n_prov + *n_req <= 4096, "logup program: interaction count");
    out.assign(blob, blob + words);
    auto lc = [&]() -> bool {
        if (at >= words) return false;
        const uint64_t nt = blob[at++];
        if (nt > 4096 || at + 3 * nt + 1 > words) return false;
        for (uint64_t t = 0; t < nt; t++) {
            const uint32_t kind = blob[at], idx = blob[at + 1];
            if (kind > 2 || (kind == 1 && idx >= prep_w) || (kind == 2 && idx >= main_w)) return false;
            out[at + 2] = bb::to_monty(blob[at + 2] % bb::P);
            at += 3;
        }
        out[at] = bb::to_monty(blob[at] % bb::P);
        at++;
        return true;
    };
    for (uint32_t k = 0; k < *n_prov + *n_req; k++) {
        LH_ARG(ctx, at < words, "logup program truncated");
        const uint32_t has_real = blob[at++];
        LH_ARG(ctx, has_real <= 1, "logup program: is_real flag");
        if (has_real) LH_ARG(ctx, lc(), "logup program: malformed is_real form");
        LH_ARG(ctx, at < words, "logup program truncated");
        const uint32_t nv = blob[at++];
        LH_ARG(ctx, nv >= 1 && nv <= 256, "logup program: value count");
        *max_values = std::max(*max_values, nv);
        for (uint32_t j = 0; j < nv; j++) LH_ARG(ctx, lc(), "logup program: malformed value form");
    }
    LH_ARG(ctx, at == words, "logup program: trailing words");
    return LURKHIP_OK;
}

ef ef_of(const uint32_t* canonical) {
    return ef{{bb::to_monty(canonical[0] % bb::P), bb::to_monty(canonical[1] % bb::P), bb::to_monty(canonical[2] % bb::P), bb::to_monty(canonical[3] % bb::P)}};
}

int32_t download(lurkhip_ctx* ctx, const uint32_t* dev, size_t words, uint32_t* host) {
    std::vector<uint32_t> tmp(words);
    LH_HIP(ctx, hipMemcpyAsync(tmp.data(), dev, words * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < words; i++) host[i] = bb::from_monty(tmp[i]);
    return LURKHIP_OK;
}

}  // namespace
}  // namespace lurkhip

using namespace lurkhip;

extern "C" {

int32_t lurkhip_logup_multiplicities(lurkhip_ctx* ctx, uint32_t height, uint32_t n_provides, const uint32_t* n_traces, const uint32_t* traces,
                                     const uint32_t* const* counts, const uint32_t* z, uint32_t* out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, height && n_provides && n_traces && traces && counts && z && out, "null argument");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    Staged st{ctx, {}};
    uint32_t* out_dev = nullptr;
    LH_TRY(st.alloc((size_t)height * n_provides * 4, &out_dev));
    const ef zz = ef_of(z);
    size_t t_at = 0;
    for (uint32_t i = 0; i < n_provides; i++) {
        LH_ARG(ctx, n_traces[i] >= 1 && n_traces[i] <= 1024 && counts[i], "provide %u: trace list", i);
        std::vector<uint32_t> zp((size_t)n_traces[i] * 4);
        for (uint32_t j = 0; j < n_traces[i]; j++) {
            // z^(trace + 1): `challenge_z.powers().skip(1)` indexed by the trace (logup/trace.rs:27-36)
            ef p = zz, acc = bb::ef_one();
            for (uint64_t e = (uint64_t)traces[t_at + j] + 1; e; e >>= 1) {
                if (e & 1) acc = bb::ef_mul(acc, p);
                p = bb::ef_sqr(p);
            }
            for (int e = 0; e < 4; e++) zp[4 * j + e] = acc.c[e];
        }
        t_at += n_traces[i];
        uint32_t *zp_dev = nullptr, *c_dev = nullptr;
        LH_TRY(st.up(zp.data(), zp.size(), false, &zp_dev));
        LH_TRY(st.up(counts[i], (size_t)height * n_traces[i], false, &c_dev));
        MultArgs a{c_dev, zp_dev, out_dev, height, n_traces[i], n_provides * 4, i};
        hipLaunchKernelGGL(k_logup_mult, dim3((height + LOGUP_BLOCK - 1) / LOGUP_BLOCK), dim3(LOGUP_BLOCK), 0, ctx->stream, a);
        LH_HIP(ctx, hipGetLastError());
    }
    return download(ctx, out_dev, (size_t)height * n_provides * 4, out);
}

int32_t lurkhip_logup_permutation_trace(lurkhip_ctx* ctx, uint32_t height, uint32_t prep_width, uint32_t main_width, const uint32_t* identity,
                                        const uint32_t* prep, const uint32_t* main, const uint32_t* multiplicities, const uint32_t* program,
                                        uint64_t program_words, const uint32_t* z, const uint32_t* r, const uint32_t* gamma, int32_t exclusive,
                                        uint32_t* out, uint32_t* sum) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, height && identity && main && program && z && r && gamma && out, "null argument");
    LH_ARG(ctx, prep_width == 0 || prep != nullptr, "preprocessed trace missing");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> blob;
    uint32_t n_prov = 0, n_req = 0, max_values = 1;
    LH_TRY(parse_blob(ctx, program, program_words, prep_width, main_width, blob, &n_prov, &n_req, &max_values));
    LH_ARG(ctx, n_prov == 0 || multiplicities != nullptr, "multiplicities missing");
    const uint32_t n_int = n_prov + n_req, w_out = (1 + n_int) * 4;
    Staged st{ctx, {}};
    LogupArgs a{};
    uint32_t *blob_dev, *id_dev, *prep_dev, *main_dev, *mult_dev, *gp_dev, *out_dev;
    LH_TRY(st.up(blob.data(), blob.size(), false, &blob_dev));
    LH_TRY(st.up(identity, height, true, &id_dev));
    LH_TRY(st.up(prep, (size_t)height * prep_width, true, &prep_dev));
    LH_TRY(st.up(main, (size_t)height * main_width, true, &main_dev));
    LH_TRY(st.up(multiplicities, (size_t)height * n_prov * 4, true, &mult_dev));
    LH_TRY(st.alloc((size_t)max_values * 4, &gp_dev));
    LH_TRY(st.alloc((size_t)height * w_out, &out_dev));
    const ef g = ef_of(gamma);
    LH_TRY(ef_powers(ctx, g.c, gp_dev, max_values));
    a.blob = blob_dev, a.identity = id_dev, a.prep = prep_dev, a.main = main_dev, a.mult = mult_dev, a.gp = gp_dev, a.out = out_dev;
    a.height = height, a.prep_w = prep_width, a.main_w = main_width;
    a.z = ef_of(z), a.r = ef_of(r);
    const dim3 grid((height + LOGUP_BLOCK - 1) / LOGUP_BLOCK);
    hipLaunchKernelGGL(k_logup_rows, grid, dim3(LOGUP_BLOCK), 0, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    // running sum of column 0 (logup/trace.rs:142-148: inclusive); `exclusive`: s_0 = 0, s_{i+1} = s_i + t_i, the sum the constraints want
    uint32_t* col0 = nullptr;
    if (exclusive) {
        LH_TRY(st.alloc((size_t)height * 4, &col0));
        hipLaunchKernelGGL(k_logup_gather_col0, grid, dim3(LOGUP_BLOCK), 0, ctx->stream, out_dev, col0, height, w_out);
        LH_TRY(scan_ef_column(ctx, col0, 4, height));
        hipLaunchKernelGGL(k_logup_shift, grid, dim3(LOGUP_BLOCK), 0, ctx->stream, col0, out_dev, height, w_out);
        LH_HIP(ctx, hipGetLastError());
        if (sum) LH_TRY(download(ctx, col0 + (size_t)(height - 1) * 4, 4, sum));
    } else {
        LH_TRY(scan_ef_column(ctx, out_dev, w_out, height));
        if (sum) LH_TRY(download(ctx, out_dev + (size_t)(height - 1) * w_out, 4, sum));
    }
    return download(ctx, out_dev, (size_t)height * w_out, out);
}

int32_t lurkhip_logup_eval_constraints(lurkhip_ctx* ctx, uint32_t n_rows, uint32_t prep_width, uint32_t main_width, const uint32_t* perm_local,
                                       const uint32_t* perm_next, const uint32_t* multiplicities, const uint32_t* identity, const uint32_t* prep,
                                       const uint32_t* main, const uint32_t* program, uint64_t program_words, const uint32_t* z, const uint32_t* r,
                                       const uint32_t* gamma, const uint32_t* final_sum, const uint32_t* selectors, int32_t air_order, uint32_t* out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, n_rows && perm_local && perm_next && identity && main && program && z && r && gamma && final_sum && selectors && out, "null argument");
    LH_ARG(ctx, prep_width == 0 || prep != nullptr, "preprocessed rows missing");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> blob;
    uint32_t n_prov = 0, n_req = 0, max_values = 1;
    LH_TRY(parse_blob(ctx, program, program_words, prep_width, main_width, blob, &n_prov, &n_req, &max_values));
    LH_ARG(ctx, n_prov == 0 || multiplicities != nullptr, "multiplicities missing");
    const uint32_t n_int = n_prov + n_req;
    Staged st{ctx, {}};
    ConsArgs a{};
    uint32_t *blob_dev, *pl, *pn, *mult_dev, *id_dev, *prep_dev, *main_dev, *gp_dev, *sel_dev, *out_dev;
    LH_TRY(st.up(blob.data(), blob.size(), false, &blob_dev));
    LH_TRY(st.up(perm_local, (size_t)n_rows * (1 + n_int) * 4, true, &pl));
    LH_TRY(st.up(perm_next, (size_t)n_rows * (1 + n_int) * 4, true, &pn));
    LH_TRY(st.up(multiplicities, (size_t)n_rows * n_prov * 4, true, &mult_dev));
    LH_TRY(st.up(identity, n_rows, true, &id_dev));
    LH_TRY(st.up(prep, (size_t)n_rows * prep_width, true, &prep_dev));
    LH_TRY(st.up(main, (size_t)n_rows * main_width, true, &main_dev));
    LH_TRY(st.up(selectors, (size_t)n_rows * 3, true, &sel_dev));
    LH_TRY(st.alloc((size_t)max_values * 4, &gp_dev));
    LH_TRY(st.alloc((size_t)n_rows * (n_int + 3) * 4, &out_dev));
    const ef g = ef_of(gamma);
    LH_TRY(ef_powers(ctx, g.c, gp_dev, max_values));
    a.blob = blob_dev, a.perm_local = pl, a.perm_next = pn, a.mult = mult_dev, a.identity = id_dev, a.prep = prep_dev, a.main = main_dev;
    a.gp = gp_dev, a.sels = sel_dev, a.out = out_dev, a.n = n_rows, a.prep_w = prep_width, a.main_w = main_width, a.air_order = air_order ? 1u : 0u;
    a.z = ef_of(z), a.r = ef_of(r), a.final_sum = ef_of(final_sum);
    hipLaunchKernelGGL(k_logup_constraints, dim3((n_rows + LOGUP_BLOCK - 1) / LOGUP_BLOCK), dim3(LOGUP_BLOCK), 0, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return download(ctx, out_dev, (size_t)n_rows * (n_int + 3) * 4, out);
}

}  // extern "C"
