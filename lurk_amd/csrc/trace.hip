// Trace generation kernels: FuncChip rows (table-driven row interpreter), MemChip, BytesChip.
//
// Replaces (T2/T3/T5/T6 in SURVEY.md 8a):
//   FuncChip::generate_trace + Func/Block/Ctrl/Op::populate_row  /root/reference/src/lair/trace.rs:72-135,145-418
//   MemChip::generate_trace                                        /root/reference/src/lair/memory.rs:30-69
//   BytesChip::{preprocessed_trace,generate_trace}                 /root/reference/src/gadgets/bytes/trace.rs:49-101
//   RequireRecord / ProvideRecord population                       /root/reference/src/air/builder.rs:152-214
//   u64 / depth gadget witnesses                                   /root/reference/src/gadgets/unsigned/{add,mul,cmp,less_than,is_zero}.rs
//
// One trace row per lane.  The reference's per-row work is a bytecode walk with hash-map lookups into
// the query record; here the host has already flattened those lookups into a per-row stream (hints +
// require records, lair/execute.cpp) and resolved everything row-independent into a micro-program
// (lair/emit.cpp), so a lane only runs field arithmetic: inverses for inequality witnesses and
// `count_inv`, products, Poseidon2 / u64 gadget witnesses, selectors.  Lanes of a wave that take the
// same match arm stay converged; rows of different arms diverge as any SIMT interpreter does.
// The row's variable map lives in per-lane scratch (dynamic indices).  Bound: HBM writes,
// width*4 bytes per row plus the streamed inputs.
#include <stdlib.h>

#include "ctx.h"
#include "jit.h"
#include "trace_kernels.h"

namespace {

using namespace lair;
using namespace lurkhip_trace;

template <int CAP>
__device__ __forceinline__ void trace_row(const TraceArgs& a, const uint32_t row_i, RowWriter& w) {
    const uint32_t* __restrict__ prog = a.prog;
    const uint32_t n_in = prog[TH_INPUT], n_out = prog[TH_OUTPUT], n_aux = prog[TH_AUX];
    // nonce for every row, padding included (trace.rs:82-84); the rest of a padding row stays zero
    w.put_int(0, a.nonce_start + row_i);
    if (row_i >= a.n_real) return;

    uint32_t map[CAP];
    uint32_t sp = 0;
    const RowMeta m = a.meta[row_i];
    const uint32_t* hints = a.stream + m.offset;
    const uint32_t* reqs = hints + m.n_hints;
    const uint32_t* dreqs = reqs + 2 * m.n_requires;
    uint32_t h = 0, r = 0, d = 0;
    const uint32_t own_depth = a.depths ? a.depths[row_i] : 0;

    // outputs, provide record, (partial: depth bytes + 2 range-check requires), inputs  (trace.rs:102-131)
    for (uint32_t i = 0; i < n_out; i++) w.put_int(1 + n_in + i, a.outputs[(size_t)row_i * n_out + i]);
    w.push_aux_int(a.provides[2 * row_i]);
    w.push_aux_int(a.provides[2 * row_i + 1]);
    if (prog[TH_PARTIAL]) {
        for (int i = 0; i < 4; i++) w.push_aux_int((own_depth >> (8 * i)) & 0xff);
        for (int i = 0; i < 2; i++) push_require(w, dreqs + 2 * d++);
    }
    for (uint32_t i = 0; i < n_in; i++) {
        uint32_t v = a.args[(size_t)row_i * n_in + i];
        w.put_int(1 + i, v);
        map[sp++] = bb::to_monty(v);
    }

    uint32_t pc = prog[TH_ENTRY];
    for (;;) {
        const uint32_t ins = prog[pc];
        const uint32_t op = ins & 0xff, flag = ins >> 8;
        if (op == T_CONST) {
            map[sp++] = prog[pc + 1];
            pc += 2;
        } else if (op == T_ADD) {
            map[sp++] = bb::add(map[prog[pc + 1]], map[prog[pc + 2]]);
            pc += 3;
        } else if (op == T_SUB) {
            map[sp++] = bb::sub(map[prog[pc + 1]], map[prog[pc + 2]]);
            pc += 3;
        } else if (op == T_MUL) {
            uint32_t f = bb::mul(map[prog[pc + 1]], map[prog[pc + 2]]);
            map[sp++] = f;
            if (flag) w.push_aux(f);
            pc += 3;
        } else if (op == T_INV) {
            uint32_t f = bb::inv(map[prog[pc + 1]]);
            map[sp++] = f;
            if (flag) w.push_aux(f);
            pc += 2;
        } else if (op == T_NOT) {
            uint32_t x = map[prog[pc + 1]];
            uint32_t dinv = x ? bb::inv(x) : 0u;
            uint32_t f = x ? 0u : bb::R1;
            map[sp++] = f;
            if (flag) {
                w.push_aux(dinv);
                w.push_aux(f);
            }
            pc += 2;
        } else if (op == T_ASSERT_NE) {
            // inverse of the first non-zero difference, zeros elsewhere (trace.rs:218-233)
            const uint32_t n = flag;
            bool found = false;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t diff = bb::sub(map[prog[pc + 1 + i]], map[prog[pc + 1 + n + i]]);
                if (!found && diff != 0) {
                    w.push_aux(bb::inv(diff));
                    found = true;
                } else {
                    w.push_aux(0u);
                }
            }
            pc += 1 + 2 * n;
        } else if (op == T_CONTAINS) {
            const uint32_t n = flag;
            const uint32_t b = map[prog[pc + 1]];
            uint32_t acc = bb::sub(map[prog[pc + 2]], b);
            for (uint32_t i = 1; i < n; i++) {
                acc = bb::mul(acc, bb::sub(map[prog[pc + 2 + i]], b));
                w.push_aux(acc);
            }
            pc += 2 + n;
        } else if (op == T_CALL) {
            const uint32_t n = prog[pc + 1];
            for (uint32_t i = 0; i < n; i++) {
                uint32_t v = hints[h++];
                map[sp++] = bb::to_monty(v);
                w.push_aux_int(v);
            }
            push_require(w, reqs + 2 * r++);
            if (flag) {
                // dependency provenance (trace.rs:235-254): callee depth bytes, DepthLessThan, 1 require
                const uint32_t cd = hints[h++];
                for (int i = 0; i < 4; i++) w.push_aux_int((cd >> (8 * i)) & 0xff);
                push_depth_less_than(w, cd, own_depth);
                push_require(w, dreqs + 2 * d++);
            }
            pc += 2;
        } else if (op == T_STORE) {
            uint32_t v = hints[h++];
            map[sp++] = bb::to_monty(v);
            w.push_aux_int(v);
            push_require(w, reqs + 2 * r++);
            pc += 1;
        } else if (op == T_LOAD) {
            const uint32_t n = prog[pc + 1];
            for (uint32_t i = 0; i < n; i++) {
                uint32_t v = hints[h++];
                map[sp++] = bb::to_monty(v);
                w.push_aux_int(v);
            }
            push_require(w, reqs + 2 * r++);
            pc += 2;
        } else if (op == T_EXTERN) {
            const uint32_t kind = prog[pc + 1], nin = prog[pc + 2], wit = prog[pc + 3], nreq = prog[pc + 4], nret = prog[pc + 5];
            const uint32_t* ins_v = prog + pc + 6;
            uint32_t in[EXTERN_MAX_IO], out[EXTERN_MAX_IO];
            for (uint32_t i = 0; i < nin && i < (uint32_t)EXTERN_MAX_IO; i++) in[i] = map[ins_v[i]];
            extern_op(w, kind, in, out, wit);
            for (uint32_t i = 0; i < nret && i < (uint32_t)EXTERN_MAX_IO; i++) map[sp++] = out[i];
            for (uint32_t i = 0; i < nreq; i++) push_require(w, reqs + 2 * r++);
            pc += 6 + nin;
        } else if (op == T_RANGE_U8) {
            const uint32_t n = prog[pc + 1];
            for (uint32_t i = 0; i < n; i++) push_require(w, reqs + 2 * r++);
            pc += 2;
        } else if (op == T_RETURN) {
            w.put_int(1 + n_in + n_out + n_aux + prog[pc + 1], 1u);
            return;
        } else if (op == T_CHOOSE) {
            const uint32_t v = map[prog[pc + 1]], n = prog[pc + 2];
            uint32_t tgt = prog[pc + 3];
            for (uint32_t i = 0; i < n; i++)
                if (prog[pc + 4 + 2 * i] == v) {
                    tgt = prog[pc + 4 + 2 * i + 1];
                    break;
                }
            pc = tgt;
        } else if (op == T_CHOOSE_MANY) {
            const uint32_t nv = prog[pc + 1], n = prog[pc + 2];
            uint32_t tgt = prog[pc + 3];
            const uint32_t* vars = prog + pc + 4;
            const uint32_t* table = vars + nv;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t* e = table + (size_t)i * (nv + 1);
                bool eq = true;
                for (uint32_t k = 0; k < nv; k++) eq = eq && e[k] == map[vars[k]];
                if (eq) {
                    tgt = e[nv];
                    break;
                }
            }
            pc = tgt;
        } else {
            return;  // corrupt program: host validates before launch
        }
    }
}

template <int CAP, bool STAGED>
__global__ __launch_bounds__(TBLOCK) void k_trace_func(TraceArgs a) {
    trace_kernel_body<STAGED>(a, [](const TraceArgs& aa, uint32_t row_i, RowWriter& w) { trace_row<CAP>(aa, row_i, w); });
}

// ---- MemChip (memory.rs:30-69): [is_real = 1, ptr = i + 1, last_nonce, last_count, values...] -----------
// (row0: the table's row the launch starts at -- a rank's block of rows of one shard proved by several ranks, lurkhip_func_trace_run_rows)
__global__ void k_trace_mem(const uint32_t* __restrict__ values, const uint32_t* __restrict__ provides, uint32_t len,
                            uint32_t n_real, uint32_t height, uint32_t* __restrict__ out, int canonical, uint32_t pitch, uint32_t row0) {
    const uint32_t width = 4 + len;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)height * width) return;
    uint32_t row = (uint32_t)(e / width), c = (uint32_t)(e - (size_t)row * width);
    uint32_t v = 0;
    if (row < n_real) {
        if (c == 0) v = 1;
        else if (c == 1) v = row0 + row + 1;
        else if (c < 4) v = provides[2 * (size_t)row + (c - 2)];
        else v = values[(size_t)row * len + (c - 4)];
    }
    out[(size_t)row * pitch + c] = canonical ? v : bb::to_monty(v);
}

// ---- BytesChip main trace (bytes/trace.rs:75-101): [is_real, 6 x (last_nonce, last_count)] ----------------
// records: [65536][12] (range_u8, range_u16, less_than, and, xor, or) x (nonce, count); all-zero rows = never required
__global__ void k_trace_bytes(const uint32_t* __restrict__ records, int is_real, uint32_t* __restrict__ out, int canonical, uint32_t pitch, uint32_t rows) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)rows * 13) return;
    uint32_t row = (uint32_t)(e / 13), c = (uint32_t)(e - (size_t)row * 13);
    uint32_t v = 0;
    if (is_real) v = c == 0 ? 1u : records[(size_t)row * 12 + (c - 1)];
    out[(size_t)row * pitch + c] = canonical ? v : bb::to_monty(v);
}

// ---- BytesChip preprocessed trace (bytes/trace.rs:49-72): [i1, i2, i1 < i2, and, xor, or] -----------------
__global__ void k_trace_bytes_preprocessed(uint32_t* __restrict__ out, int canonical) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= 65536) return;
    uint32_t i1 = row & 0xff, i2 = row >> 8;
    uint32_t v[6] = {i1, i2, i1 < i2 ? 1u : 0u, i1 & i2, i1 ^ i2, i1 | i2};
    for (int k = 0; k < 6; k++) out[(size_t)row * 6 + k] = canonical ? v[k] : bb::to_monty(v[k]);
}

}  // namespace

// The three trace generators with a row pitch (0: the trace's width): the prover's own traces are column ranges of aligned group
// buffers (lurkhip_trace_group_layout); the dense public entry points below pass 0.
namespace lurkhip {
int32_t trace_func_dev_pitched(lurkhip_ctx* ctx, const uint32_t* program_dev, const uint32_t* program_host_header,
                               uint32_t n_real, uint32_t height, uint32_t nonce_start, const uint32_t* args_dev,
                               const uint32_t* outputs_dev, const uint32_t* provides_dev, const uint32_t* depths_dev,
                               const void* meta_dev, const uint32_t* stream_dev, uint32_t* out_dev, int32_t repr, uint32_t out_pitch) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, program_dev && program_host_header && out_dev, "null argument");
    LH_ARG(ctx, program_host_header[TH_MAGIC] == TRACE_PROGRAM_MAGIC, "bad trace program header");
    LH_ARG(ctx, n_real <= height, "n_real exceeds height");
    LH_ARG(ctx, repr == LURKHIP_REPR_CANONICAL || repr == LURKHIP_REPR_MONTY, "bad repr %d", repr);
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t width = program_host_header[TH_WIDTH];
    const uint32_t max_vars = program_host_header[TH_MAX_VARS];
    LH_ARG(ctx, max_vars <= 4096, "function needs %u variables, more than the kernel's map capacity", max_vars);
    if (out_pitch == 0) out_pitch = width;
    LH_ARG(ctx, out_pitch >= width, "row pitch %u below the trace's width %u", out_pitch, width);
    if (height == 0) return LURKHIP_OK;
    const size_t tile_words = (size_t)TBLOCK * width + (((size_t)TBLOCK * width) >> 5) + 1;
    const bool staged = tile_words * 4 <= 64 * 1024;
    if (!staged) LH_HIP(ctx, hipMemset2DAsync(out_dev, (size_t)out_pitch * 4, 0, (size_t)width * 4, height, ctx->stream));
    TraceArgs a{program_dev, args_dev, outputs_dev, provides_dev, depths_dev, (const RowMeta*)meta_dev, stream_dev, out_dev,
                n_real, height, nonce_start, repr == LURKHIP_REPR_CANONICAL, out_pitch};
    dim3 grid((height + TBLOCK - 1) / TBLOCK), block(TBLOCK);
    lurkhip::span_begin(ctx, "trace_func", 2);
    const size_t lds = staged ? tile_words * 4 : 0;
    // the function's compiled row kernel, when lurkhip_trace_compile has produced one for this program (named by its hash)
    const uint64_t prog_hash = (uint64_t)program_host_header[TH_HASH_LO] | ((uint64_t)program_host_header[TH_HASH_HI] << 32);
    const lurkhip::TraceJitKernels jit = prog_hash && getenv("LURKHIP_TRACE_INTERPRET") == nullptr ? lurkhip::trace_jit_lookup(ctx->device, prog_hash) : lurkhip::TraceJitKernels{};
    if (jit.module) {
        void* params[] = {&a};
        const hipError_t le = hipModuleLaunchKernel(staged ? jit.staged : jit.flat, grid.x, 1, 1, TBLOCK, 1, 1, (unsigned)lds, ctx->stream, params, nullptr);
        lurkhip::span_end(ctx, "trace_func", 2);
        if (le != hipSuccess) return lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "launch of the compiled trace kernel failed: %s", hipGetErrorString(le));
        return LURKHIP_OK;
    }
#define LH_TRACE_LAUNCH(CAP)                                                                             \
    do {                                                                                                 \
        if (staged) hipLaunchKernelGGL((k_trace_func<CAP, true>), grid, block, lds, ctx->stream, a);     \
        else hipLaunchKernelGGL((k_trace_func<CAP, false>), grid, block, 0, ctx->stream, a);             \
    } while (0)
    if (max_vars <= 64) LH_TRACE_LAUNCH(64);
    else if (max_vars <= 256) LH_TRACE_LAUNCH(256);
    else if (max_vars <= 1024) LH_TRACE_LAUNCH(1024);
    else LH_TRACE_LAUNCH(4096);
#undef LH_TRACE_LAUNCH
    lurkhip::span_end(ctx, "trace_func", 2);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t trace_mem_dev_pitched(lurkhip_ctx* ctx, uint32_t len, uint32_t n_real, uint32_t height, const uint32_t* values_dev,
                              const uint32_t* provides_dev, uint32_t* out_dev, int32_t repr, uint32_t out_pitch, uint32_t row0) {
    LH_CHECK_CTX(ctx);
    if (out_pitch == 0) out_pitch = 4 + len;
    LH_ARG(ctx, out_pitch >= 4 + len, "row pitch %u below the trace's width %u", out_pitch, 4 + len);
    LH_ARG(ctx, out_dev && (n_real == 0 || (values_dev && provides_dev)), "null argument");
    LH_ARG(ctx, n_real <= height, "n_real exceeds height");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    size_t total = (size_t)height * (4 + len);
    if (total == 0) return LURKHIP_OK;
    hipLaunchKernelGGL(k_trace_mem, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, values_dev, provides_dev, len,
                       n_real, height, out_dev, repr == LURKHIP_REPR_CANONICAL, out_pitch, row0);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t trace_bytes_dev_pitched(lurkhip_ctx* ctx, const uint32_t* records_dev, int32_t is_real, uint32_t* out_dev, int32_t repr, uint32_t out_pitch, uint32_t rows) {
    LH_CHECK_CTX(ctx);
    if (out_pitch == 0) out_pitch = 13;
    LH_ARG(ctx, out_pitch >= 13, "row pitch %u below the byte chip's width", out_pitch);
    LH_ARG(ctx, out_dev && (!is_real || records_dev), "null argument");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    size_t total = (size_t)rows * 13;
    if (total == 0) return LURKHIP_OK;
    hipLaunchKernelGGL(k_trace_bytes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, records_dev, is_real, out_dev,
                       repr == LURKHIP_REPR_CANONICAL, out_pitch, rows);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}
}  // namespace lurkhip

extern "C" {

int32_t lurkhip_trace_func_dev(lurkhip_ctx* ctx, const uint32_t* program_dev, const uint32_t* program_host_header,
                               uint32_t n_real, uint32_t height, uint32_t nonce_start, const uint32_t* args_dev,
                               const uint32_t* outputs_dev, const uint32_t* provides_dev, const uint32_t* depths_dev,
                               const void* meta_dev, const uint32_t* stream_dev, uint32_t* out_dev, int32_t repr) {
    return lurkhip::trace_func_dev_pitched(ctx, program_dev, program_host_header, n_real, height, nonce_start, args_dev, outputs_dev, provides_dev,
                                           depths_dev, meta_dev, stream_dev, out_dev, repr, 0);
}

int32_t lurkhip_trace_mem_dev(lurkhip_ctx* ctx, uint32_t len, uint32_t n_real, uint32_t height, const uint32_t* values_dev,
                              const uint32_t* provides_dev, uint32_t* out_dev, int32_t repr) {
    return lurkhip::trace_mem_dev_pitched(ctx, len, n_real, height, values_dev, provides_dev, out_dev, repr, 0, 0);
}

int32_t lurkhip_trace_bytes_dev(lurkhip_ctx* ctx, const uint32_t* records_dev, int32_t is_real, uint32_t* out_dev, int32_t repr) {
    return lurkhip::trace_bytes_dev_pitched(ctx, records_dev, is_real, out_dev, repr, 0, 65536);
}

int32_t lurkhip_trace_bytes_preprocessed_dev(lurkhip_ctx* ctx, uint32_t* out_dev, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out_dev != nullptr, "null argument");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_trace_bytes_preprocessed, dim3(256), dim3(256), 0, ctx->stream, out_dev, repr == LURKHIP_REPR_CANONICAL);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

}  // extern "C"
