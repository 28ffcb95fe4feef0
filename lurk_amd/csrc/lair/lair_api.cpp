// C ABI over the Lair host: toplevel construction, execution, layout queries and the
// generate_trace entry points that feed the device kernels.
//
// These are the seams a Rust shim would replace:
//   Toplevel::{new, execute_by_name}        /root/reference/src/lair/toplevel.rs:28-50, execute.rs:375-417
//   FuncChip::{from_name, width, generate_trace}   /root/reference/src/lair/func_chip.rs:34-80, trace.rs:72-135
//   MemChip / BytesChip / Entrypoint generate_trace  /root/reference/src/lair/lair_chip.rs:96-120
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "../ctx.h"
#include "../jit.h"
#include "../stark.h"
#include "air.h"
#include "lair.h"

struct lurkhip_toplevel {
    lair::Toplevel t;
    std::vector<lair::LayoutSizes> layouts;
    std::vector<std::vector<uint32_t>> programs;  // lazily built
};

struct lurkhip_record {
    const lurkhip_toplevel* top;
    lair::QueryRecord q;
    explicit lurkhip_record(const lurkhip_toplevel* t) : top(t), q(t->t) {}
};

extern "C" int32_t lurkhip_trace_func_dev(lurkhip_ctx* ctx, const uint32_t* program_dev, const uint32_t* program_host_header,
                                          uint32_t n_real, uint32_t height, uint32_t nonce_start, const uint32_t* args_dev,
                                          const uint32_t* outputs_dev, const uint32_t* provides_dev, const uint32_t* depths_dev,
                                          const void* meta_dev, const uint32_t* stream_dev, uint32_t* out_dev, int32_t repr);
extern "C" int32_t lurkhip_trace_mem_dev(lurkhip_ctx* ctx, uint32_t len, uint32_t n_real, uint32_t height,
                                         const uint32_t* values_dev, const uint32_t* provides_dev, uint32_t* out_dev, int32_t repr);
extern "C" int32_t lurkhip_trace_bytes_dev(lurkhip_ctx* ctx, const uint32_t* records_dev, int32_t is_real, uint32_t* out_dev,
                                           int32_t repr);
namespace lurkhip {  // trace.hip: the generators with a row pitch (0: dense)
int32_t trace_func_dev_pitched(lurkhip_ctx* ctx, const uint32_t* program_dev, const uint32_t* program_host_header, uint32_t n_real, uint32_t height,
                               uint32_t nonce_start, const uint32_t* args_dev, const uint32_t* outputs_dev, const uint32_t* provides_dev,
                               const uint32_t* depths_dev, const void* meta_dev, const uint32_t* stream_dev, uint32_t* out_dev, int32_t repr,
                               uint32_t out_pitch);
int32_t trace_mem_dev_pitched(lurkhip_ctx* ctx, uint32_t len, uint32_t n_real, uint32_t height, const uint32_t* values_dev, const uint32_t* provides_dev,
                              uint32_t* out_dev, int32_t repr, uint32_t out_pitch, uint32_t row0);
int32_t trace_bytes_dev_pitched(lurkhip_ctx* ctx, const uint32_t* records_dev, int32_t is_real, uint32_t* out_dev, int32_t repr, uint32_t out_pitch, uint32_t rows);
// commit.hip: the layout lurkhip_trace_group_layout reports
void plan_source_groups(int n, const uint32_t* log_heights, const uint32_t* widths, uint32_t* pitch, uint32_t* col_start, int32_t* group, int32_t* n_groups);
}  // namespace lurkhip

namespace {

thread_local std::string g_err;

int32_t fail(lurkhip_ctx* ctx, int32_t code, const std::string& msg) {
    g_err = msg;
    return lurkhip::set_error(ctx, code, "%s", msg.c_str());
}

template <class F>
int32_t guarded(lurkhip_ctx* ctx, F&& f) {
    try {
        return f();
    } catch (const lair::ParseError& e) {
        return fail(ctx, LURKHIP_ERR_PARSE, e.what());
    } catch (const lair::ExecError& e) {
        return fail(ctx, LURKHIP_ERR_EXEC, e.what());
    } catch (const std::exception& e) {
        return fail(ctx, LURKHIP_ERR_EXEC, std::string("internal error: ") + e.what());
    }
}

// cores this process may actually use: the affinity mask capped by the cgroup CPU quota (cpu.max).  A container on a 256-thread
// host often has a quota of a dozen cores: one worker per hardware thread would spend its time being throttled.
uint32_t usable_cores() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = atol(quota) / period;
            if (q >= 1) n = std::min<uint32_t>(n, (uint32_t)q);
        }
        fclose(f);
    }
    return n;
}

uint32_t next_pow2(uint32_t n) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

const std::vector<uint32_t>& program_of(lurkhip_toplevel* top, uint32_t idx) {
    if (top->programs.size() != top->t.funcs.size()) top->programs.resize(top->t.funcs.size());
    if (top->programs[idx].empty()) top->programs[idx] = lair::build_trace_program(top->t, top->t.funcs[idx], nullptr);
    return top->programs[idx];
}

}  // namespace

extern "C" {

int32_t lurkhip_toplevel_new(const char* source, int32_t with_lurk_chips, lurkhip_toplevel** out) {
    if (!source || !out) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    return guarded(nullptr, [&]() -> int32_t {
        auto funcs = lair::parse_funcs(source);
        auto* top = new lurkhip_toplevel();
        try {
            top->t = lair::Toplevel::build(funcs, with_lurk_chips ? lair::lurk_chip_map() : std::vector<lair::Chip>());
            for (const auto& f : top->t.funcs) top->layouts.push_back(lair::compute_layout_sizes(top->t, f));
        } catch (...) {
            delete top;
            throw;
        }
        *out = top;
        return LURKHIP_OK;
    });
}

int32_t lurkhip_toplevel_from_bytecode(const uint32_t* blob, uint64_t n_words, lurkhip_toplevel** out) {
    if (!blob || !out) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    return guarded(nullptr, [&]() -> int32_t {
        auto top = std::make_unique<lurkhip_toplevel>();
        top->t = lair::toplevel_from_bytecode(blob, (size_t)n_words);
        for (const auto& f : top->t.funcs) top->layouts.push_back(lair::compute_layout_sizes(top->t, f));
        *out = top.release();
        return LURKHIP_OK;
    });
}

int64_t lurkhip_toplevel_to_bytecode(const lurkhip_toplevel* top, uint32_t* out, uint64_t capacity_words) {
    if (!top) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "null argument");
    int64_t n = 0;
    int32_t st = guarded(nullptr, [&]() -> int32_t {
        std::vector<uint32_t> w = lair::toplevel_to_bytecode(top->t);
        n = (int64_t)w.size();
        if (out && capacity_words >= w.size()) memcpy(out, w.data(), w.size() * 4);
        return LURKHIP_OK;
    });
    return st == LURKHIP_OK ? n : (int64_t)st;
}

int32_t lurkhip_toplevel_free(lurkhip_toplevel* top) {
    delete top;
    return LURKHIP_OK;
}

int32_t lurkhip_toplevel_num_funcs(const lurkhip_toplevel* top) { return top ? (int32_t)top->t.funcs.size() : LURKHIP_ERR_INVALID_ARG; }

int32_t lurkhip_toplevel_func_index(const lurkhip_toplevel* top, const char* name) {
    if (!top || !name) return LURKHIP_ERR_INVALID_ARG;
    auto it = top->t.func_index.find(name);
    return it == top->t.func_index.end() ? LURKHIP_ERR_INVALID_ARG : (int32_t)it->second;
}

// info[0..8) = input_size, output_size, partial, invertible, layout nonce, input, output, aux, sel
int32_t lurkhip_toplevel_func_info(const lurkhip_toplevel* top, int32_t func_idx, uint32_t* info) {
    if (!top || !info || func_idx < 0 || (size_t)func_idx >= top->t.funcs.size()) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "bad func index");
    const auto& f = top->t.funcs[func_idx];
    const auto& l = top->layouts[func_idx];
    uint32_t v[9] = {f.input_size, f.output_size, f.partial, f.invertible, l.nonce, l.input, l.output, l.aux, l.sel};
    memcpy(info, v, sizeof v);
    return LURKHIP_OK;
}

// AIR of one function's chip (lair/air.rs:158-552), lowered for the device VM
int32_t lurkhip_air_func(const lurkhip_toplevel* top, int32_t func_idx, lurkhip_air** out) {
    if (!top || !out || func_idx < 0 || (size_t)func_idx >= top->t.funcs.size()) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "bad func index");
    *out = nullptr;
    return guarded(nullptr, [&]() -> int32_t { return lurkhip_air_from_chip(lair::build_func_air(top->t, top->t.funcs[func_idx]), out); });
}

int32_t lurkhip_record_new(const lurkhip_toplevel* top, lurkhip_record** out) {
    if (!top || !out) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "null argument");
    *out = new lurkhip_record(top);
    return LURKHIP_OK;
}

int32_t lurkhip_record_free(lurkhip_record* r) {
    delete r;
    return LURKHIP_OK;
}

int32_t lurkhip_record_clean(lurkhip_record* r) {
    if (!r) return LURKHIP_ERR_INVALID_ARG;
    r->q.clean();
    return LURKHIP_OK;
}

// Toplevel::execute (execute.rs:375-392); `out` must hold output_size values
int32_t lurkhip_execute(lurkhip_record* r, int32_t func_idx, const uint32_t* args, uint32_t n_args, uint32_t* out) {
    if (!r || func_idx < 0 || (size_t)func_idx >= r->top->t.funcs.size()) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "bad func index");
    return guarded(nullptr, [&]() -> int32_t {
        lair::List a(args, args + n_args);
        for (uint32_t v : a)
            if (v >= lair::P) throw lair::ExecError("argument is not a canonical field element");
        lair::List o = lair::execute(r->top->t, r->top->t.funcs[func_idx], a, r->q);
        if (out && !o.empty()) memcpy(out, o.data(), o.size() * 4);
        return LURKHIP_OK;
    });
}

int32_t lurkhip_record_inject_inv_query(lurkhip_record* r, int32_t func_idx, const uint32_t* inp, uint32_t n_inp,
                                        const uint32_t* out, uint32_t n_out) {
    if (!r) return LURKHIP_ERR_INVALID_ARG;
    return guarded(nullptr, [&]() -> int32_t {
        for (uint32_t i = 0; i < n_inp; i++)
            if (inp[i] >= lair::P) throw lair::ExecError("injected preimage is not made of canonical field elements");
        for (uint32_t i = 0; i < n_out; i++)
            if (out[i] >= lair::P) throw lair::ExecError("injected image is not made of canonical field elements");
        r->q.inject_inv_query((uint32_t)func_idx, lair::List(inp, inp + n_inp), lair::List(out, out + n_out));
        return LURKHIP_OK;
    });
}

// number of queries of a func (kind 0) or mem table of length `index` (kind 1); kind 2: public values length
int64_t lurkhip_record_count(const lurkhip_record* r, int32_t kind, int32_t index) {
    if (!r) return LURKHIP_ERR_INVALID_ARG;
    try {
        if (kind == 0) return (int64_t)r->q.func_queries.at(index).size();
        if (kind == 1) return (int64_t)r->q.mem_queries.at(lair::mem_index_from_len((uint32_t)index)).size();
        if (kind == 2) return r->q.has_public_values ? (int64_t)r->q.public_values.size() : -1;
        if (kind == 3) return (int64_t)r->q.bytes.size();
        if (kind == 4) return (int64_t)r->q.emitted.size();
    } catch (...) {
    }
    return LURKHIP_ERR_INVALID_ARG;
}

int32_t lurkhip_record_public_values(const lurkhip_record* r, uint32_t* out) {
    if (!r || !out || !r->q.has_public_values) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "Public values not set");
    memcpy(out, r->q.public_values.data(), r->q.public_values.size() * 4);
    return LURKHIP_OK;
}

int64_t lurkhip_record_num_shards(const lurkhip_record* r, uint32_t max_shard_size) {
    if (!r || max_shard_size == 0) return LURKHIP_ERR_INVALID_ARG;
    return (int64_t)lair::num_shards(r->q, max_shard_size);
}

// Shape of the FuncChip trace of `func_idx` for one shard: n_real rows, padded height, width.
int32_t lurkhip_func_trace_shape(const lurkhip_record* r, int32_t func_idx, uint32_t shard_index, uint32_t max_shard_size,
                                 uint32_t* n_real, uint32_t* height, uint32_t* width) {
    if (!r || func_idx < 0 || (size_t)func_idx >= r->top->t.funcs.size()) return fail(nullptr, LURKHIP_ERR_INVALID_ARG, "bad func index");
    auto [s, e] = lair::shard_range(r->q.func_queries[func_idx].size(), shard_index, max_shard_size);
    uint32_t n = (uint32_t)(e - s);
    if (n_real) *n_real = n;
    if (height) *height = next_pow2(n);  // `0.next_power_of_two()` is 1 in Rust (trace.rs:79)
    if (width) *width = r->top->layouts[func_idx].total();
    return LURKHIP_OK;
}

// Compiles the function's trace micro-program to a straight-line row kernel (trace_jit.cpp, hiprtc) and loads it on the
// context's device: every later trace of that function on that device (lurkhip_trace_func_dev recognises the program by the
// hash in its header) runs the compiled kernel instead of the interpreter.  Worth it for chips of 2^17 rows and more.
int32_t lurkhip_trace_compile(lurkhip_ctx* ctx, lurkhip_toplevel* top, int32_t func_idx) {
    LH_CHECK_CTX(ctx);
    if (!top || func_idx < 0 || (size_t)func_idx >= top->t.funcs.size()) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "bad toplevel/func index");
    return guarded(ctx, [&]() -> int32_t {
        if (hipSetDevice(ctx->device) != hipSuccess) return lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "hipSetDevice failed");
        std::string log;
        if (!lurkhip::trace_jit_compile(ctx->device, program_of(top, (uint32_t)func_idx), &log))
            return lurkhip::set_error(ctx, LURKHIP_ERR_EXEC, "compiling the trace kernel of %s failed: %s", top->t.funcs[func_idx].name.c_str(), log.c_str());
        return LURKHIP_OK;
    });
}

// The same without loading (no device needed: build-time cache warming, CPU test of the generator): code object bytes, or a
// negative error with the compiler's message in `log`.
int32_t lurkhip_trace_compile_check(lurkhip_toplevel* top, int32_t func_idx, char* log, uint32_t log_cap) {
    if (!top || func_idx < 0 || (size_t)func_idx >= top->t.funcs.size()) return LURKHIP_ERR_INVALID_ARG;
    std::string l;
    size_t n = 0;
    try {
        n = lurkhip::trace_jit_compile_only(program_of(top, (uint32_t)func_idx), &l);
    } catch (const std::exception& e) {
        l = e.what();
    }
    if (log && log_cap) {
        const size_t k = std::min<size_t>(l.size(), log_cap - 1);
        memcpy(log, l.data(), k);
        log[k] = 0;
    }
    return n ? (int32_t)std::min<size_t>(n, 0x7fffffff) : LURKHIP_ERR_EXEC;
}

// The generated source of that kernel (inspection / tests): returns its length, copying at most cap - 1 characters.
int32_t lurkhip_trace_source(lurkhip_toplevel* top, int32_t func_idx, char* out, uint32_t cap) {
    if (!top || func_idx < 0 || (size_t)func_idx >= top->t.funcs.size()) return LURKHIP_ERR_INVALID_ARG;
    try {
        const std::string src = lurkhip::trace_jit_source(program_of(top, (uint32_t)func_idx));
        if (out && cap) {
            const size_t k = std::min<size_t>(src.size(), cap - 1);
            memcpy(out, src.data(), k);
            out[k] = 0;
        }
        return (int32_t)std::min<size_t>(src.size(), 0x7fffffff);
    } catch (const std::exception& e) {
        return fail(nullptr, LURKHIP_ERR_EXEC, e.what());
    }
}

// Device-resident inputs of one FuncChip trace (program + per-row arrays + row stream), so that the
// kernel can be re-run without touching the host (bench.py times exactly that).
}  // extern "C"

struct lurkhip_func_trace {
    std::vector<uint32_t> header;  // TH_WORDS words of the program
    void* dev = nullptr;           // one pooled block: program | args | outs | prov | depths | meta | stream
    size_t o_prog = 0, o_args = 0, o_outs = 0, o_prov = 0, o_dep = 0, o_meta = 0, o_str = 0, total = 0;
    uint32_t n = 0, height = 0, width = 0, start = 0;
    bool partial = false;
    size_t stream_words = 0;
    int kind = 0;        // 0: FuncChip, 1: MemChip, 2: BytesChip
    uint32_t mem_len = 0;
    bool is_real = false;
    void* host_keep = nullptr;  // page-locked source of an upload nobody waited for: lives as long as the handle
};

extern "C" {

// FuncChip::generate_trace (trace.rs:72-135), split in two: `prepare` flattens the shard's queries of one
// function into device-resident kernel inputs, `run` launches the row kernel into a height x width buffer.
int32_t lurkhip_func_trace_prepare(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r, int32_t func_idx,
                                   uint32_t shard_index, uint32_t max_shard_size, lurkhip_func_trace** out) {
    LH_CHECK_CTX(ctx);
    if (!top || !r || !out || r->top != top || func_idx < 0 || (size_t)func_idx >= top->t.funcs.size())
        return fail(ctx, LURKHIP_ERR_INVALID_ARG, "bad toplevel/record/func index");
    *out = nullptr;
    return guarded(ctx, [&]() -> int32_t {
        const lair::Func& f = top->t.funcs[func_idx];
        const lair::QueryMap& qm = r->q.func_queries[func_idx];
        auto [start, end] = lair::shard_range(qm.size(), shard_index, max_shard_size);
        const uint32_t n = (uint32_t)(end - start), height = next_pow2(n);
        const std::vector<uint32_t>& prog = program_of(top, (uint32_t)func_idx);
        // ---- flatten the per-row inputs (args / outputs / provide / depth / stream)
        std::vector<uint32_t> args((size_t)n * f.input_size), outs((size_t)n * f.output_size), prov((size_t)n * 2), depths(n);
        std::vector<lair::RowMeta> meta(n);
        std::vector<uint32_t> stream;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t* key = qm.key(start + i);
            const lair::QueryResult& res = qm.vals[start + i];
            if (!res.has_output) throw lair::ExecError("Result not computed");
            memcpy(&args[(size_t)i * f.input_size], key, f.input_size * 4);
            memcpy(&outs[(size_t)i * f.output_size], qm.output(res), f.output_size * 4);
            prov[2 * (size_t)i] = res.provide.nonce;
            prov[2 * (size_t)i + 1] = res.provide.count;
            depths[i] = res.depth;
            if (stream.size() + res.n_hints + 2 * ((size_t)res.n_requires + res.n_depth_requires) > 0xffffff00ull)
                throw lair::ExecError("row stream exceeds 2^32 words; use a smaller shard");
            meta[i].offset = (uint32_t)stream.size();
            meta[i].n_hints = res.n_hints;
            meta[i].n_requires = res.n_requires;
            meta[i].n_depth_requires = res.n_depth_requires;
            stream.insert(stream.end(), qm.hints(res), qm.hints(res) + res.n_hints);
            const lair::Record* recs = qm.requires_of(res);  // requires, then depth requires
            for (uint32_t k = 0; k < res.n_requires + res.n_depth_requires; k++) {
                stream.push_back(recs[k].nonce);
                stream.push_back(recs[k].count);
            }
        }
        // ---- one staging buffer: program | args | outs | prov | depths | meta | stream
        auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
        auto* p = new lurkhip_func_trace();
        p->o_prog = 0;
        p->o_args = al(p->o_prog + prog.size() * 4);
        p->o_outs = al(p->o_args + args.size() * 4);
        p->o_prov = al(p->o_outs + outs.size() * 4);
        p->o_dep = al(p->o_prov + prov.size() * 4);
        p->o_meta = al(p->o_dep + depths.size() * 4);
        p->o_str = al(p->o_meta + meta.size() * sizeof(lair::RowMeta));
        p->total = al(p->o_str + stream.size() * 4 + 4);
        p->n = n;
        p->height = height;
        p->width = prog[lair::TH_WIDTH];
        p->start = (uint32_t)start;
        p->partial = f.partial;
        p->stream_words = stream.size();
        p->header.assign(prog.begin(), prog.begin() + lair::TH_WORDS);
        std::vector<uint8_t> host(p->total, 0);
        memcpy(&host[p->o_prog], prog.data(), prog.size() * 4);
        if (n) {
            memcpy(&host[p->o_args], args.data(), args.size() * 4);
            memcpy(&host[p->o_outs], outs.data(), outs.size() * 4);
            memcpy(&host[p->o_prov], prov.data(), prov.size() * 4);
            memcpy(&host[p->o_dep], depths.data(), depths.size() * 4);
            memcpy(&host[p->o_meta], meta.data(), meta.size() * sizeof(lair::RowMeta));
            if (!stream.empty()) memcpy(&host[p->o_str], stream.data(), stream.size() * 4);
        }
        int32_t s = lurkhip::pool_alloc(ctx, p->total, &p->dev);
        if (s != LURKHIP_OK) {
            delete p;
            return s;
        }
        hipError_t e = hipMemcpyAsync(p->dev, host.data(), p->total, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // `host` dies at scope exit
        if (e != hipSuccess) {
            lurkhip::pool_release(ctx, p->dev);
            delete p;
            return lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "row-stream upload failed: %s", hipGetErrorString(e));
        }
        *out = p;
        return LURKHIP_OK;
    });
}

// The same flattening for several functions of one shard at once, on host threads: every function's rows are cut into
// ranges, a first sweep sizes each range's part of the row stream, a second one writes the rows straight into one page-locked
// staging buffer (no intermediate vectors), and a function's block is queued for upload on the context's stream as soon as its
// last range is done -- the copy of one function runs under the flattening of the next.  Nothing waits for the copies here:
// they are ordered before any later work on the stream, and the staging buffer is not rewritten before `prep_done` has passed.
// (The reference parallelises trace generation per row, trace.rs:86-132; the per-row work left on the host here is this copy.)
// out[i] = nullptr for a function without rows in the shard.  n_threads = 0: one per usable core (usable_cores), at most 32.
int32_t lurkhip_func_trace_prepare_many(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r, uint32_t n_funcs,
                                        const int32_t* func_idx, uint32_t shard_index, uint32_t max_shard_size, uint32_t n_threads,
                                        lurkhip_func_trace** out) {
    LH_CHECK_CTX(ctx);
    if (!top || !r || !out || !func_idx || r->top != top) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "bad toplevel/record");
    for (uint32_t k = 0; k < n_funcs; k++) {
        out[k] = nullptr;
        if (func_idx[k] < 0 || (size_t)func_idx[k] >= top->t.funcs.size()) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "bad func index");
    }
    return guarded(ctx, [&]() -> int32_t {
        constexpr uint32_t RANGE = 1u << 14;  // rows per task
        struct Job {
            const lair::Func* f = nullptr;
            const lair::QueryMap* qm = nullptr;
            const std::vector<uint32_t>* prog = nullptr;
            size_t start = 0;
            uint32_t n = 0, first_range = 0, n_ranges = 0;
            size_t stage_off = 0;
            lurkhip_func_trace* p = nullptr;
            std::atomic<uint32_t> left{0};
        };
        std::vector<Job> jobs(n_funcs);
        uint32_t total_ranges = 0;
        for (uint32_t k = 0; k < n_funcs; k++) {
            Job& j = jobs[k];
            j.f = &top->t.funcs[func_idx[k]];
            j.qm = &r->q.func_queries[func_idx[k]];
            auto [s0, e0] = lair::shard_range(j.qm->size(), shard_index, max_shard_size);
            j.start = s0;
            j.n = (uint32_t)(e0 - s0);
            j.prog = &program_of(top, (uint32_t)func_idx[k]);
            j.first_range = total_ranges;
            j.n_ranges = (j.n + RANGE - 1) / RANGE;
            total_ranges += j.n_ranges;
        }
        struct Range {
            uint32_t job, lo, hi;
            size_t words = 0;  // row-stream words of the range, then its first word's offset
        };
        std::vector<Range> ranges(total_ranges);
        for (uint32_t k = 0; k < n_funcs; k++)
            for (uint32_t c = 0; c < jobs[k].n_ranges; c++)
                ranges[jobs[k].first_range + c] = Range{k, c * RANGE, std::min(jobs[k].n, (c + 1) * RANGE), 0};
        uint32_t nt = n_threads ? n_threads : std::min(32u, usable_cores());
        nt = std::max(1u, std::min(nt, std::max(1u, total_ranges)));
        std::string worker_err;
        std::mutex err_mu;
        auto run_parallel = [&](auto&& body) {
            std::atomic<uint32_t> next{0};
            auto work = [&]() {
                try {
                    for (uint32_t i; (i = next.fetch_add(1)) < total_ranges;) body(i);
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> g(err_mu);
                    if (worker_err.empty()) worker_err = e.what();
                    next.store(total_ranges);
                }
            };
            std::vector<std::thread> ths;
            for (uint32_t t = 1; t < nt; t++) ths.emplace_back(work);
            work();
            for (auto& t : ths) t.join();
        };
        // ---- sweep 1: words of the row stream per range
        run_parallel([&](uint32_t i) {
            Range& g = ranges[i];
            const Job& j = jobs[g.job];
            size_t wsum = 0;
            for (uint32_t row = g.lo; row < g.hi; row++) {
                const lair::QueryResult& res = j.qm->vals[j.start + row];
                if (!res.has_output) throw lair::ExecError("Result not computed");
                wsum += res.n_hints + 2 * ((size_t)res.n_requires + res.n_depth_requires);
            }
            g.words = wsum;
        });
        if (!worker_err.empty()) throw lair::ExecError(worker_err);
        // ---- layouts: program | args | outs | prov | depths | meta | stream, one staging block per function
        auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
        size_t stage_total = 0;
        for (uint32_t k = 0; k < n_funcs; k++) {
            Job& j = jobs[k];
            if (j.n == 0) continue;
            size_t words = 0;
            for (uint32_t c = 0; c < j.n_ranges; c++) {
                Range& g = ranges[j.first_range + c];
                const size_t w = g.words;
                g.words = words;  // now: offset of the range's first word
                words += w;
            }
            if (words > 0xffffff00ull) throw lair::ExecError("row stream exceeds 2^32 words; use a smaller shard");
            auto* p = new lurkhip_func_trace();
            j.p = p;
            const lair::Func& f = *j.f;
            p->o_prog = 0;
            p->o_args = al(p->o_prog + j.prog->size() * 4);
            p->o_outs = al(p->o_args + (size_t)j.n * f.input_size * 4);
            p->o_prov = al(p->o_outs + (size_t)j.n * f.output_size * 4);
            p->o_dep = al(p->o_prov + (size_t)j.n * 8);
            p->o_meta = al(p->o_dep + (size_t)j.n * 4);
            p->o_str = al(p->o_meta + (size_t)j.n * sizeof(lair::RowMeta));
            p->total = al(p->o_str + words * 4 + 4);
            p->n = j.n;
            p->height = next_pow2(j.n);
            p->width = (*j.prog)[lair::TH_WIDTH];
            p->start = (uint32_t)j.start;
            p->partial = f.partial;
            p->stream_words = words;
            p->header.assign(j.prog->begin(), j.prog->begin() + lair::TH_WORDS);
            j.stage_off = stage_total;
            stage_total += p->total;
            j.left.store(j.n_ranges);
        }
        auto cleanup = [&]() {
            for (Job& j : jobs)
                if (j.p) {
                    if (j.p->dev) lurkhip::pool_release(ctx, j.p->dev);
                    delete j.p;
                    j.p = nullptr;
                }
        };
        // ---- the staging buffer of this call (two take turns): wait until the uploads that last read it are done, grow if needed
        const int turn = ctx->prep_turn;
        ctx->prep_turn ^= 1;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess && ctx->prep_done[turn]) e = hipEventSynchronize(ctx->prep_done[turn]);
        if (e == hipSuccess && ctx->prep_stage_bytes[turn] < stage_total) {
            if (ctx->prep_stage[turn]) (void)hipHostFree(ctx->prep_stage[turn]);
            ctx->prep_stage[turn] = nullptr;
            ctx->prep_stage_bytes[turn] = 0;
            e = hipHostMalloc(&ctx->prep_stage[turn], stage_total + stage_total / 8 + 4096, hipHostMallocDefault);
            if (e == hipSuccess) ctx->prep_stage_bytes[turn] = stage_total + stage_total / 8 + 4096;
        }
        if (e == hipSuccess && !ctx->prep_done[turn]) e = hipEventCreateWithFlags(&ctx->prep_done[turn], hipEventDisableTiming);
        if (e != hipSuccess) {
            cleanup();
            return lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "row-stream staging failed: %s", hipGetErrorString(e));
        }
        uint8_t* stage = (uint8_t*)ctx->prep_stage[turn];
        // ---- sweep 2: the rows, written in place; the main thread uploads finished functions in order while the others run
        std::atomic<uint32_t> next{0};
        auto fill = [&](uint32_t i) {
            const Range& g = ranges[i];
            Job& j = jobs[g.job];
            const lair::Func& f = *j.f;
            const lair::QueryMap& qm = *j.qm;
            uint8_t* base = stage + j.stage_off;
            const lurkhip_func_trace* p = j.p;
            uint32_t* args = (uint32_t*)(base + p->o_args);
            uint32_t* outs = (uint32_t*)(base + p->o_outs);
            uint32_t* prov = (uint32_t*)(base + p->o_prov);
            uint32_t* deps = (uint32_t*)(base + p->o_dep);
            lair::RowMeta* meta = (lair::RowMeta*)(base + p->o_meta);
            uint32_t* str = (uint32_t*)(base + p->o_str);
            if (g.lo == 0) {
                memcpy(base + p->o_prog, j.prog->data(), j.prog->size() * 4);
                str[p->stream_words] = 0;  // the padding word behind the stream
            }
            size_t at = g.words;
            for (uint32_t row = g.lo; row < g.hi; row++) {
                const uint32_t* key = qm.key(j.start + row);
                const lair::QueryResult& res = qm.vals[j.start + row];
                memcpy(&args[(size_t)row * f.input_size], key, f.input_size * 4);
                memcpy(&outs[(size_t)row * f.output_size], qm.output(res), f.output_size * 4);
                prov[2 * (size_t)row] = res.provide.nonce;
                prov[2 * (size_t)row + 1] = res.provide.count;
                deps[row] = res.depth;
                meta[row].offset = (uint32_t)at;
                meta[row].n_hints = res.n_hints;
                meta[row].n_requires = res.n_requires;
                meta[row].n_depth_requires = res.n_depth_requires;
                memcpy(&str[at], qm.hints(res), (size_t)res.n_hints * 4);
                at += res.n_hints;
                const lair::Record* recs = qm.requires_of(res);  // requires, then depth requires
                const uint32_t nr = res.n_requires + res.n_depth_requires;
                for (uint32_t q = 0; q < nr; q++) {
                    str[at++] = recs[q].nonce;
                    str[at++] = recs[q].count;
                }
            }
            j.left.fetch_sub(1, std::memory_order_release);
        };
        auto work = [&]() {
            for (uint32_t i; (i = next.fetch_add(1)) < total_ranges;) fill(i);
        };
        std::vector<std::thread> ths;
        for (uint32_t t = 1; t < nt; t++) ths.emplace_back(work);
        int32_t status = LURKHIP_OK;
        for (uint32_t k = 0; k < n_funcs && status == LURKHIP_OK; k++) {
            Job& j = jobs[k];
            if (!j.p) continue;
            // help with the ranges until this function's block is complete
            while (j.left.load(std::memory_order_acquire) != 0) {
                const uint32_t i = next.fetch_add(1);
                if (i < total_ranges) fill(i);
                else std::this_thread::yield();
            }
            status = lurkhip::pool_alloc(ctx, j.p->total, &j.p->dev);
            if (status != LURKHIP_OK) break;
            e = hipMemcpyAsync(j.p->dev, stage + j.stage_off, j.p->total, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) status = lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "row-stream upload failed: %s", hipGetErrorString(e));
        }
        next.store(total_ranges);
        for (auto& t : ths) t.join();
        if (status == LURKHIP_OK) {
            e = hipEventRecord(ctx->prep_done[turn], ctx->stream);
            if (e != hipSuccess) status = lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "row-stream upload failed: %s", hipGetErrorString(e));
        }
        if (status != LURKHIP_OK) {
            (void)hipStreamSynchronize(ctx->stream);
            cleanup();
            return status;
        }
        for (uint32_t k = 0; k < n_funcs; k++) out[k] = jobs[k].p;
        return LURKHIP_OK;
    });
}

// shape[0..5) = n_real, height, width, bytes of device-resident inputs, stream words
int32_t lurkhip_func_trace_shape_of(const lurkhip_func_trace* p, uint64_t* shape) {
    if (!p || !shape) return LURKHIP_ERR_INVALID_ARG;
    shape[0] = p->n;
    shape[1] = p->height;
    shape[2] = p->width;
    shape[3] = p->total;
    shape[4] = p->stream_words;
    return LURKHIP_OK;
}

// MemChip / BytesChip counterparts of lurkhip_func_trace_prepare: the chip's inputs (values + provide records, or the
// 65536 x 6 byte-lookup records) resident on the device, run by lurkhip_func_trace_run.
int32_t lurkhip_mem_trace_prepare(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t mem_len, lurkhip_func_trace** out) {
    LH_CHECK_CTX(ctx);
    if (!r || !out) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    return guarded(ctx, [&]() -> int32_t {
        const auto& mm = r->q.mem_queries[lair::mem_index_from_len(mem_len)];
        const uint32_t n = (uint32_t)mm.size();
        const size_t host_words = (size_t)n * (mem_len + 2) + 4;
        uint32_t* host = nullptr;
        if (hipHostMalloc((void**)&host, host_words * 4, hipHostMallocDefault) != hipSuccess)
            return lurkhip::set_error(ctx, LURKHIP_ERR_OOM, "page-locked staging of a memory table failed");
        memset(host, 0, host_words * 4);
        for (uint32_t i = 0; i < n; i++) {
            memcpy(&host[(size_t)i * mem_len], mm.key(i), mem_len * 4);
            host[(size_t)n * mem_len + 2 * i] = mm.vals[i].provide.nonce;
            host[(size_t)n * mem_len + 2 * i + 1] = mm.vals[i].provide.count;
        }
        auto* p = new lurkhip_func_trace();
        p->kind = 1;
        p->mem_len = mem_len;
        p->n = n;
        p->height = std::max(4u, next_pow2(n));
        p->width = 4 + mem_len;
        p->total = host_words * 4;
        p->host_keep = host;
        int32_t s = lurkhip::pool_alloc(ctx, p->total, &p->dev);
        hipError_t e = hipSuccess;
        if (s == LURKHIP_OK) e = hipMemcpyAsync(p->dev, host, p->total, hipMemcpyHostToDevice, ctx->stream);  // ordered before later work on the stream
        if (s != LURKHIP_OK || e != hipSuccess) {
            (void)hipStreamSynchronize(ctx->stream);
            if (p->dev) lurkhip::pool_release(ctx, p->dev);
            (void)hipHostFree(host);
            delete p;
            return s != LURKHIP_OK ? s : lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "mem table upload failed: %s", hipGetErrorString(e));
        }
        *out = p;
        return LURKHIP_OK;
    });
}

int32_t lurkhip_bytes_trace_prepare(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t shard_index, lurkhip_func_trace** out) {
    LH_CHECK_CTX(ctx);
    if (!r || !out) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    return guarded(ctx, [&]() -> int32_t {
        const bool is_real = shard_index == 0 && !r->q.bytes.empty();
        const size_t host_words = (size_t)65536 * 12;
        uint32_t* host = nullptr;
        if (hipHostMalloc((void**)&host, host_words * 4, hipHostMallocDefault) != hipSuccess)
            return lurkhip::set_error(ctx, LURKHIP_ERR_OOM, "page-locked staging of the byte records failed");
        memset(host, 0, host_words * 4);
        if (is_real)
            for (uint32_t key = 0; key < 65536; key++)
                for (int k = 0; k < lair::BYTES_KINDS; k++) {
                    const lair::Record rec = r->q.bytes.get((uint16_t)key, k);
                    host[(size_t)key * 12 + 2 * k] = rec.nonce;
                    host[(size_t)key * 12 + 2 * k + 1] = rec.count;
                }
        auto* p = new lurkhip_func_trace();
        p->kind = 2;
        p->is_real = is_real;
        p->n = is_real ? 65536 : 0;
        p->height = 65536;
        p->width = 13;
        p->total = host_words * 4;
        p->host_keep = host;
        int32_t s = lurkhip::pool_alloc(ctx, p->total, &p->dev);
        hipError_t e = hipSuccess;
        if (s == LURKHIP_OK) e = hipMemcpyAsync(p->dev, host, p->total, hipMemcpyHostToDevice, ctx->stream);  // ordered before later work on the stream
        if (s != LURKHIP_OK || e != hipSuccess) {
            (void)hipStreamSynchronize(ctx->stream);
            if (p->dev) lurkhip::pool_release(ctx, p->dev);
            (void)hipHostFree(host);
            delete p;
            return s != LURKHIP_OK ? s : lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "byte record upload failed: %s", hipGetErrorString(e));
        }
        *out = p;
        return LURKHIP_OK;
    });
}

int32_t lurkhip_func_trace_run(lurkhip_ctx* ctx, const lurkhip_func_trace* p, uint32_t* out_dev, int32_t repr) {
    return lurkhip_func_trace_run_pitched(ctx, p, out_dev, 0, repr);
}

// Rows [first_row, first_row + n_rows) of the trace only, into out_dev[n_rows][out_pitch]: a rank's block of rows when several ranks
// prove one shard together (include/lurkhip.h: lurkhip_shard_commit_split with main_row_blocks).  Every generator works row by row
// from per-row inputs, so a block is the same kernels on offset pointers: padding rows keep their nonce (trace.rs:82-84: range.start
// + i for ALL rows), a memory table its pointer column i + 1 (memory.rs:43-52).
int32_t lurkhip_func_trace_run_rows(lurkhip_ctx* ctx, const lurkhip_func_trace* p, uint32_t first_row, uint32_t n_rows, uint32_t* out_dev, uint32_t out_pitch,
                                    int32_t repr) {
    LH_CHECK_CTX(ctx);
    if (!p || !out_dev) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    if ((uint64_t)first_row + n_rows > p->height) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "row block outside the trace");
    const uint32_t r0 = std::min(first_row, p->n), real = std::min(first_row + n_rows, p->n) - r0;  // the block's real rows are r0 .. r0 + real
    if (p->kind == 1)
        return lurkhip::trace_mem_dev_pitched(ctx, p->mem_len, real, n_rows, (const uint32_t*)p->dev + (size_t)r0 * p->mem_len,
                                              (const uint32_t*)p->dev + (size_t)p->n * p->mem_len + 2 * (size_t)r0, out_dev, repr, out_pitch, first_row);
    if (p->kind == 2)
        return lurkhip::trace_bytes_dev_pitched(ctx, p->is_real ? (const uint32_t*)p->dev + (size_t)first_row * 12 : nullptr, p->is_real ? 1 : 0, out_dev, repr, out_pitch, n_rows);
    const uint8_t* d = (const uint8_t*)p->dev;
    const uint32_t n_in = p->header[lair::TH_INPUT], n_out = p->header[lair::TH_OUTPUT];
    return lurkhip::trace_func_dev_pitched(ctx, (const uint32_t*)(d + p->o_prog), p->header.data(), real, n_rows, p->start + first_row,
                                           (const uint32_t*)(d + p->o_args) + (size_t)r0 * n_in, (const uint32_t*)(d + p->o_outs) + (size_t)r0 * n_out,
                                           (const uint32_t*)(d + p->o_prov) + 2 * (size_t)r0, p->partial ? (const uint32_t*)(d + p->o_dep) + r0 : nullptr,
                                           d + p->o_meta + (size_t)r0 * sizeof(lair::RowMeta), (const uint32_t*)(d + p->o_str), out_dev, repr, out_pitch);
}

int32_t lurkhip_func_trace_run_pitched(lurkhip_ctx* ctx, const lurkhip_func_trace* p, uint32_t* out_dev, uint32_t out_pitch, int32_t repr) {
    LH_CHECK_CTX(ctx);
    if (!p || !out_dev) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    if (p->kind == 1)
        return lurkhip::trace_mem_dev_pitched(ctx, p->mem_len, p->n, p->height, (const uint32_t*)p->dev,
                                              (const uint32_t*)p->dev + (size_t)p->n * p->mem_len, out_dev, repr, out_pitch, 0);
    if (p->kind == 2) return lurkhip::trace_bytes_dev_pitched(ctx, (const uint32_t*)p->dev, p->is_real ? 1 : 0, out_dev, repr, out_pitch, 65536);
    const uint8_t* d = (const uint8_t*)p->dev;
    return lurkhip::trace_func_dev_pitched(ctx, (const uint32_t*)(d + p->o_prog), p->header.data(), p->n, p->height, p->start,
                                           (const uint32_t*)(d + p->o_args), (const uint32_t*)(d + p->o_outs), (const uint32_t*)(d + p->o_prov),
                                           p->partial ? (const uint32_t*)(d + p->o_dep) : nullptr, d + p->o_meta, (const uint32_t*)(d + p->o_str),
                                           out_dev, repr, out_pitch);
}

int32_t lurkhip_trace_group_layout(uint32_t n, const uint32_t* log_heights, const uint32_t* widths, uint32_t* pitches, uint32_t* col_starts,
                                   int32_t* groups, int32_t* n_groups) {
    if (!log_heights || !widths || !pitches || !col_starts || !groups || !n_groups) return LURKHIP_ERR_INVALID_ARG;
    lurkhip::plan_source_groups((int)n, log_heights, widths, pitches, col_starts, groups, n_groups);
    return LURKHIP_OK;
}

int32_t lurkhip_func_trace_run_many(lurkhip_ctx* ctx, uint32_t n, const lurkhip_func_trace* const* ps, uint32_t* const* outs_dev, int32_t repr) {
    return lurkhip_func_trace_run_many_pitched(ctx, n, ps, outs_dev, nullptr, repr);
}

int32_t lurkhip_func_trace_run_many_pitched(lurkhip_ctx* ctx, uint32_t n, const lurkhip_func_trace* const* ps, uint32_t* const* outs_dev,
                                            const uint32_t* out_pitches, int32_t repr) {
    LH_CHECK_CTX(ctx);
    if (n && (!ps || !outs_dev)) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    // The chips' trace kernels are independent of one another.  The tall ones (2^13 rows and more) fill the device and stay on the
    // context's stream; the short ones -- the hash chips' one Poseidon2 witness per lane, ingress / egress, the memory tables: a few
    // waves each, 70 us of latency per launch -- are dealt to the context's side lanes, tallest first, and run under one another
    // and under the tall kernels (with no tall chip the context's own stream takes a share).  Joined before the call returns
    // control of the stream: whatever is queued next (the main commitment) is ordered after every trace.
    constexpr uint32_t TALL = 1u << 13;
    std::vector<uint32_t> order;
    uint32_t n_tall = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!ps[i] || !outs_dev[i]) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
        order.push_back(i);
        n_tall += ps[i]->height >= TALL ? 1u : 0u;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ps[a]->height > ps[b]->height; });
    lurkhip::SideLane lane(ctx);
    if (n - n_tall >= 2) LH_TRY(lane.open());
    const uint32_t slots = (uint32_t)lane.lanes + (n_tall ? 0u : 1u);  // slot `lanes` (when there is one) = the context's own stream
    uint32_t k = 0;
    for (uint32_t i : order) {
        const bool tall = ps[i]->height >= TALL;
        const uint32_t slot = tall ? 0u : k++ % slots;
        const auto on_side = lane.on_side(!tall && slot < (uint32_t)lane.lanes, slot);
        LH_TRY(lurkhip_func_trace_run_pitched(ctx, ps[i], outs_dev[i], out_pitches ? out_pitches[i] : 0u, repr));
    }
    return lane.close();
}

// ---- a prepared trace as bytes (round 5): the executing process hands a shard's kernel inputs to the process that proves it.
// The handle IS one device block plus a few numbers, so the blob is: 28 words of header (the last four: the block's length and its checksum) | the TH_WORDS program header (FuncChip) |
// the block.  Little-endian words; the block's own layout is lurkhip_func_trace_prepare's.
namespace {
constexpr uint32_t BLOB_MAGIC = 0x3254464cu;  // "LFT2" (round 6: the block's length and a checksum travel in the header)
constexpr size_t BLOB_HEAD_WORDS = 28;
// 64-bit multiply-xorshift fold of the block's words: a truncated or corrupted blob must not import cleanly
uint64_t blob_checksum(const uint8_t* body, uint64_t bytes) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ bytes;
    const uint64_t n = bytes / 8;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t v;
        memcpy(&v, body + 8 * i, 8);
        h = (h ^ v) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32;
    }
    for (uint64_t i = 8 * n; i < bytes; i++) h = (h ^ body[i]) * 0x100000001B3ull;
    return h;
}
}  // namespace

int32_t lurkhip_func_trace_export_size(const lurkhip_func_trace* p, uint64_t* bytes) {
    if (!p || !bytes) return LURKHIP_ERR_INVALID_ARG;
    *bytes = (BLOB_HEAD_WORDS + p->header.size()) * 4 + p->total;
    return LURKHIP_OK;
}

int32_t lurkhip_func_trace_export(lurkhip_ctx* ctx, const lurkhip_func_trace* p, void* out_host, uint64_t bytes) {
    LH_CHECK_CTX(ctx);
    if (!p || !out_host) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    uint64_t need = 0;
    (void)lurkhip_func_trace_export_size(p, &need);
    if (bytes < need) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "export buffer too small");
    uint32_t* w = (uint32_t*)out_host;
    const uint64_t offs[7] = {p->o_prog, p->o_args, p->o_outs, p->o_prov, p->o_dep, p->o_meta, p->o_str};
    w[0] = BLOB_MAGIC;
    w[1] = (uint32_t)p->kind;
    w[2] = p->mem_len;
    w[3] = (p->is_real ? 1u : 0u) | (p->partial ? 2u : 0u);
    w[4] = p->n;
    w[5] = p->height;
    w[6] = p->width;
    w[7] = p->start;
    w[8] = (uint32_t)p->header.size();
    w[9] = (uint32_t)p->stream_words;                  // (< 2^32: a shard has at most 2^22 rows of a few hundred words)
    memcpy(&w[10], offs, sizeof offs);                 // words 10 .. 23
    memcpy(&w[BLOB_HEAD_WORDS], p->header.data(), p->header.size() * 4);
    uint8_t* body = (uint8_t*)(w + BLOB_HEAD_WORDS + p->header.size());
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the upload that filled the block (and anything still reading it)
    LH_HIP(ctx, hipMemcpy(body, p->dev, p->total, hipMemcpyDeviceToHost));
    const uint64_t tail[2] = {(uint64_t)p->total, blob_checksum(body, p->total)};
    memcpy(&w[24], tail, sizeof tail);  // words 24 .. 27
    return LURKHIP_OK;
}

int32_t lurkhip_func_trace_import(lurkhip_ctx* ctx, const void* blob_host, uint64_t bytes, lurkhip_func_trace** out) {
    LH_CHECK_CTX(ctx);
    if (!blob_host || !out) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    return guarded(ctx, [&]() -> int32_t {
        const uint32_t* w = (const uint32_t*)blob_host;
        if (bytes < BLOB_HEAD_WORDS * 4 || w[0] != BLOB_MAGIC) return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "not a prepared-trace blob");
        const uint32_t kind = w[1], hw = w[8];
        uint64_t offs[7];
        memcpy(offs, &w[10], sizeof offs);
        if (kind > 2 || hw > 4096 || (kind == 0 ? hw != lair::TH_WORDS : hw != 0) || bytes < (BLOB_HEAD_WORDS + hw) * 4)
            return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "prepared-trace blob: bad header");
        const uint64_t total = bytes - (BLOB_HEAD_WORDS + hw) * 4;
        uint64_t tail[2];
        memcpy(tail, &w[24], sizeof tail);
        if (tail[0] != total) return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "prepared-trace blob: %llu bytes of block where the header says %llu (truncated?)",
                                                        (unsigned long long)total, (unsigned long long)tail[0]);
        if (blob_checksum((const uint8_t*)blob_host + (BLOB_HEAD_WORDS + hw) * 4, total) != tail[1])
            return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "prepared-trace blob: checksum mismatch");
        auto* p = new lurkhip_func_trace();
        p->kind = (int)kind;
        p->mem_len = w[2];
        p->is_real = (w[3] & 1u) != 0;
        p->partial = (w[3] & 2u) != 0;
        p->n = w[4];
        p->height = w[5];
        p->width = w[6];
        p->start = w[7];
        p->stream_words = w[9];
        p->header.assign(w + BLOB_HEAD_WORDS, w + BLOB_HEAD_WORDS + hw);
        p->o_prog = offs[0], p->o_args = offs[1], p->o_outs = offs[2], p->o_prov = offs[3], p->o_dep = offs[4], p->o_meta = offs[5], p->o_str = offs[6];
        p->total = total;
        bool ok = p->n <= p->height && p->height > 0 && (p->height & (p->height - 1)) == 0 && p->width > 0 && total > 0;
        for (uint64_t o : offs) ok = ok && o <= total && o % 4 == 0;
        if (kind == 0) {
            ok = ok && p->header[lair::TH_MAGIC] == lair::TRACE_PROGRAM_MAGIC && p->header[lair::TH_WIDTH] == p->width;
            // every section inside the block and ahead of the next one: the trace kernels read n rows of each
            const uint64_t n = p->n, n_in = p->header[lair::TH_INPUT], n_out = p->header[lair::TH_OUTPUT];
            ok = ok && offs[0] <= offs[1] && offs[1] + n * n_in * 4 <= offs[2] && offs[2] + n * n_out * 4 <= offs[3] && offs[3] + n * 8 <= offs[4] &&
                 offs[4] + n * 4 <= offs[5] && offs[5] + n * sizeof(lair::RowMeta) <= offs[6] && offs[6] + (uint64_t)p->stream_words * 4 <= total;
        }
        if (kind == 1) ok = ok && (p->mem_len == 2 || p->mem_len == 3 || p->mem_len == 4 || p->mem_len == 5 || p->mem_len == 6 || p->mem_len == 8) &&
                            p->width == 4 + p->mem_len && total >= ((uint64_t)p->n * (p->mem_len + 2)) * 4;
        if (kind == 2) ok = ok && p->width == 13 && p->height == 65536 && total >= (uint64_t)65536 * 12 * 4;
        if (!ok) {
            delete p;
            return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "prepared-trace blob: inconsistent shape");
        }
        int32_t s = lurkhip::pool_alloc(ctx, total, &p->dev);
        if (s == LURKHIP_OK && hipMemcpy(p->dev, (const uint8_t*)blob_host + (BLOB_HEAD_WORDS + hw) * 4, total, hipMemcpyHostToDevice) != hipSuccess)
            s = lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "upload of a prepared trace failed");
        if (s != LURKHIP_OK) {
            if (p->dev) lurkhip::pool_release(ctx, p->dev);
            delete p;
            return s;
        }
        *out = p;
        return LURKHIP_OK;
    });
}

int32_t lurkhip_func_trace_free(lurkhip_ctx* ctx, lurkhip_func_trace* p) {
    LH_CHECK_CTX_NOLOCK(ctx);
    if (!p) return LURKHIP_OK;
    lurkhip::pool_release(ctx, p->dev);
    if (p->host_keep) {
        // the upload that reads it was queued on this context's own stream; ctx->stream may be a side stream right now (a locked
        // call on another thread reroutes it for a scope), main_stream never is
        (void)hipStreamSynchronize(ctx->main_stream);
        (void)hipHostFree(p->host_keep);
    }
    delete p;
    return LURKHIP_OK;
}

int32_t lurkhip_generate_trace_func_dev(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r, int32_t func_idx,
                                        uint32_t shard_index, uint32_t max_shard_size, uint32_t* out_dev, int32_t repr) {
    lurkhip_func_trace* p = nullptr;
    LH_TRY(lurkhip_func_trace_prepare(ctx, top, r, func_idx, shard_index, max_shard_size, &p));
    int32_t s = lurkhip_func_trace_run(ctx, p, out_dev, repr);
    lurkhip_func_trace_free(ctx, p);
    return s;
}

int32_t lurkhip_generate_trace_func(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r, int32_t func_idx,
                                    uint32_t shard_index, uint32_t max_shard_size, uint32_t* out_host, int32_t repr) {
    LH_CHECK_CTX(ctx);
    uint32_t n = 0, h = 0, w = 0;
    LH_TRY(lurkhip_func_trace_shape(r, func_idx, shard_index, max_shard_size, &n, &h, &w));
    void* dout = nullptr;
    LH_TRY(lurkhip::arena_get(ctx, 1, (size_t)h * w * 4, &dout));
    LH_TRY(lurkhip_generate_trace_func_dev(ctx, top, r, func_idx, shard_index, max_shard_size, (uint32_t*)dout, repr));
    LH_HIP(ctx, hipMemcpyAsync(out_host, dout, (size_t)h * w * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LURKHIP_OK;
}

// MemChip::generate_trace (memory.rs:30-69): height = max(4, next_pow2(len(mem))), width = 4 + mem_len
int32_t lurkhip_mem_trace_shape(const lurkhip_record* r, uint32_t mem_len, uint32_t* n_real, uint32_t* height, uint32_t* width) {
    if (!r) return LURKHIP_ERR_INVALID_ARG;
    return guarded(nullptr, [&]() -> int32_t {
        const auto& mm = r->q.mem_queries[lair::mem_index_from_len(mem_len)];
        uint32_t n = (uint32_t)mm.size();
        if (n_real) *n_real = n;
        if (height) *height = std::max(4u, next_pow2(n));
        if (width) *width = 4 + mem_len;
        return LURKHIP_OK;
    });
}

int32_t lurkhip_generate_trace_mem(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t mem_len, uint32_t* out_host, int32_t repr) {
    LH_CHECK_CTX(ctx);
    if (!r || !out_host) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    return guarded(ctx, [&]() -> int32_t {
        const auto& mm = r->q.mem_queries[lair::mem_index_from_len(mem_len)];
        const uint32_t n = (uint32_t)mm.size(), height = std::max(4u, next_pow2(n)), width = 4 + mem_len;
        std::vector<uint32_t> host((size_t)n * (mem_len + 2) + 1);
        for (uint32_t i = 0; i < n; i++) {
            memcpy(&host[(size_t)i * mem_len], mm.key(i), mem_len * 4);
            host[(size_t)n * mem_len + 2 * i] = mm.vals[i].provide.nonce;
            host[(size_t)n * mem_len + 2 * i + 1] = mm.vals[i].provide.count;
        }
        void *din = nullptr, *dout = nullptr;
        LH_TRY(lurkhip::arena_get(ctx, 3, host.size() * 4, &din));
        LH_TRY(lurkhip::arena_get(ctx, 1, (size_t)height * width * 4, &dout));
        LH_HIP(ctx, hipMemcpyAsync(din, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LH_TRY(lurkhip_trace_mem_dev(ctx, mem_len, n, height, (const uint32_t*)din, (const uint32_t*)din + (size_t)n * mem_len,
                                     (uint32_t*)dout, repr));
        LH_HIP(ctx, hipMemcpyAsync(out_host, dout, (size_t)height * width * 4, hipMemcpyDeviceToHost, ctx->stream));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return LURKHIP_OK;
    });
}

// BytesChip::generate_trace for this record (65536 x 13); an empty record gives the all-zero trace
// (bytes/trace.rs:81-84), as does any shard other than 0 (lair_chip.rs:104-111).
int32_t lurkhip_generate_trace_bytes(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t shard_index, uint32_t* out_host, int32_t repr) {
    LH_CHECK_CTX(ctx);
    if (!r || !out_host) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    return guarded(ctx, [&]() -> int32_t {
        const bool is_real = shard_index == 0 && !r->q.bytes.empty();
        std::vector<uint32_t> host((size_t)65536 * 12, 0);
        if (is_real)
            for (uint32_t key = 0; key < 65536; key++)
                for (int k = 0; k < lair::BYTES_KINDS; k++) {
                    const lair::Record rec = r->q.bytes.get((uint16_t)key, k);
                    host[(size_t)key * 12 + 2 * k] = rec.nonce;
                    host[(size_t)key * 12 + 2 * k + 1] = rec.count;
                }
        void *din = nullptr, *dout = nullptr;
        LH_TRY(lurkhip::arena_get(ctx, 3, host.size() * 4, &din));
        LH_TRY(lurkhip::arena_get(ctx, 1, (size_t)65536 * 13 * 4, &dout));
        LH_HIP(ctx, hipMemcpyAsync(din, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LH_TRY(lurkhip_trace_bytes_dev(ctx, (const uint32_t*)din, is_real, (uint32_t*)dout, repr));
        LH_HIP(ctx, hipMemcpyAsync(out_host, dout, (size_t)65536 * 13 * 4, hipMemcpyDeviceToHost, ctx->stream));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return LURKHIP_OK;
    });
}

const char* lurkhip_lair_last_error(void) { return g_err.c_str(); }

}  // extern "C"
