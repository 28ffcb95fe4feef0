// C ABI over the Lair host: toplevel construction, execution, layout queries and the
// generate_trace entry points that feed the device kernels.
//
// These are the seams a Rust shim would replace:
//   Toplevel::{new, execute_by_name}        /root/reference/src/lair/toplevel.rs:28-50, execute.rs:375-417
//   FuncChip::{from_name, width, generate_trace}   /root/reference/src/lair/func_chip.rs:34-80, trace.rs:72-135
//   MemChip / BytesChip / Entrypoint generate_trace  /root/reference/src/lair/lair_chip.rs:96-120
#include <cstring>

#include "../ctx.h"
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
        if (out) memcpy(out, o.data(), o.size() * 4);
        return LURKHIP_OK;
    });
}

int32_t lurkhip_record_inject_inv_query(lurkhip_record* r, int32_t func_idx, const uint32_t* inp, uint32_t n_inp,
                                        const uint32_t* out, uint32_t n_out) {
    if (!r) return LURKHIP_ERR_INVALID_ARG;
    return guarded(nullptr, [&]() -> int32_t {
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
        if (kind == 3) return (int64_t)r->q.bytes.records.size();
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
        std::vector<uint32_t> host((size_t)n * (mem_len + 2) + 4, 0);
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
        p->total = host.size() * 4;
        int32_t s = lurkhip::pool_alloc(ctx, p->total, &p->dev);
        hipError_t e = hipSuccess;
        if (s == LURKHIP_OK) e = hipMemcpyAsync(p->dev, host.data(), p->total, hipMemcpyHostToDevice, ctx->stream);
        if (s == LURKHIP_OK && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (s != LURKHIP_OK || e != hipSuccess) {
            if (p->dev) lurkhip::pool_release(ctx, p->dev);
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
        const bool is_real = shard_index == 0 && !r->q.bytes.records.empty();
        std::vector<uint32_t> host((size_t)65536 * 12, 0);
        if (is_real)
            for (const auto& kv : r->q.bytes.records) {
                const lair::Record recs[6] = {kv.second.range_u8, kv.second.range_u16, kv.second.less_than,
                                              kv.second.and_,     kv.second.xor_,      kv.second.or_};
                for (int k = 0; k < 6; k++) {
                    host[(size_t)kv.first * 12 + 2 * k] = recs[k].nonce;
                    host[(size_t)kv.first * 12 + 2 * k + 1] = recs[k].count;
                }
            }
        auto* p = new lurkhip_func_trace();
        p->kind = 2;
        p->is_real = is_real;
        p->n = is_real ? 65536 : 0;
        p->height = 65536;
        p->width = 13;
        p->total = host.size() * 4;
        int32_t s = lurkhip::pool_alloc(ctx, p->total, &p->dev);
        hipError_t e = hipSuccess;
        if (s == LURKHIP_OK) e = hipMemcpyAsync(p->dev, host.data(), p->total, hipMemcpyHostToDevice, ctx->stream);
        if (s == LURKHIP_OK && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (s != LURKHIP_OK || e != hipSuccess) {
            if (p->dev) lurkhip::pool_release(ctx, p->dev);
            delete p;
            return s != LURKHIP_OK ? s : lurkhip::set_error(ctx, LURKHIP_ERR_HIP, "byte record upload failed: %s", hipGetErrorString(e));
        }
        *out = p;
        return LURKHIP_OK;
    });
}

int32_t lurkhip_func_trace_run(lurkhip_ctx* ctx, const lurkhip_func_trace* p, uint32_t* out_dev, int32_t repr) {
    LH_CHECK_CTX(ctx);
    if (!p || !out_dev) return fail(ctx, LURKHIP_ERR_INVALID_ARG, "null argument");
    if (p->kind == 1)
        return lurkhip_trace_mem_dev(ctx, p->mem_len, p->n, p->height, (const uint32_t*)p->dev,
                                     (const uint32_t*)p->dev + (size_t)p->n * p->mem_len, out_dev, repr);
    if (p->kind == 2) return lurkhip_trace_bytes_dev(ctx, (const uint32_t*)p->dev, p->is_real ? 1 : 0, out_dev, repr);
    const uint8_t* d = (const uint8_t*)p->dev;
    return lurkhip_trace_func_dev(ctx, (const uint32_t*)(d + p->o_prog), p->header.data(), p->n, p->height, p->start,
                                  (const uint32_t*)(d + p->o_args), (const uint32_t*)(d + p->o_outs), (const uint32_t*)(d + p->o_prov),
                                  p->partial ? (const uint32_t*)(d + p->o_dep) : nullptr, d + p->o_meta, (const uint32_t*)(d + p->o_str),
                                  out_dev, repr);
}

int32_t lurkhip_func_trace_free(lurkhip_ctx* ctx, lurkhip_func_trace* p) {
    LH_CHECK_CTX(ctx);
    if (!p) return LURKHIP_OK;
    lurkhip::pool_release(ctx, p->dev);
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
        const bool is_real = shard_index == 0 && !r->q.bytes.records.empty();
        std::vector<uint32_t> host((size_t)65536 * 12, 0);
        if (is_real)
            for (const auto& kv : r->q.bytes.records) {
                const lair::Record recs[6] = {kv.second.range_u8, kv.second.range_u16, kv.second.less_than,
                                              kv.second.and_,     kv.second.xor_,      kv.second.or_};
                for (int k = 0; k < 6; k++) {
                    host[(size_t)kv.first * 12 + 2 * k] = recs[k].nonce;
                    host[(size_t)kv.first * 12 + 2 * k + 1] = recs[k].count;
                }
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
