// Host emitter: bytecode -> degree-resolved micro-program for the trace kernel.
//
// The walk mirrors /root/reference/src/lair/trace.rs:256-418 op by op; everything that is the same for
// every row of a function (degrees, aux ownership, sizes of callee outputs, chip shapes) is decided
// here once instead of per row on the device.
#include <algorithm>

#include "../babybear.h"
#include "lair.h"
#include "trace_program.h"

namespace lair {

namespace {

struct Emitter {
    const Toplevel& t;
    std::vector<uint32_t> code;
    uint32_t max_vars = 0;

    explicit Emitter(const Toplevel& tl) : t(tl) {}

    // returns the word offset of the emitted block
    uint32_t emit_block(const Block& b, std::vector<uint8_t> degrees /* by value: restored per branch */) {
        const uint32_t start = (uint32_t)code.size();
        for (const Op& op : b.ops) emit_op(op, degrees);
        max_vars = std::max<uint32_t>(max_vars, (uint32_t)degrees.size());
        const Ctrl& c = b.ctrl;
        if (c.kind == Ctrl::Return) {
            code.push_back(T_RETURN);
            code.push_back(c.ident);
            return start;
        }
        // reserve the jump table, emit the targets, then patch
        const bool many = c.kind == Ctrl::ChooseMany;
        const uint32_t nv = many ? (uint32_t)c.vars.size() : 1;
        const uint32_t n_cases = (uint32_t)c.branches.size();
        const uint32_t head = (uint32_t)code.size();
        if (many) {
            code.push_back(T_CHOOSE_MANY);
            code.push_back(nv);
            code.push_back(n_cases);
            code.push_back(0);
            for (uint32_t v : c.vars) code.push_back(v);
        } else {
            code.push_back(T_CHOOSE);
            code.push_back(c.var);
            code.push_back(n_cases);
            code.push_back(0);
        }
        const uint32_t table = (uint32_t)code.size();
        code.resize(code.size() + (size_t)n_cases * (nv + 1));
        // distinct blocks are emitted once even when several keys share them
        std::vector<std::pair<const Block*, uint32_t>> done;
        auto target = [&](const Block* blk) {
            for (auto& d : done)
                if (d.first == blk) return d.second;
            uint32_t off = emit_block(*blk, degrees);
            done.push_back({blk, off});
            return off;
        };
        for (uint32_t i = 0; i < n_cases; i++) {
            const auto& kv = c.branches[i];
            uint32_t off = target(kv.second.get());
            uint32_t* slot = &code[table + (size_t)i * (nv + 1)];
            for (uint32_t k = 0; k < nv; k++) slot[k] = bb::to_monty(kv.first[k]);
            slot[nv] = off;
        }
        if (c.def) {
            uint32_t off = emit_block(*c.def, degrees);
            code[head + 3] = off;
        }
        return start;
    }

    void emit_op(const Op& op, std::vector<uint8_t>& deg) {
        switch (op.kind) {
            case OpKind::AssertEq:
            case OpKind::Emit:
            case OpKind::Breakpoint:
            case OpKind::Debug:
                break;
            case OpKind::AssertNe:
                code.push_back(T_ASSERT_NE | ((uint32_t)op.a.size() << 8));
                code.insert(code.end(), op.a.begin(), op.a.end());
                code.insert(code.end(), op.b.begin(), op.b.end());
                break;
            case OpKind::Contains:
                code.push_back(T_CONTAINS | ((uint32_t)op.a.size() << 8));
                code.push_back(op.y);
                code.insert(code.end(), op.a.begin(), op.a.end());
                break;
            case OpKind::Const:
                code.push_back(T_CONST);
                code.push_back(bb::to_monty(op.c));
                deg.push_back(0);
                break;
            case OpKind::Add:
            case OpKind::Sub:
                code.push_back(op.kind == OpKind::Add ? T_ADD : T_SUB);
                code.push_back(op.x);
                code.push_back(op.y);
                deg.push_back(std::max(deg.at(op.x), deg.at(op.y)));
                break;
            case OpKind::Mul: {
                int d = deg.at(op.x) + deg.at(op.y);
                bool aux = d >= 2;
                code.push_back(T_MUL | ((aux ? 1u : 0u) << 8));
                code.push_back(op.x);
                code.push_back(op.y);
                deg.push_back(aux ? 1 : (uint8_t)d);
                break;
            }
            case OpKind::Inv:
            case OpKind::Not: {
                bool aux = deg.at(op.x) != 0;
                code.push_back((op.kind == OpKind::Inv ? T_INV : T_NOT) | ((aux ? 1u : 0u) << 8));
                code.push_back(op.x);
                deg.push_back(aux ? 1 : 0);
                break;
            }
            case OpKind::Call:
            case OpKind::PreImg: {
                const Func& f = t.funcs.at(op.x);
                uint32_t n = op.kind == OpKind::Call ? f.output_size : f.input_size;
                code.push_back(T_CALL | ((f.partial ? 1u : 0u) << 8));
                code.push_back(n);
                deg.insert(deg.end(), n, 1);
                break;
            }
            case OpKind::Store:
                code.push_back(T_STORE);
                deg.push_back(1);
                break;
            case OpKind::Load:
                code.push_back(T_LOAD);
                code.push_back(op.x);
                deg.insert(deg.end(), op.x, 1);
                break;
            case OpKind::ExternCall: {
                const Chip& c = t.chips.at(op.x);
                code.push_back(T_EXTERN);
                code.push_back((uint32_t)c.kind);
                code.push_back((uint32_t)op.a.size());
                code.push_back(c.witness_size);
                code.push_back(c.require_size);
                code.push_back(c.witness_return_size);
                code.insert(code.end(), op.a.begin(), op.a.end());
                // the row's variable map grows by what populate_witness returns (trace.rs:393-396)
                deg.insert(deg.end(), c.witness_return_size, 1);
                break;
            }
            case OpKind::RangeU8:
                code.push_back(T_RANGE_U8);
                code.push_back((uint32_t)((op.a.size() / 2) + (op.a.size() % 2)));
                break;
        }
    }
};

}  // namespace

std::vector<uint32_t> build_trace_program(const Toplevel& t, const Func& f, uint32_t* max_vars) {
    LayoutSizes ls = compute_layout_sizes(t, f);
    Emitter em(t);
    em.code.assign(TH_WORDS, 0);
    std::vector<uint8_t> degrees(f.input_size, 1);
    uint32_t entry = em.emit_block(f.body, degrees);
    em.code[TH_MAGIC] = TRACE_PROGRAM_MAGIC;
    em.code[TH_WIDTH] = ls.total();
    em.code[TH_INPUT] = ls.input;
    em.code[TH_OUTPUT] = ls.output;
    em.code[TH_AUX] = ls.aux;
    em.code[TH_SEL] = ls.sel;
    em.code[TH_PARTIAL] = f.partial ? 1 : 0;
    em.code[TH_ENTRY] = entry;
    em.code[TH_MAX_VARS] = em.max_vars;
    const uint64_t hash = trace_program_hash(em.code.data(), em.code.size());
    em.code[TH_HASH_LO] = (uint32_t)hash;
    em.code[TH_HASH_HI] = (uint32_t)(hash >> 32);
    if (max_vars) *max_vars = em.max_vars;
    return em.code;
}

}  // namespace lair
