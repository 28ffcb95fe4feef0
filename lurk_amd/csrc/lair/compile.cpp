// Lair IR -> bytecode (expand + compile) and the per-function column layout.
//
// Follows /root/reference/src/lair/toplevel.rs:241-879 (expand: If/Match/MatchMany -> Choose /
// ChooseMany with their assertion prologues, Div -> Inv+Mul, Eq -> Sub+Not; compile: variables ->
// stack indices, return idents in compile order, branch state save/restore) and
// /root/reference/src/lair/func_chip.rs:90-276 (layout sizes, shared aux columns across branches).
#include <algorithm>
#include <functional>

#include "lair.h"

namespace lair {

uint32_t finv(uint32_t a) {
    if (a % P == 0) throw ExecError("inverse of zero");
    uint64_t r = 1, b = a % P;
    uint32_t e = P - 2;
    while (e) {
        if (e & 1) r = r * b % P;
        b = b * b % P;
        e >>= 1;
    }
    return (uint32_t)r;
}

const uint32_t MEM_TABLE_SIZES[NUM_MEM_TABLES] = {2, 3, 4, 5, 6, 8};
int mem_index_from_len(uint32_t len) {
    for (int i = 0; i < NUM_MEM_TABLES; i++)
        if (MEM_TABLE_SIZES[i] == len) return i;
    throw ExecError("There are no mem tables of size " + std::to_string(len));
}

const Block* Ctrl::match_case(const List& key) const {
    auto it = std::lower_bound(branches.begin(), branches.end(), key,
                               [](const std::pair<List, std::shared_ptr<Block>>& e, const List& k) { return e.first < k; });
    if (it != branches.end() && it->first == key) return it->second.get();
    return def.get();
}

namespace {

// ------------------------------------------------------------------ expand (toplevel.rs:292-301,423-527,739-753)
struct ExpandCtx {
    int uniq = 0;
    Var new_var(int size) { return Var{"$" + std::to_string(uniq++), size}; }
};

std::shared_ptr<BlockE> expand_block(const BlockE& b, std::vector<OpE> ops, ExpandCtx& ctx);

OpE mk(OpEKind k, std::vector<Var> out, std::vector<Var> in, List consts = {}) {
    OpE o;
    o.kind = k;
    o.out = std::move(out);
    o.in = std::move(in);
    o.consts = std::move(consts);
    return o;
}

void expand_op(const OpE& op, std::vector<OpE>& ops, ExpandCtx& ctx) {
    if (op.kind == OpEKind::Div) {
        Var inv = ctx.new_var(op.in[1].size);
        ops.push_back(mk(OpEKind::Inv, {inv}, {op.in[1]}));
        ops.push_back(mk(OpEKind::Mul, op.out, {op.in[0], inv}));
    } else if (op.kind == OpEKind::Eq) {
        Var ne = ctx.new_var(op.in[0].size);
        ops.push_back(mk(OpEKind::Sub, {ne}, {op.in[0], op.in[1]}));
        ops.push_back(mk(OpEKind::Not, op.out, {ne}));
    } else {
        ops.push_back(op);
    }
}

CtrlE expand_ctrl(const CtrlE& c, ExpandCtx& ctx) {
    CtrlE out;
    switch (c.kind) {
        case CtrlEKind::Return:
            return c;
        case CtrlEKind::If: {
            const Var& x = c.var;
            Var zero = ctx.new_var(x.size);
            List arr((size_t)x.size, 0u);
            std::vector<OpE> tops = {mk(OpEKind::Array, {zero}, {}, arr), mk(OpEKind::AssertNe, {}, {x, zero})};
            auto t = expand_block(*c.t, tops, ctx);
            std::vector<OpE> fops = {mk(OpEKind::Array, {zero}, {}, arr), mk(OpEKind::AssertEq, {}, {x, zero})};
            auto f = expand_block(*c.f, fops, ctx);
            out.kind = x.size == 1 ? CtrlEKind::Choose : CtrlEKind::ChooseMany;
            out.var = x;
            out.branches.push_back(CaseE{arr, f, CaseType::Constrained});
            out.def = t;
            return out;
        }
        case CtrlEKind::Match: {
            out.kind = CtrlEKind::Choose;
            out.var = c.var;
            for (const auto& br : c.branches) {
                std::vector<OpE> ops;
                if (br.constrained == CaseType::Constrained) {
                    Var arr = ctx.new_var((int)br.keys.size());
                    ops.push_back(mk(OpEKind::Array, {arr}, {}, br.keys));
                    ops.push_back(mk(OpEKind::Contains, {}, {arr, c.var}));
                }
                out.branches.push_back(CaseE{br.keys, expand_block(*br.block, ops, ctx), br.constrained});
            }
            if (c.def) {
                std::vector<OpE> ops;
                if (c.def_constrained == CaseType::Constrained) {
                    for (const auto& br : c.branches)
                        for (uint32_t f : br.keys) {
                            Var fv = ctx.new_var(1);
                            ops.push_back(mk(OpEKind::Const, {fv}, {}, {f}));
                            ops.push_back(mk(OpEKind::AssertNe, {}, {c.var, fv}));
                        }
                }
                out.def = expand_block(*c.def, ops, ctx);
            }
            return out;
        }
        case CtrlEKind::MatchMany: {
            out.kind = CtrlEKind::ChooseMany;
            out.var = c.var;
            for (const auto& br : c.branches) {
                std::vector<OpE> ops;
                if (br.constrained == CaseType::Constrained) {
                    Var arr = ctx.new_var((int)br.keys.size());
                    ops.push_back(mk(OpEKind::Array, {arr}, {}, br.keys));
                    ops.push_back(mk(OpEKind::AssertEq, {}, {c.var, arr}));
                }
                out.branches.push_back(CaseE{br.keys, expand_block(*br.block, ops, ctx), br.constrained});
            }
            if (c.def) {
                std::vector<OpE> ops;
                if (c.def_constrained == CaseType::Constrained) {
                    for (const auto& br : c.branches) {
                        Var arr = ctx.new_var((int)br.keys.size());
                        ops.push_back(mk(OpEKind::Array, {arr}, {}, br.keys));
                        ops.push_back(mk(OpEKind::AssertNe, {}, {c.var, arr}));
                    }
                }
                out.def = expand_block(*c.def, ops, ctx);
            }
            return out;
        }
        case CtrlEKind::Choose:
        case CtrlEKind::ChooseMany: {
            out.kind = c.kind;
            out.var = c.var;
            for (const auto& br : c.branches) out.branches.push_back(CaseE{br.keys, expand_block(*br.block, {}, ctx), br.constrained});
            if (c.def) out.def = expand_block(*c.def, {}, ctx);
            return out;
        }
    }
    return out;
}

std::shared_ptr<BlockE> expand_block(const BlockE& b, std::vector<OpE> ops, ExpandCtx& ctx) {
    auto out = std::make_shared<BlockE>();
    for (const auto& op : b.ops) expand_op(op, ops, ctx);
    out->ops = std::move(ops);
    out->ctrl = expand_ctrl(b.ctrl, ctx);
    return out;
}

// ------------------------------------------------------------------ compile (toplevel.rs:255-283,303-322,529-574,755-879)
struct FuncInfo {
    uint32_t input_size, output_size;
    bool partial;
};

struct LinkCtx {
    uint32_t var_index = 0;
    uint32_t return_ident = 0;
    std::vector<uint32_t> return_idents;
    std::map<std::pair<std::string, int>, std::vector<uint32_t>> link_map;  // (name,size) -> indices
    const std::vector<FuncE>* funcs;
    const std::unordered_map<std::string, uint32_t>* func_index;
    const std::vector<Chip>* chips;
    const std::unordered_map<std::string, uint32_t>* chip_index;
    std::string fname;
    bool partial = false;

    uint32_t new_var() { return var_index++; }
    void link_new(const Var& v) {
        std::vector<uint32_t> idx;
        for (int i = 0; i < v.size; i++) idx.push_back(new_var());
        link_map[{v.name, v.size}] = idx;
    }
    void link(const Var& v, std::vector<uint32_t> idx) { link_map[{v.name, v.size}] = std::move(idx); }
    const std::vector<uint32_t>& get(const Var& v) const {
        auto it = link_map.find({v.name, v.size});
        if (it == link_map.end()) throw ParseError("in " + fname + ": variable " + v.name + " is unbound");
        return it->second;
    }
    std::vector<uint32_t> flat(const std::vector<Var>& vs) const {
        std::vector<uint32_t> r;
        for (const auto& v : vs) {
            const auto& g = get(v);
            r.insert(r.end(), g.begin(), g.end());
        }
        return r;
    }
};

int total_size(const std::vector<Var>& v) {
    int s = 0;
    for (auto& x : v) s += x.size;
    return s;
}

std::shared_ptr<Block> compile_block(const BlockE& b, LinkCtx& ctx);

void compile_op(const OpE& op, std::vector<Op>& ops, LinkCtx& ctx) {
    auto err = [&](const std::string& m) { throw ParseError("in " + ctx.fname + ": " + m); };
    Op o;
    switch (op.kind) {
        case OpEKind::AssertNe:
        case OpEKind::AssertEq:
            if (op.in[0].size != op.in[1].size) err("size mismatch in assertion");
            o.kind = op.kind == OpEKind::AssertNe ? OpKind::AssertNe : OpKind::AssertEq;
            o.a = ctx.get(op.in[0]);
            o.b = ctx.get(op.in[1]);
            ops.push_back(o);
            break;
        case OpEKind::Contains:
            if (op.in[1].size != 1) err("contains! needs a size-1 needle");
            o.kind = OpKind::Contains;
            o.a = ctx.get(op.in[0]);
            o.y = ctx.get(op.in[1])[0];
            ops.push_back(o);
            break;
        case OpEKind::Const:
            o.kind = OpKind::Const;
            o.c = op.consts[0];
            ops.push_back(o);
            ctx.link_new(op.out[0]);
            break;
        case OpEKind::Array:
            for (uint32_t f : op.consts) {
                Op k;
                k.kind = OpKind::Const;
                k.c = f;
                ops.push_back(k);
            }
            ctx.link_new(op.out[0]);
            break;
        case OpEKind::Add:
        case OpEKind::Sub:
        case OpEKind::Mul: {
            if (op.in[0].size != op.in[1].size || op.in[0].size != op.out[0].size) err("size mismatch in arithmetic");
            std::vector<uint32_t> a = ctx.get(op.in[0]), b = ctx.get(op.in[1]);
            for (size_t i = 0; i < a.size(); i++) {
                Op k;
                k.kind = op.kind == OpEKind::Add ? OpKind::Add : op.kind == OpEKind::Sub ? OpKind::Sub : OpKind::Mul;
                k.x = a[i];
                k.y = b[i];
                ops.push_back(k);
            }
            ctx.link_new(op.out[0]);
            break;
        }
        case OpEKind::Inv: {
            std::vector<uint32_t> a = ctx.get(op.in[0]);
            for (uint32_t x : a) {
                Op k;
                k.kind = OpKind::Inv;
                k.x = x;
                ops.push_back(k);
            }
            ctx.link_new(op.out[0]);
            break;
        }
        case OpEKind::Not:
            if (op.in[0].size != 1 || op.out[0].size != 1) err("not needs size-1 operands");
            o.kind = OpKind::Not;
            o.x = ctx.get(op.in[0])[0];
            ops.push_back(o);
            ctx.link_new(op.out[0]);
            break;
        case OpEKind::Call:
        case OpEKind::PreImg: {
            auto it = ctx.func_index->find(op.name);
            if (it == ctx.func_index->end()) err("Unknown function " + op.name);
            const FuncE& callee = (*ctx.funcs)[it->second];
            int in_sz = total_size(callee.input_params), out_sz = callee.output_size;
            if (callee.partial && !ctx.partial) err("partial function " + op.name + " called from a total one");
            if (op.kind == OpEKind::Call) {
                if (total_size(op.in) != in_sz) err("Input mismatch on call to " + op.name);
                if (total_size(op.out) != out_sz) err("Output mismatch on call to " + op.name);
                o.kind = OpKind::Call;
            } else {
                if (total_size(op.out) != in_sz) err("Input mismatch on preimg of " + op.name);
                if (total_size(op.in) != out_sz) err("Output mismatch on preimg of " + op.name);
                o.kind = OpKind::PreImg;
            }
            o.x = it->second;
            o.a = ctx.flat(op.in);
            ops.push_back(o);
            for (const auto& t : op.out) ctx.link_new(t);
            break;
        }
        case OpEKind::Store:
            o.kind = OpKind::Store;
            o.a = ctx.flat(op.in);
            ops.push_back(o);
            ctx.link_new(op.out[0]);
            break;
        case OpEKind::Load:
            o.kind = OpKind::Load;
            o.x = (uint32_t)total_size(op.out);
            o.y = ctx.get(op.in[0])[0];
            ops.push_back(o);
            for (const auto& t : op.out) ctx.link_new(t);
            break;
        case OpEKind::Slice: {
            if (total_size(op.out) != total_size(op.in)) err("size mismatch in slice");
            std::vector<uint32_t> args = ctx.flat(op.in);
            size_t i = 0;
            for (const auto& pat : op.out) {
                ctx.link(pat, std::vector<uint32_t>(args.begin() + i, args.begin() + i + pat.size));
                i += pat.size;
            }
            break;
        }
        case OpEKind::ExternCall: {
            auto it = ctx.chip_index->find(op.name);
            if (it == ctx.chip_index->end()) err("Unknown extern chip " + op.name);
            const Chip& chip = (*ctx.chips)[it->second];
            if ((uint32_t)total_size(op.in) != chip.input_size) err("Input mismatch on extern call " + op.name);
            if ((uint32_t)total_size(op.out) != chip.output_size) err("Output mismatch on extern call " + op.name);
            o.kind = OpKind::ExternCall;
            o.x = it->second;
            o.a = ctx.flat(op.in);
            ops.push_back(o);
            for (const auto& t : op.out) ctx.link_new(t);
            break;
        }
        case OpEKind::Emit:
            o.kind = OpKind::Emit;
            o.a = ctx.flat(op.in);
            ops.push_back(o);
            break;
        case OpEKind::RangeU8:
            o.kind = OpKind::RangeU8;
            o.a = ctx.flat(op.in);
            ops.push_back(o);
            break;
        case OpEKind::Breakpoint:
            o.kind = OpKind::Breakpoint;
            ops.push_back(o);
            break;
        case OpEKind::Debug:
            o.kind = OpKind::Debug;
            ops.push_back(o);
            break;
        case OpEKind::Div:
        case OpEKind::Eq:
            err("Expand first");
    }
}

Ctrl compile_ctrl(const CtrlE& c, LinkCtx& ctx) {
    Ctrl out;
    switch (c.kind) {
        case CtrlEKind::Return:
            out.kind = Ctrl::Return;
            out.ret = ctx.flat(c.ret);
            out.ident = ctx.return_ident;
            ctx.return_idents.push_back(ctx.return_ident);
            ctx.return_ident++;
            return out;
        case CtrlEKind::Choose: {
            out.kind = Ctrl::Choose;
            out.var = ctx.get(c.var)[0];
            for (const auto& br : c.branches) {
                auto saved_index = ctx.var_index;
                auto saved_map = ctx.link_map;
                auto blk = compile_block(*br.block, ctx);
                ctx.var_index = saved_index;
                ctx.link_map = saved_map;
                for (uint32_t f : br.keys) out.branches.push_back({List{f}, blk});
                out.unique_branches.push_back(blk);
            }
            std::stable_sort(out.branches.begin(), out.branches.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            if (c.def) out.def = compile_block(*c.def, ctx);
            return out;
        }
        case CtrlEKind::ChooseMany: {
            out.kind = Ctrl::ChooseMany;
            out.vars = ctx.get(c.var);
            for (const auto& br : c.branches) {
                if (br.keys.size() != out.vars.size()) throw ParseError("in " + ctx.fname + ": pattern size mismatch");
                auto saved_index = ctx.var_index;
                auto saved_map = ctx.link_map;
                auto blk = compile_block(*br.block, ctx);
                ctx.var_index = saved_index;
                ctx.link_map = saved_map;
                out.branches.push_back({br.keys, blk});
            }
            std::stable_sort(out.branches.begin(), out.branches.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            if (c.def) out.def = compile_block(*c.def, ctx);
            return out;
        }
        default:
            throw ParseError("Expand first");
    }
}

std::shared_ptr<Block> compile_block(const BlockE& b, LinkCtx& ctx) {
    auto out = std::make_shared<Block>();
    for (const auto& op : b.ops) compile_op(op, out->ops, ctx);
    std::vector<uint32_t> saved;
    saved.swap(ctx.return_idents);
    out->ctrl = compile_ctrl(b.ctrl, ctx);
    std::vector<uint32_t> mine;
    mine.swap(ctx.return_idents);
    if (mine.empty()) throw ParseError("A block must have at least one return ident");
    ctx.return_idents = saved;
    ctx.return_idents.insert(ctx.return_idents.end(), mine.begin(), mine.end());
    out->return_idents = mine;
    return out;
}

// ------------------------------------------------------------------ layout (func_chip.rs:118-276)
using Degree = uint8_t;

void layout_block(const Toplevel& t, const Block& b, std::vector<Degree>& degrees, uint32_t& aux, uint32_t& sel);

void layout_op(const Toplevel& t, const Op& op, std::vector<Degree>& degrees, uint32_t& aux) {
    switch (op.kind) {
        case OpKind::AssertEq:
            break;
        case OpKind::AssertNe:
            aux += (uint32_t)op.a.size();
            break;
        case OpKind::Contains:
            aux += (uint32_t)op.a.size() - 1;
            break;
        case OpKind::Const:
            degrees.push_back(0);
            break;
        case OpKind::Add:
        case OpKind::Sub:
            degrees.push_back(std::max(degrees.at(op.x), degrees.at(op.y)));
            break;
        case OpKind::Mul: {
            int deg = degrees.at(op.x) + degrees.at(op.y);
            if (deg < 2) degrees.push_back((Degree)deg);
            else {
                degrees.push_back(1);
                aux += 1;
            }
            break;
        }
        case OpKind::Inv:
            if (degrees.at(op.x) == 0) degrees.push_back(0);
            else {
                degrees.push_back(1);
                aux += 1;
            }
            break;
        case OpKind::Not:
            if (degrees.at(op.x) == 0) degrees.push_back(0);
            else {
                degrees.push_back(1);
                aux += 2;
            }
            break;
        case OpKind::Call:
        case OpKind::PreImg: {
            const Func& f = t.funcs.at(op.x);
            uint32_t n = op.kind == OpKind::Call ? f.output_size : f.input_size;
            aux += n + 3;
            if (f.partial) aux += DEPTH_W + DEPTH_LESS_THAN_SIZE + 3 * DEPTH_LT_REQUIRES;
            degrees.insert(degrees.end(), n, 1);
            break;
        }
        case OpKind::Store:
            aux += 4;
            degrees.push_back(1);
            break;
        case OpKind::Load:
            aux += op.x + 3;
            degrees.insert(degrees.end(), op.x, 1);
            break;
        case OpKind::ExternCall: {
            const Chip& c = t.chips.at(op.x);
            uint32_t aux_size = c.witness_size + c.require_size * 3;
            aux += aux_size;
            // the reference extends `degrees` by aux_size entries, not output_size (func_chip.rs:262-268)
            degrees.insert(degrees.end(), aux_size, 1);
            break;
        }
        case OpKind::RangeU8:
            aux += 3 * (uint32_t)((op.a.size() / 2) + (op.a.size() % 2));
            break;
        case OpKind::Emit:
        case OpKind::Breakpoint:
        case OpKind::Debug:
            break;
    }
}

void layout_ctrl(const Toplevel& t, const Ctrl& c, std::vector<Degree>& degrees, uint32_t& aux, uint32_t& sel) {
    if (c.kind == Ctrl::Return) {
        sel += 1;
        return;
    }
    size_t dlen = degrees.size();
    uint32_t max_aux = aux;
    auto process = [&](const Block& b) {
        uint32_t block_aux = aux;
        layout_block(t, b, degrees, block_aux, sel);
        degrees.resize(dlen);
        max_aux = std::max(max_aux, block_aux);
    };
    if (c.kind == Ctrl::Choose) {
        for (const auto& b : c.unique_branches) process(*b);
    } else {
        for (const auto& kv : c.branches) process(*kv.second);
    }
    if (c.def) process(*c.def);
    aux = max_aux;
}

void layout_block(const Toplevel& t, const Block& b, std::vector<Degree>& degrees, uint32_t& aux, uint32_t& sel) {
    for (const auto& op : b.ops) layout_op(t, op, degrees, aux);
    layout_ctrl(t, b.ctrl, degrees, aux, sel);
}

}  // namespace

Toplevel Toplevel::build(const std::vector<FuncE>& funcs_e, const std::vector<Chip>& chips) {
    Toplevel t;
    t.chips = chips;
    for (uint32_t i = 0; i < chips.size(); i++) t.chip_index[chips[i].name] = i;
    for (uint32_t i = 0; i < funcs_e.size(); i++) {
        if (t.func_index.count(funcs_e[i].name)) throw ParseError("duplicate function " + funcs_e[i].name);
        t.func_index[funcs_e[i].name] = i;
    }
    for (uint32_t i = 0; i < funcs_e.size(); i++) {
        const FuncE& fe = funcs_e[i];
        ExpandCtx ectx;
        auto body = expand_block(fe.body, {}, ectx);
        LinkCtx ctx;
        ctx.funcs = &funcs_e;
        ctx.func_index = &t.func_index;
        ctx.chips = &t.chips;
        ctx.chip_index = &t.chip_index;
        ctx.fname = fe.name;
        ctx.partial = fe.partial;
        for (const auto& v : fe.input_params) ctx.link_new(v);
        Func f;
        f.name = fe.name;
        f.invertible = fe.invertible;
        f.partial = fe.partial;
        f.index = i;
        f.input_size = (uint32_t)total_size(fe.input_params);
        f.output_size = (uint32_t)fe.output_size;
        // check the declared return size on every Return (toplevel.rs:328-335)
        std::function<void(const BlockE&)> chk = [&](const BlockE& b) {
            if (b.ctrl.kind == CtrlEKind::Return) {
                if (total_size(b.ctrl.ret) != fe.output_size)
                    throw ParseError("in " + fe.name + ": Return size " + std::to_string(total_size(b.ctrl.ret)) +
                                     " different from expected size of return " + std::to_string(fe.output_size));
                return;
            }
            for (const auto& br : b.ctrl.branches) chk(*br.block);
            if (b.ctrl.def) chk(*b.ctrl.def);
        };
        chk(*body);
        f.body = *compile_block(*body, ctx);
        t.funcs.push_back(std::move(f));
    }
    return t;
}

const Func& Toplevel::func_by_name(const std::string& n) const {
    auto it = func_index.find(n);
    if (it == func_index.end()) throw ExecError("Func " + n + " not found");
    return funcs[it->second];
}

LayoutSizes compute_layout_sizes(const Toplevel& t, const Func& f) {
    LayoutSizes s;
    s.input = f.input_size;
    s.output = f.output_size;
    uint32_t aux = 2;  // last nonce, last count
    if (f.partial) {
        uint32_t num_requires = (DEPTH_W / 2) + (DEPTH_W % 2);
        aux += DEPTH_W + 3 * num_requires;
    }
    uint32_t sel = 0;
    std::vector<Degree> degrees(f.input_size, 1);
    layout_block(t, f.body, degrees, aux, sel);
    s.aux = aux;
    s.sel = sel;
    return s;
}

}  // namespace lair
