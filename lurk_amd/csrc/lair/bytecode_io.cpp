// Compiled-toplevel exchange: a flat u32 serialisation of the index-based Lair bytecode
// (/root/reference/src/lair/bytecode.rs:12-146: Op / Block / Ctrl / Cases / Func) so that a host which has its own
// compiler -- the reference's Rust `Toplevel::new` (/root/reference/src/lair/toplevel.rs:38-72) -- can hand its compiled
// functions over the C ABI instead of source text (include/lurkhip.h: lurkhip_toplevel_from_bytecode), and so that two
// compilers can be compared word for word (lurkhip_toplevel_to_bytecode).
//
// Format "LBC1" (all words little-endian uint32; field elements canonical):
//   blob   := 0x3143424C, version = 1, n_chips, n_funcs, STR chip_name * n_chips, FUNC * n_funcs
//   STR    := byte length, then ceil(len / 4) words holding the UTF-8 bytes (little-endian, zero padded)
//   LIST   := n, then n words
//   FUNC   := STR name, flags (bit 0 invertible, bit 1 partial), input_size, output_size, BLOCK      (index = position)
//   BLOCK  := n_ops, OP * n_ops, CTRL, LIST return_idents
//   OP     := tag, payload; tags in the declaration order of bytecode.rs:12-60
//             0 AssertEq  LIST a, LIST b      (the error formatter is host-only and not carried)
//             1 AssertNe  LIST a, LIST b
//             2 Contains  LIST array, needle
//             3 Const     value
//             4 Add x y | 5 Sub x y | 6 Mul x y | 7 Inv x | 8 Not x
//             9 Call      func index, LIST args
//            10 PreImg    func index, LIST outputs (the callee's outputs, whose preimage is looked up)
//            11 Store     LIST values
//            12 Load      len, pointer
//            13 ExternCall chip index, LIST args
//            14 Emit      LIST values
//            15 RangeU8   LIST values
//            16 Breakpoint
//            17 Debug     STR message
//   CTRL   := 0 (Return)     ident, LIST vars
//           | 1 (Choose)     var, n_unique, BLOCK * n_unique, n_keys, (key, unique index) * n_keys  [ascending keys],
//                            has_default, [BLOCK default]
//           | 2 (ChooseMany) LIST vars, n_branches, (LIST key, BLOCK) * n_branches  [ascending keys], has_default, [BLOCK default]
// `Choose` carries each source branch once (the third member of Ctrl::Choose, bytecode.rs:76) and the key -> branch map as
// indices into that list, instead of the reference's per-key clones of the block.
// Chip names are resolved against the native chips of /root/reference/src/core/chipset.rs:28-63; an unknown name is an error
// (a chip this library has no witness generator and no AIR for cannot be proved).
//
// The importer validates everything the interpreter, the layout pass and the device programs index with: stack references,
// callee / chip indices and arities, return sizes and selector numbering -- a malformed blob is an error, not a crash.
#include <algorithm>
#include <functional>

#include "lair.h"

namespace lair {

namespace {

constexpr uint32_t MAGIC = 0x3143424Cu;  // "LBC1"
constexpr uint32_t VERSION = 1;

struct Writer {
    std::vector<uint32_t> w;
    void u(uint32_t v) { w.push_back(v); }
    void str(const std::string& s) {
        u((uint32_t)s.size());
        for (size_t i = 0; i < s.size(); i += 4) {
            uint32_t v = 0;
            for (size_t k = 0; k < 4 && i + k < s.size(); k++) v |= (uint32_t)(uint8_t)s[i + k] << (8 * k);
            u(v);
        }
    }
    void list(const std::vector<uint32_t>& l) {
        u((uint32_t)l.size());
        w.insert(w.end(), l.begin(), l.end());
    }
};

void write_block(Writer& o, const Block& b);

void write_op(Writer& o, const Op& op) {
    o.u((uint32_t)op.kind);
    switch (op.kind) {
        case OpKind::AssertEq:
        case OpKind::AssertNe:
            o.list(op.a);
            o.list(op.b);
            break;
        case OpKind::Contains:
            o.list(op.a);
            o.u(op.y);
            break;
        case OpKind::Const:
            o.u(op.c);
            break;
        case OpKind::Add:
        case OpKind::Sub:
        case OpKind::Mul:
            o.u(op.x);
            o.u(op.y);
            break;
        case OpKind::Inv:
        case OpKind::Not:
            o.u(op.x);
            break;
        case OpKind::Call:
        case OpKind::PreImg:
        case OpKind::ExternCall:
            o.u(op.x);
            o.list(op.a);
            break;
        case OpKind::Store:
        case OpKind::Emit:
        case OpKind::RangeU8:
            o.list(op.a);
            break;
        case OpKind::Load:
            o.u(op.x);
            o.u(op.y);
            break;
        case OpKind::Breakpoint:
            break;
        case OpKind::Debug:
            o.str("");
            break;
    }
}

void write_ctrl(Writer& o, const Ctrl& c) {
    o.u((uint32_t)c.kind == (uint32_t)Ctrl::Return ? 0u : (c.kind == Ctrl::Choose ? 1u : 2u));
    if (c.kind == Ctrl::Return) {
        o.u(c.ident);
        o.list(c.ret);
    } else if (c.kind == Ctrl::Choose) {
        o.u(c.var);
        o.u((uint32_t)c.unique_branches.size());
        for (const auto& b : c.unique_branches) write_block(o, *b);
        o.u((uint32_t)c.branches.size());
        for (const auto& kv : c.branches) {
            o.u(kv.first.at(0));
            size_t idx = 0;
            while (idx < c.unique_branches.size() && c.unique_branches[idx].get() != kv.second.get()) idx++;
            if (idx == c.unique_branches.size()) throw ExecError("Choose branch is not one of its unique branches");
            o.u((uint32_t)idx);
        }
        o.u(c.def ? 1u : 0u);
        if (c.def) write_block(o, *c.def);
    } else {
        o.list(c.vars);
        o.u((uint32_t)c.branches.size());
        for (const auto& kv : c.branches) {
            o.list(kv.first);
            write_block(o, *kv.second);
        }
        o.u(c.def ? 1u : 0u);
        if (c.def) write_block(o, *c.def);
    }
}

void write_block(Writer& o, const Block& b) {
    o.u((uint32_t)b.ops.size());
    for (const auto& op : b.ops) write_op(o, op);
    write_ctrl(o, b.ctrl);
    o.list(b.return_idents);
}

struct Reader {
    const uint32_t* p;
    size_t n, pos = 0;
    int depth = 0;
    uint32_t u() {
        if (pos >= n) throw ParseError("bytecode blob is truncated");
        return p[pos++];
    }
    // a count that the rest of the blob must be able to hold (every element takes at least `min_words` words)
    uint32_t count(size_t min_words = 1) {
        uint32_t c = u();
        if ((size_t)c * min_words > n - pos) throw ParseError("bytecode blob: count runs past the end");
        return c;
    }
    std::string str() {
        uint32_t len = u();
        size_t words = ((size_t)len + 3) / 4;
        if (words > n - pos) throw ParseError("bytecode blob: string runs past the end");
        std::string s(len, '\0');
        for (uint32_t i = 0; i < len; i++) s[i] = (char)((p[pos + i / 4] >> (8 * (i % 4))) & 0xff);
        pos += words;
        return s;
    }
    std::vector<uint32_t> list() {
        uint32_t c = count();
        std::vector<uint32_t> l(p + pos, p + pos + c);
        pos += c;
        return l;
    }
};

std::shared_ptr<Block> read_block(Reader& r);

Op read_op(Reader& r) {
    Op op;
    uint32_t tag = r.u();
    if (tag > (uint32_t)OpKind::Debug) throw ParseError("bytecode blob: unknown op tag " + std::to_string(tag));
    op.kind = (OpKind)tag;
    switch (op.kind) {
        case OpKind::AssertEq:
        case OpKind::AssertNe:
            op.a = r.list();
            op.b = r.list();
            break;
        case OpKind::Contains:
            op.a = r.list();
            op.y = r.u();
            break;
        case OpKind::Const:
            op.c = r.u();
            break;
        case OpKind::Add:
        case OpKind::Sub:
        case OpKind::Mul:
            op.x = r.u();
            op.y = r.u();
            break;
        case OpKind::Inv:
        case OpKind::Not:
            op.x = r.u();
            break;
        case OpKind::Call:
        case OpKind::PreImg:
        case OpKind::ExternCall:
            op.x = r.u();
            op.a = r.list();
            break;
        case OpKind::Store:
        case OpKind::Emit:
        case OpKind::RangeU8:
            op.a = r.list();
            break;
        case OpKind::Load:
            op.x = r.u();
            op.y = r.u();
            break;
        case OpKind::Breakpoint:
            break;
        case OpKind::Debug:
            (void)r.str();
            break;
    }
    return op;
}

Ctrl read_ctrl(Reader& r) {
    Ctrl c;
    uint32_t tag = r.u();
    if (tag == 0) {
        c.kind = Ctrl::Return;
        c.ident = r.u();
        c.ret = r.list();
    } else if (tag == 1) {
        c.kind = Ctrl::Choose;
        c.var = r.u();
        uint32_t nu = r.count(4);
        for (uint32_t i = 0; i < nu; i++) c.unique_branches.push_back(read_block(r));
        uint32_t nk = r.count(2);
        for (uint32_t i = 0; i < nk; i++) {
            uint32_t key = r.u(), idx = r.u();
            if (idx >= nu) throw ParseError("bytecode blob: Choose key points past the branch list");
            if (!c.branches.empty() && !(c.branches.back().first[0] < key)) throw ParseError("bytecode blob: Choose keys must ascend");
            c.branches.push_back({List{key}, c.unique_branches[idx]});
        }
        if (r.u()) c.def = read_block(r);
    } else if (tag == 2) {
        c.kind = Ctrl::ChooseMany;
        c.vars = r.list();
        uint32_t nb = r.count(5);
        for (uint32_t i = 0; i < nb; i++) {
            List key = r.list();
            if (key.size() != c.vars.size()) throw ParseError("bytecode blob: ChooseMany pattern size mismatch");
            if (!c.branches.empty() && !(c.branches.back().first < key)) throw ParseError("bytecode blob: ChooseMany keys must ascend");
            auto blk = read_block(r);
            c.branches.push_back({std::move(key), blk});
        }
        if (r.u()) c.def = read_block(r);
    } else {
        throw ParseError("bytecode blob: unknown ctrl tag " + std::to_string(tag));
    }
    return c;
}

std::shared_ptr<Block> read_block(Reader& r) {
    if (++r.depth > 4096) throw ParseError("bytecode blob: blocks nest too deeply");
    auto b = std::make_shared<Block>();
    uint32_t n_ops = r.count();
    b->ops.reserve(n_ops);
    for (uint32_t i = 0; i < n_ops; i++) b->ops.push_back(read_op(r));
    b->ctrl = read_ctrl(r);
    b->return_idents = r.list();
    r.depth--;
    return b;
}

// what the interpreter, the layout pass and the emitters take for granted about a compiled function
struct Validator {
    const Toplevel& t;
    const Func& f;
    // Return selectors: a permutation of 0 .. n_returns-1.  NOT required to ascend in the order the blocks are stored: the
    // reference numbers them in SOURCE order (toplevel.rs:531-540) and then sorts a ChooseMany's arms by key
    // (toplevel.rs:557-570, map.rs:16-22), so a `match` over an array whose arms are not written in ascending key order has
    // its selectors out of storage order.
    std::vector<bool> ident_seen;
    void bad(const std::string& m) const { throw ParseError("bytecode of " + f.name + ": " + m); }
    void refs(const std::vector<uint32_t>& l, uint64_t height) const {
        for (uint32_t v : l)
            if (v >= height) bad("stack reference " + std::to_string(v) + " above the stack height " + std::to_string(height));
    }
    // returns the return idents of the block, in order
    static bool has_mem_table(size_t len) { return len == 2 || len == 3 || len == 4 || len == 5 || len == 6 || len == 8; }
    // the stack height is tracked in 64 bits with a bound: a crafted blob must not wrap it past the reference checks
    static constexpr uint64_t MAX_HEIGHT = 1u << 24;
    std::vector<uint32_t> block(const Block& b, uint64_t height) {
        if (height > MAX_HEIGHT) bad("stack height out of range");
        for (const Op& op : b.ops) {
            if (height > MAX_HEIGHT) bad("stack height out of range");
            switch (op.kind) {
                case OpKind::AssertEq:
                case OpKind::AssertNe:
                    if (op.a.size() != op.b.size() || op.a.empty()) bad("assert operands differ in size");
                    refs(op.a, height);
                    refs(op.b, height);
                    break;
                case OpKind::Contains:
                    if (op.a.empty()) bad("contains over an empty array");
                    refs(op.a, height);
                    refs({op.y}, height);
                    break;
                case OpKind::Const:
                    if (op.c >= P) bad("constant is not a canonical field element");
                    height += 1;
                    break;
                case OpKind::Add:
                case OpKind::Sub:
                case OpKind::Mul:
                    refs({op.x, op.y}, height);
                    height += 1;
                    break;
                case OpKind::Inv:
                case OpKind::Not:
                    refs({op.x}, height);
                    height += 1;
                    break;
                case OpKind::Call:
                case OpKind::PreImg: {
                    if (op.x >= t.funcs.size()) bad("callee index out of range");
                    const Func& g = t.funcs[op.x];
                    const bool call = op.kind == OpKind::Call;
                    if (op.a.size() != (call ? g.input_size : g.output_size)) bad("wrong number of arguments for " + g.name);
                    if (g.partial && !f.partial) bad("the partial " + g.name + " called from a non-partial function");
                    if (!call && !g.invertible) bad("preimage of " + g.name + ", which is not invertible");
                    refs(op.a, height);
                    height += call ? g.output_size : g.input_size;
                    break;
                }
                case OpKind::Store:
                    if (!has_mem_table(op.a.size())) bad("store of " + std::to_string(op.a.size()) + " values: memory tables exist for 2, 3, 4, 5, 6 and 8");
                    refs(op.a, height);
                    height += 1;
                    break;
                case OpKind::Load:
                    if (!has_mem_table(op.x)) bad("load of " + std::to_string(op.x) + " values: memory tables exist for 2, 3, 4, 5, 6 and 8");
                    refs({op.y}, height);
                    height += op.x;
                    break;
                case OpKind::ExternCall: {
                    if (op.x >= t.chips.size()) bad("chip index out of range");
                    const Chip& c = t.chips[op.x];
                    if (op.a.size() != c.input_size) bad("wrong number of arguments for chip " + c.name);
                    refs(op.a, height);
                    height += c.output_size;
                    break;
                }
                case OpKind::Emit:
                case OpKind::RangeU8:
                    refs(op.a, height);
                    break;
                case OpKind::Breakpoint:
                case OpKind::Debug:
                    break;
            }
        }
        std::vector<uint32_t> idents;
        const Ctrl& c = b.ctrl;
        if (c.kind == Ctrl::Return) {
            if (c.ret.size() != f.output_size) bad("return size differs from the declared output size");
            refs(c.ret, height);
            if (c.ident >= (1u << 24)) bad("implausible return selector");
            if (ident_seen.size() <= c.ident) ident_seen.resize((size_t)c.ident + 1, false);
            if (ident_seen[c.ident]) bad("two returns share a selector");
            ident_seen[c.ident] = true;
            idents.push_back(c.ident);
        } else {
            if (c.kind == Ctrl::Choose) {
                refs({c.var}, height);
                for (const auto& kv : c.branches)
                    if (kv.first[0] >= P) bad("match key is not a canonical field element");
                for (const auto& blk : c.unique_branches) {
                    bool referenced = false;
                    for (const auto& kv : c.branches) referenced = referenced || kv.second == blk;
                    if (!referenced) bad("a unique branch that no key selects");
                    auto sub = block(*blk, height);
                    idents.insert(idents.end(), sub.begin(), sub.end());
                }
            } else {
                if (c.vars.empty()) bad("ChooseMany over no variables");
                refs(c.vars, height);
                for (const auto& kv : c.branches) {
                    for (uint32_t k : kv.first)
                        if (k >= P) bad("match key is not a canonical field element");
                    auto sub = block(*kv.second, height);
                    idents.insert(idents.end(), sub.begin(), sub.end());
                }
            }
            if (c.def) {
                auto sub = block(*c.def, height);
                idents.insert(idents.end(), sub.begin(), sub.end());
            }
            if (idents.empty()) bad("a block must have at least one return");
        }
        {
            std::vector<uint32_t> have = idents, want = b.return_idents;
            std::sort(have.begin(), have.end());
            std::sort(want.begin(), want.end());
            if (have != want) bad("return_idents do not list the block's returns");
        }
        return idents;
    }
};

}  // namespace

std::vector<uint32_t> toplevel_to_bytecode(const Toplevel& t) {
    Writer o;
    o.u(MAGIC);
    o.u(VERSION);
    o.u((uint32_t)t.chips.size());
    o.u((uint32_t)t.funcs.size());
    for (const auto& c : t.chips) o.str(c.name);
    for (const auto& f : t.funcs) {
        o.str(f.name);
        o.u((f.invertible ? 1u : 0u) | (f.partial ? 2u : 0u));
        o.u(f.input_size);
        o.u(f.output_size);
        write_block(o, f.body);
    }
    return std::move(o.w);
}

Toplevel toplevel_from_bytecode(const uint32_t* words, size_t n_words) {
    Reader r{words, n_words};
    if (r.u() != MAGIC) throw ParseError("bytecode blob: bad magic (expected \"LBC1\")");
    if (r.u() != VERSION) throw ParseError("bytecode blob: unsupported format version");
    const uint32_t n_chips = r.count(), n_funcs = r.count(5);
    Toplevel t;
    const std::vector<Chip> native = lurk_chip_map();
    for (uint32_t i = 0; i < n_chips; i++) {
        std::string name = r.str();
        auto it = std::find_if(native.begin(), native.end(), [&](const Chip& c) { return c.name == name; });
        if (it == native.end()) throw ParseError("bytecode blob: no native chip named " + name);
        if (t.chip_index.count(name)) throw ParseError("bytecode blob: duplicate chip " + name);
        t.chip_index[name] = i;
        t.chips.push_back(*it);
    }
    for (uint32_t i = 0; i < n_funcs; i++) {
        Func f;
        f.name = r.str();
        uint32_t flags = r.u();
        if (flags > 3) throw ParseError("bytecode blob: unknown function flags");
        f.invertible = flags & 1;
        f.partial = flags & 2;
        f.index = i;
        f.input_size = r.u();
        f.output_size = r.u();
        if (f.input_size > (1u << 16) || f.output_size > (1u << 16)) throw ParseError("bytecode blob: implausible function arity");
        f.body = *read_block(r);
        if (t.func_index.count(f.name)) throw ParseError("duplicate function " + f.name);
        t.func_index[f.name] = i;
        t.funcs.push_back(std::move(f));
    }
    if (r.pos != r.n) throw ParseError("bytecode blob: trailing words");
    for (const auto& f : t.funcs) {
        Validator v{t, f, {}};
        v.block(f.body, f.input_size);
        for (bool seen : v.ident_seen)
            if (!seen) v.bad("return selectors are not 0 .. n_returns-1");
    }
    return t;
}

}  // namespace lair
