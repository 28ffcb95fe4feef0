// The memoising Lair interpreter and the host side of the extern chips.
//
// Follows /root/reference/src/lair/execute.rs:375-392,436-784 (explicit exec-entry stack and caller
// stack, IndexMap nonces, `provide.count = 1` for the top-level query, depth bookkeeping for partial
// functions) and the `execute` halves of /root/reference/src/core/{poseidon,u64}.rs.
//
// One addition to the reference's QueryResult: while a query executes, every value that
// `populate_row` would later re-derive through hash-map lookups (callee outputs, preimages, store
// pointers, loaded values, callee depths) is appended to `hints` in bytecode order.  The device trace
// kernel then needs no lookups at all (DESIGN.md "row stream").
#include <algorithm>
#include <cstring>
#include <deque>
#include <new>

#include <sys/mman.h>

#include "../babybear.h"
#include "../p2_params.h"
#include "lair.h"

namespace lair {

// ------------------------------------------------------------------ huge-page allocator (lair.h: HugeAlloc)
namespace {
constexpr size_t HUGE_MIN = (size_t)4 << 20, HUGE_PAGE = (size_t)2 << 20;
}
void* huge_alloc(size_t bytes) {
    if (bytes < HUGE_MIN) return ::operator new(bytes);
    const size_t len = (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
    // over-map by one huge page, keep the aligned part
    char* base = (char*)mmap(nullptr, len + HUGE_PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) throw std::bad_alloc();
    char* p = (char*)(((uintptr_t)base + HUGE_PAGE - 1) & ~(uintptr_t)(HUGE_PAGE - 1));
    if (p > base) munmap(base, (size_t)(p - base));
    const size_t tail = (size_t)(base + len + HUGE_PAGE - (p + len));
    if (tail) munmap(p + len, tail);
    (void)madvise(p, len, MADV_HUGEPAGE);  // advisory: small pages if the kernel declines
    return p;
}
void huge_free(void* p, size_t bytes) {
    if (bytes < HUGE_MIN) {
        ::operator delete(p);
        return;
    }
    munmap(p, (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1));
}

// ------------------------------------------------------------------ byte records (gadgets/bytes/record.rs:112-158)
void BytesRecord::range_check_u8_pair(uint8_t i1, uint8_t i2, uint32_t nonce, std::vector<Record>& rq) {
    uint16_t key = (uint16_t)(i1 | (i2 << 8));
    rq.push_back(at(key).range_u8.new_lookup(nonce));
}
void BytesRecord::range_check_u8_iter(const uint8_t* b, size_t n, uint32_t nonce, std::vector<Record>& rq) {
    for (size_t i = 0; i < n; i += 2) range_check_u8_pair(b[i], i + 1 < n ? b[i + 1] : 0, nonce, rq);
}
bool BytesRecord::less_than(uint8_t i1, uint8_t i2, uint32_t nonce, std::vector<Record>& rq) {
    uint16_t key = (uint16_t)(i1 | (i2 << 8));
    rq.push_back(at(key).less_than.new_lookup(nonce));
    return i1 < i2;
}
void BytesRecord::range_check_u16(uint16_t v, uint32_t nonce, std::vector<Record>& rq) {
    rq.push_back(at(v).range_u16.new_lookup(nonce));
}

// ------------------------------------------------------------------ host Poseidon2 (for `execute` only)
namespace {

void host_external_layer(int w, uint32_t* s) {
    for (int i = 0; i < w; i += 4) {
        uint32_t x0 = s[i], x1 = s[i + 1], x2 = s[i + 2], x3 = s[i + 3];
        uint32_t t01 = bb::add(x0, x1), t23 = bb::add(x2, x3), t0123 = bb::add(t01, t23);
        uint32_t t01123 = bb::add(t0123, x1), t01233 = bb::add(t0123, x3);
        s[i + 3] = bb::add(t01233, bb::dbl(x0));
        s[i + 1] = bb::add(t01123, bb::dbl(x2));
        s[i] = bb::add(t01123, t01);
        s[i + 2] = bb::add(t01233, t23);
    }
    // width 4 included: p3 adds the column sums for every width that is a multiple of four [UPSTREAM-RECALL]
    uint32_t sums[4] = {0, 0, 0, 0};
    for (int i = 0; i < w; i++) sums[i & 3] = bb::add(sums[i & 3], s[i]);
    for (int i = 0; i < w; i++) s[i] = bb::add(s[i], sums[i & 3]);
}

// canonical in / canonical out
void host_poseidon2(int width, const uint32_t* in, uint32_t* out) {
    const lurk_p2_param_row* row = nullptr;
    for (int i = 0; i < LURK_P2_NUM_WIDTHS; i++)
        if (LURK_P2_PARAMS[i].width == width) row = &LURK_P2_PARAMS[i];
    if (!row) throw ExecError("unsupported Poseidon2 width");
    uint32_t s[48];
    for (int i = 0; i < width; i++) s[i] = bb::to_monty(in[i]);
    auto sbox = [](uint32_t x) { return bb::pow7_from_cube(x, bb::cube(x)); };
    host_external_layer(width, s);
    for (int half = 0; half < 2; half++) {
        for (int r = half * 4; r < half * 4 + 4; r++) {
            for (int i = 0; i < width; i++) s[i] = sbox(bb::add(s[i], bb::to_monty(row->ext_rc[r * width + i])));
            host_external_layer(width, s);
        }
        if (half == 0) {
            for (int r = 0; r < row->rounds_p; r++) {
                s[0] = sbox(bb::add(s[0], bb::to_monty(row->int_rc[r])));
                uint32_t sum = 0;
                for (int i = 0; i < width; i++) sum = bb::add(sum, s[i]);
                for (int i = 0; i < width; i++) s[i] = bb::add(bb::mul(s[i], bb::to_monty(row->diag[i])), sum);
            }
        }
    }
    for (int i = 0; i < width; i++) out[i] = bb::from_monty(s[i]);
}

uint64_t into_u64(const uint32_t* v) {
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) {
        if (v[i] > 255) throw ExecError("u64 limb out of byte range");
        r |= (uint64_t)v[i] << (8 * i);
    }
    return r;
}
List u64_bytes(uint64_t v) {
    List r(8);
    for (int i = 0; i < 8; i++) r[i] = (uint32_t)((v >> (8 * i)) & 0xff);
    return r;
}

}  // namespace

List Chip::execute(const List& input, uint32_t nonce, BytesRecord& bytes, std::vector<Record>& rq) const {
    if (input.size() != input_size) throw ExecError("chip " + name + ": input size mismatch");
    switch (kind) {
        case CHIP_HASHER3:
        case CHIP_HASHER4:
        case CHIP_HASHER5: {
            uint32_t out[48];
            host_poseidon2((int)input_size, input.data(), out);
            return List(out, out + 8);
        }
        case CHIP_U64_ADD:
        case CHIP_U64_SUB: {
            // Sum/Diff::populate: 8 result bytes, range-checked in pairs (unsigned/add.rs:65-78,120-133)
            uint64_t a = into_u64(&input[0]), b = into_u64(&input[8]);
            uint64_t r = kind == CHIP_U64_ADD ? a + b : a - b;
            uint8_t by[8];
            for (int i = 0; i < 8; i++) by[i] = (uint8_t)(r >> (8 * i));
            bytes.range_check_u8_iter(by, 8, nonce, rq);
            return u64_bytes(r);
        }
        case CHIP_U64_MUL: {
            // Product::populate (unsigned/mul.rs:24-64,125-133): 8 u16 carry checks, then 4 byte-pair checks
            uint64_t a = into_u64(&input[0]), b = into_u64(&input[8]);
            uint32_t products[8] = {0};
            for (int i = 0; i < 8; i++)
                for (int j = 0; i + j < 8; j++) products[i + j] += ((a >> (8 * i)) & 0xff) * ((b >> (8 * j)) & 0xff);
            uint16_t carry = 0;
            uint8_t res[8];
            for (int k = 0; k < 8; k++) {
                uint32_t o = products[k] + carry;
                res[k] = (uint8_t)(o & 0xff);
                carry = (uint16_t)((o >> 8) & 0xffff);
                bytes.range_check_u16(carry, nonce, rq);
            }
            bytes.range_check_u8_iter(res, 8, nonce, rq);
            uint64_t r = 0;
            for (int k = 0; k < 8; k++) r |= (uint64_t)res[k] << (8 * k);
            return u64_bytes(r);
        }
        case CHIP_U64_LESSTHAN: {
            // CompareWitness::populate (unsigned/cmp.rs:22-47): one less_than lookup on the most
            // significant differing byte pair, or on (0, 0) when equal
            uint64_t a = into_u64(&input[0]), b = into_u64(&input[8]);
            for (int i = 7; i >= 0; i--) {
                uint8_t l = (uint8_t)(a >> (8 * i)), r = (uint8_t)(b >> (8 * i));
                if (l != r) {
                    bool lt = bytes.less_than(l, r, nonce, rq);
                    return List{lt ? 1u : 0u};
                }
            }
            bytes.less_than(0, 0, nonce, rq);
            return List{0u};
        }
        case CHIP_U64_ISZERO: {
            uint64_t a = into_u64(&input[0]);
            return List{a == 0 ? 1u : 0u};
        }
        case CHIP_U64_DIVREM: {
            // DivRem::populate (unsigned/div_rem.rs:33-62): byte lookups in the order q bytes (4 pairs), q * b
            // (8 u16 carries, 4 result pairs), r = a - q b (4 pairs), r < b (1 less_than), q b <= a (1 less_than)
            uint64_t a = into_u64(&input[0]), b = into_u64(&input[8]);
            if (b == 0) throw ExecError("expected input to be non-zero");
            const uint64_t qv = a / b, qb = qv * b, r = a - qb;
            auto bytes_of = [](uint64_t v, uint8_t* by) {
                for (int i = 0; i < 8; i++) by[i] = (uint8_t)(v >> (8 * i));
            };
            uint8_t by[8];
            bytes_of(qv, by);
            bytes.range_check_u8_iter(by, 8, nonce, rq);
            {
                uint32_t products[8] = {0};
                for (int i = 0; i < 8; i++)
                    for (int j = 0; i + j < 8; j++) products[i + j] += ((qv >> (8 * i)) & 0xff) * ((b >> (8 * j)) & 0xff);
                uint16_t carry = 0;
                uint8_t res[8];
                for (int k = 0; k < 8; k++) {
                    uint32_t o = products[k] + carry;
                    res[k] = (uint8_t)(o & 0xff);
                    carry = (uint16_t)((o >> 8) & 0xffff);
                    bytes.range_check_u16(carry, nonce, rq);
                }
                bytes.range_check_u8_iter(res, 8, nonce, rq);
            }
            bytes_of(r, by);
            bytes.range_check_u8_iter(by, 8, nonce, rq);
            auto msb_less_than = [&](uint64_t l, uint64_t rr, bool strict) {
                for (int i = 7; i >= 0; i--) {
                    uint8_t x = (uint8_t)(l >> (8 * i)), y = (uint8_t)(rr >> (8 * i));
                    if (x != y) {
                        bytes.less_than(x, y, nonce, rq);
                        return;
                    }
                }
                if (strict) throw ExecError("less-than witness on equal operands");
                bytes.less_than(0, 0, nonce, rq);
            };
            msb_less_than(r, b, true);    // LessThanWitness::populate asserts r < b
            msb_less_than(qb, a, false);  // CompareWitness::populate
            List out = u64_bytes(qv), rem = u64_bytes(r);
            out.insert(out.end(), rem.begin(), rem.end());
            return out;
        }
        case CHIP_BIGNUM_LESSTHAN: {
            // BigNumCompareWitness::populate (gadgets/big_num/cmp.rs:24-49): most significant differing field element,
            // both as 4-byte words (FieldToWord32: less_than(msb, 0x78), then 2 byte pairs each), then CompareWitness<4>
            uint32_t l = 0, r = 0;
            for (int i = 7; i >= 0; i--)
                if (input[i] != input[8 + i]) {
                    l = input[i];
                    r = input[8 + i];
                    break;
                }
            auto field_to_word = [&](uint32_t v) {
                uint8_t by[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
                bytes.less_than(by[3], 0x78, nonce, rq);
                bytes.range_check_u8_iter(by, 4, nonce, rq);
            };
            field_to_word(l);
            field_to_word(r);
            for (int i = 3; i >= 0; i--) {
                uint8_t x = (uint8_t)(l >> (8 * i)), y = (uint8_t)(r >> (8 * i));
                if (x != y) {
                    bool lt = bytes.less_than(x, y, nonce, rq);
                    return List{lt ? 1u : 0u};
                }
            }
            bytes.less_than(0, 0, nonce, rq);
            return List{0u};
        }
        default:
            throw ExecError("extern chip " + name + " is not supported by this build");
    }
}

std::vector<Chip> lurk_chip_map() {
    // core/chipset.rs:28-63.  Sizes: core/poseidon.rs:44-59, core/u64.rs:46-83, SURVEY appendix B.
    auto hasher = [](const char* n, ChipKind k, uint32_t w, uint32_t rp) {
        Chip c;
        c.name = n;
        c.kind = k;
        c.input_size = w;
        c.output_size = 8;
        c.witness_size = 8 + 16 * w + w + (rp - 1) + rp;
        c.require_size = 0;
        c.witness_return_size = w;  // populate_witness returns the whole state (core/poseidon.rs:71)
        return c;
    };
    auto u64 = [](const char* n, ChipKind k, uint32_t in, uint32_t out, uint32_t wit, uint32_t req) {
        Chip c;
        c.name = n;
        c.kind = k;
        c.input_size = in;
        c.output_size = out;
        c.witness_size = wit;
        c.require_size = req;
        c.witness_return_size = out;
        return c;
    };
    return {
        hasher("hasher3", CHIP_HASHER3, 24, 21),
        hasher("hasher4", CHIP_HASHER4, 32, 30),
        hasher("hasher5", CHIP_HASHER5, 40, 38),
        u64("u64_add", CHIP_U64_ADD, 16, 8, 8, 4),
        u64("u64_sub", CHIP_U64_SUB, 16, 8, 8, 4),
        u64("u64_mul", CHIP_U64_MUL, 16, 8, 16, 12),
        u64("u64_divrem", CHIP_U64_DIVREM, 16, 16, 62, 22),  // DivRem<_, 8> (gadgets/unsigned/div_rem.rs:16-31,112-124)
        u64("u64_lessthan", CHIP_U64_LESSTHAN, 16, 1, 12, 1),
        u64("u64_iszero", CHIP_U64_ISZERO, 8, 1, 9, 0),
        u64("big_num_lessthan", CHIP_BIGNUM_LESSTHAN, 16, 1, 28, 7),  // BigNumCompareWitness (gadgets/big_num/cmp.rs:13-22)
    };
}

// ------------------------------------------------------------------ query maps
void QueryMap::grow() {
    const size_t cap = slots.empty() ? 1024 : slots.size() * 2;
    slots.assign(cap, 0);
    const size_t mask = cap - 1;
    for (size_t i = 0; i < vals.size(); i++) {
        size_t s = hash(key(i), key_len) & mask;
        while (slots[s]) s = (s + 1) & mask;
        slots[s] = (uint32_t)i + 1;
    }
}

uint32_t QueryMap::push_hashed(const uint32_t* k, uint32_t n, const QueryResult& v, uint64_t h) {
    if (vals.empty()) key_len = n;
    if (n != key_len) throw ExecError("query table key length changed");
    if ((vals.size() + 1) * 2 > slots.size()) grow();
    const uint32_t i = (uint32_t)vals.size();
    key_pool.insert(key_pool.end(), k, k + n);
    vals.push_back(v);
    const size_t mask = slots.size() - 1;
    size_t s = h & mask;
    while (slots[s]) s = (s + 1) & mask;
    slots[s] = i + 1;
    return i;
}

QueryRecord::QueryRecord(const Toplevel& t) {
    func_queries.resize(t.funcs.size());
    for (const auto& f : t.funcs) {
        if (f.invertible) inv_func_queries.emplace_back(new std::unordered_map<List, List, VecHash>());
        else inv_func_queries.emplace_back(nullptr);
    }
    mem_queries.resize(NUM_MEM_TABLES);
}

void QueryRecord::clean() {
    for (auto& q : func_queries) q.clear();
    for (auto& q : mem_queries) q.clear();
    bytes.clear();
    emitted.clear();
}

void QueryRecord::inject_inv_query(uint32_t func_idx, const List& inp, const List& out) {
    if (func_idx >= inv_func_queries.size() || !inv_func_queries[func_idx]) throw ExecError("Inverse query map not found");
    (*inv_func_queries[func_idx])[out] = inp;
}

size_t num_shards(const QueryRecord& r, uint32_t max_shard_size) {
    size_t mx = 0;
    for (const auto& q : r.func_queries) mx = std::max(mx, q.size());
    return mx / max_shard_size + (mx % max_shard_size ? 1 : 0);
}

// ------------------------------------------------------------------ the interpreter (execute.rs:436-784)
namespace {

// One activation of a Func.  The reference keeps an exec-entry stack plus a caller stack (execute.rs:436-470); a block's
// control node is always its last entry, so a (block, next op) cursor per activation is the same traversal.  Activations
// live in a grow-only pool and are reused (their vectors keep their capacity): multi-million-query executions recurse
// millions of frames deep, and allocating five vectors per call was most of the interpreter's time.
struct Frame {
    const Block* blk = nullptr;
    size_t ip = 0;
    bool preimg = false;
    bool partial = false;
    uint32_t func_index = 0;
    uint32_t nonce = 0;
    // where this activation's slices start in the five arenas (Arenas below)
    size_t map0 = 0, req0 = 0, dep0 = 0, dreq0 = 0, hint0 = 0;
};

// The variable map, requires, callee depths, depth requires and hints of every live activation, each kind in one
// contiguous stack: only the innermost activation grows, and a return truncates the stacks to its start offsets, so live
// memory is exactly the live data (no per-frame capacity slack, no allocation per call).
struct Arenas {
    BigVec<uint32_t> map, depths, hints;
    BigVec<Record> requires_, depth_requires;
};

void depth_less_than_populate(uint32_t lhs, uint32_t rhs, BytesRecord& bytes, uint32_t nonce, std::vector<Record>& rq) {
    // LessThanWitness::populate (unsigned/less_than.rs:20-41): one less_than lookup on the most
    // significant differing byte
    if (!(lhs < rhs)) throw ExecError("depth ordering violated");
    for (int i = DEPTH_W - 1; i >= 0; i--) {
        uint8_t l = (uint8_t)(lhs >> (8 * i)), r = (uint8_t)(rhs >> (8 * i));
        if (l != r) {
            bytes.less_than(l, r, nonce, rq);
            return;
        }
    }
}

}  // namespace

static std::pair<List, uint32_t> func_execute(const Toplevel& t, const Func& self, const List& args, QueryRecord& q) {
    {
        QueryResult top;
        top.provide.count = 1;
        q.func_queries[self.index].insert_full(args, top);
    }
    std::deque<Frame> frames;  // stable addresses; frames[0 .. depth] are live
    Arenas A;
    size_t depth = 0;
    frames.emplace_back();
    Frame* f = &frames[0];
    f->blk = &self.body;
    f->func_index = self.index;
    f->nonce = (uint32_t)q.func_queries[self.index].find(args);
    f->partial = self.partial;
    A.map.insert(A.map.end(), args.begin(), args.end());
    List key, inp, out;  // scratch, reused
    std::vector<Record> chip_requires;

    auto enter = [&](bool preimg, uint32_t callee_index, const List& input, uint64_t input_hash) {
        // (the lookup that brought us here missed: the key is absent, its hash known)
        const uint32_t callee_nonce = q.func_queries[callee_index].push_hashed(input.data(), (uint32_t)input.size(), QueryResult(), input_hash);
        depth++;
        if (depth == frames.size()) frames.emplace_back();
        Frame* n = &frames[depth];
        const Func& cf = t.funcs[callee_index];
        n->blk = &cf.body;
        n->ip = 0;
        n->preimg = preimg;
        n->partial = cf.partial;
        n->func_index = callee_index;
        n->nonce = callee_nonce;
        n->map0 = A.map.size();
        n->req0 = A.requires_.size();
        n->dep0 = A.depths.size();
        n->dreq0 = A.depth_requires.size();
        n->hint0 = A.hints.size();
        A.map.insert(A.map.end(), input.begin(), input.end());
        f = n;
    };

    for (;;) {
        if (f->ip < f->blk->ops.size()) {
            const Op& op = f->blk->ops[f->ip++];
            // the innermost activation's variable map is the top slice of A.map; `map` is re-read per op because a push may
            // move the arena (values are fetched before they are pushed)
            const uint32_t* map = A.map.data() + f->map0;
            const uint32_t nonce = f->nonce;
            switch (op.kind) {
                case OpKind::AssertEq:
                    for (size_t i = 0; i < op.a.size(); i++)
                        if (map[op.a[i]] != map[op.b[i]]) throw ExecError("assert_eq! failed in " + t.funcs[f->func_index].name);
                    break;
                case OpKind::AssertNe: {
                    bool unequal = false;
                    for (size_t i = 0; i < op.a.size(); i++)
                        if (map[op.a[i]] != map[op.b[i]]) {
                            unequal = true;
                            break;
                        }
                    if (!unequal) throw ExecError("assert_ne! failed in " + t.funcs[f->func_index].name);
                    break;
                }
                case OpKind::Contains: {
                    bool found = false;
                    for (uint32_t a : op.a) found = found || map[a] == map[op.y];
                    if (!found) throw ExecError("contains! failed in " + t.funcs[f->func_index].name);
                    break;
                }
                case OpKind::Call:
                case OpKind::PreImg: {
                    const bool pre = op.kind == OpKind::PreImg;
                    const uint32_t callee = op.x;
                    key.clear();
                    for (uint32_t v : op.a) key.push_back(map[v]);
                    if (pre) {
                        auto& inv = q.inv_func_queries[callee];
                        if (!inv) throw ExecError("Missing inverse map");
                        auto it = inv->find(key);
                        if (it == inv->end()) throw ExecError("Preimg not found");
                        inp = it->second;
                    } else {
                        inp = key;
                    }
                    QueryMap& cqm = q.func_queries[callee];
                    const uint64_t inp_hash = QueryMap::hash(inp.data(), (uint32_t)inp.size());
                    if (!cqm.vals.empty() && inp.size() != cqm.key_len) throw ExecError("query table key length changed");
                    int idx = cqm.find_hashed(inp.data(), (uint32_t)inp.size(), inp_hash);
                    if (idx >= 0) {
                        QueryResult& res = cqm.vals[idx];
                        if (!res.has_output) throw ExecError("Loop detected");
                        const uint32_t n_out = t.funcs[callee].output_size;
                        const uint32_t* res_out = cqm.output(res);
                        if (pre && (key.size() != n_out || memcmp(res_out, key.data(), (size_t)n_out * 4) != 0))
                            throw ExecError("memoized output differs from preimage key");
                        const uint32_t* ext = pre ? inp.data() : res_out;
                        const size_t n_ext = pre ? inp.size() : n_out;
                        A.map.insert(A.map.end(), ext, ext + n_ext);
                        A.hints.insert(A.hints.end(), ext, ext + n_ext);
                        A.requires_.push_back(res.provide.new_lookup(nonce));
                        const bool callee_partial = t.funcs[callee].partial;
                        if (callee_partial) A.hints.push_back(res.depth);
                        if (f->partial && callee_partial) A.depths.push_back(res.depth);
                    } else {
                        enter(pre, callee, inp, inp_hash);
                    }
                    break;
                }
                case OpKind::Const:
                    A.map.push_back(op.c);
                    break;
                case OpKind::Add:
                    A.map.push_back(fadd(map[op.x], map[op.y]));
                    break;
                case OpKind::Sub:
                    A.map.push_back(fsub(map[op.x], map[op.y]));
                    break;
                case OpKind::Mul:
                    A.map.push_back(fmul(map[op.x], map[op.y]));
                    break;
                case OpKind::Inv:
                    A.map.push_back(finv(map[op.x]));
                    break;
                case OpKind::Not:
                    A.map.push_back(map[op.x] == 0 ? 1u : 0u);
                    break;
                case OpKind::Store: {
                    key.clear();
                    for (uint32_t v : op.a) key.push_back(map[v]);
                    QueryMap& mm = q.mem_queries[mem_index_from_len((uint32_t)key.size())];
                    const uint64_t key_hash = QueryMap::hash(key.data(), (uint32_t)key.size());
                    int i = mm.find_hashed(key.data(), (uint32_t)key.size(), key_hash);
                    if (i < 0) i = (int)mm.push_hashed(key.data(), (uint32_t)key.size(), QueryResult(), key_hash);
                    uint32_t ptr = (uint32_t)(i + 1);
                    A.map.push_back(ptr);
                    A.hints.push_back(ptr);
                    A.requires_.push_back(mm.vals[i].provide.new_lookup(nonce));
                    break;
                }
                case OpKind::Load: {
                    uint32_t ptr = map[op.y];
                    QueryMap& mm = q.mem_queries[mem_index_from_len(op.x)];
                    if (ptr == 0 || ptr > mm.size()) throw ExecError("Unbound pointer");
                    const uint32_t* vals = mm.key(ptr - 1);
                    A.map.insert(A.map.end(), vals, vals + op.x);
                    A.hints.insert(A.hints.end(), vals, vals + op.x);
                    A.requires_.push_back(mm.vals[ptr - 1].provide.new_lookup(nonce));
                    break;
                }
                case OpKind::ExternCall: {
                    key.clear();
                    for (uint32_t v : op.a) key.push_back(map[v]);
                    chip_requires.clear();
                    List res = t.chips[op.x].execute(key, nonce, q.bytes, chip_requires);
                    A.requires_.insert(A.requires_.end(), chip_requires.begin(), chip_requires.end());
                    A.map.insert(A.map.end(), res.begin(), res.end());
                    break;
                }
                case OpKind::Emit: {
                    List v;
                    for (uint32_t a : op.a) v.push_back(map[a]);
                    q.emitted.push_back(v);
                    break;
                }
                case OpKind::RangeU8: {
                    std::vector<uint8_t> by;
                    for (uint32_t a : op.a) {
                        if (map[a] > 255) throw ExecError("Variable not in u8 range");
                        by.push_back((uint8_t)map[a]);
                    }
                    chip_requires.clear();
                    q.bytes.range_check_u8_iter(by.data(), by.size(), nonce, chip_requires);
                    A.requires_.insert(A.requires_.end(), chip_requires.begin(), chip_requires.end());
                    break;
                }
                case OpKind::Breakpoint:
                case OpKind::Debug:
                    break;
            }
            continue;
        }
        const Ctrl& c = f->blk->ctrl;
        if (c.kind == Ctrl::Choose) {
            key.assign(1, A.map[f->map0 + c.var]);
            const Block* b = c.match_case(key);
            if (!b) throw ExecError("No match");
            f->blk = b;
            f->ip = 0;
            continue;
        }
        if (c.kind == Ctrl::ChooseMany) {
            key.clear();
            for (uint32_t v : c.vars) key.push_back(A.map[f->map0 + v]);
            const Block* b = c.match_case(key);
            if (!b) throw ExecError("No match");
            f->blk = b;
            f->ip = 0;
            continue;
        }
        // Return
        out.clear();
        for (uint32_t v : c.ret) out.push_back(A.map[f->map0 + v]);
        const uint32_t func_index = f->func_index, nonce = f->nonce;
        QueryMap& qm = q.func_queries[func_index];
        QueryResult& result = qm.vals[nonce];
        if (result.has_output) throw ExecError("query evaluated twice");
        inp.assign(qm.key(nonce), qm.key(nonce) + qm.key_len);
        if (q.inv_func_queries[func_index]) (*q.inv_func_queries[func_index])[out] = inp;
        if (f->partial) {
            uint32_t d_self = 0;
            for (size_t i = f->dep0; i < A.depths.size(); i++) d_self = std::max(d_self, A.depths[i] + 1);
            uint8_t by[4] = {(uint8_t)d_self, (uint8_t)(d_self >> 8), (uint8_t)(d_self >> 16), (uint8_t)(d_self >> 24)};
            chip_requires.clear();
            q.bytes.range_check_u8_iter(by, 4, nonce, chip_requires);
            for (size_t i = f->dep0; i < A.depths.size(); i++) depth_less_than_populate(A.depths[i], d_self, q.bytes, nonce, chip_requires);
            A.depth_requires.insert(A.depth_requires.end(), chip_requires.begin(), chip_requires.end());
            result.depth = d_self;
        }
        // finalize: the activation's slices move into the table's pools
        result.out_off = (uint32_t)qm.pool.size();
        qm.pool.insert(qm.pool.end(), out.begin(), out.end());
        result.hint_off = (uint32_t)qm.pool.size();
        result.n_hints = (uint32_t)(A.hints.size() - f->hint0);
        qm.pool.insert(qm.pool.end(), A.hints.begin() + f->hint0, A.hints.end());
        result.req_off = (uint32_t)qm.rec_pool.size();
        result.n_requires = (uint32_t)(A.requires_.size() - f->req0);
        result.n_depth_requires = (uint32_t)(A.depth_requires.size() - f->dreq0);
        qm.rec_pool.insert(qm.rec_pool.end(), A.requires_.begin() + f->req0, A.requires_.end());
        qm.rec_pool.insert(qm.rec_pool.end(), A.depth_requires.begin() + f->dreq0, A.depth_requires.end());
        if (qm.pool.size() > 0xfffffff0ull || qm.rec_pool.size() > 0xfffffff0ull) throw ExecError("query table pools exceed 2^32 words");
        result.has_output = true;
        if (depth == 0) {
            uint32_t d_top = 0;
            for (size_t i = f->dep0; i < A.depths.size(); i++) d_top = std::max(d_top, A.depths[i] + 1);
            return {out, d_top};
        }
        // pop: the caller's slices are the arenas' tops again
        A.map.resize(f->map0);
        A.requires_.resize(f->req0);
        A.depths.resize(f->dep0);
        A.depth_requires.resize(f->dreq0);
        A.hints.resize(f->hint0);
        const bool callee_partial = f->partial, callee_preimg = f->preimg;
        const uint32_t callee_depth = result.depth;
        depth--;
        f = &frames[depth];
        const List& ext = callee_preimg ? inp : out;
        A.map.insert(A.map.end(), ext.begin(), ext.end());
        A.hints.insert(A.hints.end(), ext.begin(), ext.end());
        // `result` still refers to the callee's table slot: nothing was inserted between the Return and here
        A.requires_.push_back(result.provide.new_lookup(f->nonce));
        if (callee_partial) A.hints.push_back(callee_depth);
        if (f->partial && callee_partial) A.depths.push_back(callee_depth);
    }
}

List execute(const Toplevel& t, const Func& func, const List& args, QueryRecord& record) {
    if (args.size() != func.input_size) throw ExecError("Argument mismatch");
    auto [out, depth] = func_execute(t, func, args, record);
    record.public_values = args;
    record.public_values.insert(record.public_values.end(), out.begin(), out.end());
    if (func.partial)
        for (int i = 0; i < 4; i++) record.public_values.push_back((depth >> (8 * i)) & 0xff);
    record.has_public_values = true;
    return out;
}

}  // namespace lair
