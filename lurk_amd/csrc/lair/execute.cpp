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
#include <mutex>
#include <new>
#include <unordered_map>

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
    rq.push_back(at(key, BYTES_RANGE_U8).new_lookup(nonce));
}
void BytesRecord::range_check_u8_iter(const uint8_t* b, size_t n, uint32_t nonce, std::vector<Record>& rq) {
    for (size_t i = 0; i < n; i += 2) range_check_u8_pair(b[i], i + 1 < n ? b[i + 1] : 0, nonce, rq);
}
bool BytesRecord::less_than(uint8_t i1, uint8_t i2, uint32_t nonce, std::vector<Record>& rq) {
    uint16_t key = (uint16_t)(i1 | (i2 << 8));
    rq.push_back(at(key, BYTES_LESS_THAN).new_lookup(nonce));
    return i1 < i2;
}
void BytesRecord::range_check_u16(uint16_t v, uint32_t nonce, std::vector<Record>& rq) {
    rq.push_back(at(v, BYTES_RANGE_U16).new_lookup(nonce));
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
    // bits = 4 x slots -> blocks of 512 bits = cap / 128; the entries waiting for their seat are re-seated with the rest
    const size_t blocks = cap / 128;
    uint32_t log_blocks = 0;
    while (((size_t)1 << log_blocks) < blocks) log_blocks++;
    bloom.assign(blocks * 8, 0);
    bloom_shift = 32 - log_blocks;
    if (log_blocks == 0) throw ExecError("query table index too small for its filter");
    pending_head = n_pending = 0;
    for (size_t i = 0; i < vals.size(); i++) {
        size_t s = hashes[i] & mask;  // (a table holds fewer than 2^31 entries: the index needs no more than 32 hash bits)
        while (slots[s]) s = (s + 1) & mask;
        slots[s] = (uint32_t)i + 1;
        bloom_add(hashes[i]);
    }
}

uint32_t QueryMap::push_hashed(const uint32_t* k, uint32_t n, const QueryResult& v, uint64_t h) {
    if (vals.empty()) key_len = n;
    if (n != key_len) throw ExecError("query table key length changed");
    if ((vals.size() + 1) * 2 > slots.size()) grow();
    if (vals.size() >= 0x7fffffffull) throw ExecError("query table exceeds 2^31 entries");
    const uint32_t i = (uint32_t)vals.size();
    key_pool.append(k, n);
    vals.push_back(v);
    hashes.push_back((uint32_t)h);
    bloom_add((uint32_t)h);
    const size_t mask = slots.size() - 1;
    if (n_pending == MAX_PENDING) {
        // the oldest waiting entry takes its seat: its line was requested MAX_PENDING insertions ago
        const Pending p = pending[pending_head];
        pending_head = (pending_head + 1) % MAX_PENDING;
        n_pending--;
        size_t s = p.h32 & mask;
        while (slots[s]) s = (s + 1) & mask;
        slots[s] = p.index + 1;
    }
    pending[(pending_head + n_pending) % MAX_PENDING] = Pending{i, (uint32_t)h};
    n_pending++;
    __builtin_prefetch(&slots[(uint32_t)h & mask], 1);
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

// Pre-decoded functions.  A function is one flat array of fixed-size operations: a block's operations, its control
// operation, then the blocks it branches to; operand lists live in one pool.  (Round 3: walking `Block::ops` -- 80-byte
// `Op`s with two std::vectors each -- and `Ctrl` through shared_ptrs was a fifth of the interpreter's time; runs of constants,
// which the Lurk toplevel is full of (tags), become one copy.)
enum XKind : uint32_t {
    X_ASSERT_EQ, X_ASSERT_NE, X_CONTAINS, X_CONST, X_CONST_RUN, X_ADD, X_SUB, X_MUL, X_INV, X_NOT, X_CALL, X_PREIMG, X_STORE,
    X_LOAD, X_EXTERN, X_EMIT, X_RANGE_U8, X_CHOOSE, X_CHOOSE_MANY, X_RETURN
};
constexpr uint32_t X_NONE = 0xffffffffu;
struct XOp {
    uint32_t kind;
    uint32_t n;        // length of the operand list `a` (CONST_RUN: constants; CHOOSE / CHOOSE_MANY: cases)
    uint32_t x, y, c;  // as in Op; STORE / LOAD: x = memory table index, LOAD: n = length; CHOOSE: x = variable, y = default
                       // target; CHOOSE_MANY: x = key length, y = default target
    uint32_t a, b;     // offsets into XProgram::args.  CHOOSE: a = keys[n] (sorted), b = targets[n]; CHOOSE_MANY: a = vars[x],
                       // b = keys[n][x] (sorted) then targets[n]
};
struct XProgram {
    std::vector<XOp> ops;
    std::vector<uint32_t> args;
    std::vector<uint32_t> entry;  // per function: its first operation
};

struct XBuilder {
    XProgram& p;
    std::unordered_map<const Block*, uint32_t> placed;
    uint32_t list(const std::vector<uint32_t>& v) {
        const uint32_t off = (uint32_t)p.args.size();
        p.args.insert(p.args.end(), v.begin(), v.end());
        return off;
    }
    // a length without a table fails when the operation runs, as in the reference (execute.rs:243-256), not when it is decoded
    static uint32_t table_of(uint32_t len) {
        for (int i = 0; i < NUM_MEM_TABLES; i++)
            if (MEM_TABLE_SIZES[i] == len) return (uint32_t)i;
        return X_NONE;
    }
    uint32_t block(const Block& blk) {
        auto it = placed.find(&blk);
        if (it != placed.end()) return it->second;
        const uint32_t start = (uint32_t)p.ops.size();
        placed[&blk] = start;
        for (size_t i = 0; i < blk.ops.size(); i++) {
            const Op& op = blk.ops[i];
            XOp x{};
            x.x = op.x, x.y = op.y, x.c = op.c;
            switch (op.kind) {
                case OpKind::AssertEq:
                case OpKind::AssertNe:
                    x.kind = op.kind == OpKind::AssertEq ? X_ASSERT_EQ : X_ASSERT_NE;
                    x.n = (uint32_t)op.a.size(), x.a = list(op.a), x.b = list(op.b);
                    break;
                case OpKind::Contains:
                    x.kind = X_CONTAINS, x.n = (uint32_t)op.a.size(), x.a = list(op.a);
                    break;
                case OpKind::Const: {
                    size_t j = i;
                    while (j < blk.ops.size() && blk.ops[j].kind == OpKind::Const) j++;
                    if (j - i >= 2) {
                        List cs;
                        for (size_t k = i; k < j; k++) cs.push_back(blk.ops[k].c);
                        x.kind = X_CONST_RUN, x.n = (uint32_t)cs.size(), x.a = list(cs);
                        i = j - 1;
                    } else {
                        x.kind = X_CONST;
                    }
                    break;
                }
                case OpKind::Add: x.kind = X_ADD; break;
                case OpKind::Sub: x.kind = X_SUB; break;
                case OpKind::Mul: x.kind = X_MUL; break;
                case OpKind::Inv: x.kind = X_INV; break;
                case OpKind::Not: x.kind = X_NOT; break;
                case OpKind::Call:
                case OpKind::PreImg:
                    x.kind = op.kind == OpKind::Call ? X_CALL : X_PREIMG;
                    x.n = (uint32_t)op.a.size(), x.a = list(op.a);
                    break;
                case OpKind::Store:
                    x.kind = X_STORE, x.n = (uint32_t)op.a.size(), x.a = list(op.a), x.x = table_of(x.n);
                    break;
                case OpKind::Load:
                    x.kind = X_LOAD, x.n = op.x, x.x = table_of(op.x);
                    break;
                case OpKind::ExternCall:
                    x.kind = X_EXTERN, x.n = (uint32_t)op.a.size(), x.a = list(op.a);
                    break;
                case OpKind::Emit:
                    x.kind = X_EMIT, x.n = (uint32_t)op.a.size(), x.a = list(op.a);
                    break;
                case OpKind::RangeU8:
                    x.kind = X_RANGE_U8, x.n = (uint32_t)op.a.size(), x.a = list(op.a);
                    break;
                case OpKind::Breakpoint:
                case OpKind::Debug:
                    continue;
            }
            p.ops.push_back(x);
        }
        const Ctrl& c = blk.ctrl;
        const uint32_t at = (uint32_t)p.ops.size();
        XOp x{};
        if (c.kind == Ctrl::Return) {
            x.kind = X_RETURN, x.n = (uint32_t)c.ret.size(), x.a = list(c.ret), x.x = c.ident;
            p.ops.push_back(x);
            return start;
        }
        const bool many = c.kind == Ctrl::ChooseMany;
        const uint32_t m = many ? (uint32_t)c.vars.size() : 1, n = (uint32_t)c.branches.size();
        x.kind = many ? X_CHOOSE_MANY : X_CHOOSE;
        x.n = n, x.x = many ? m : c.var, x.y = X_NONE;
        if (many) x.a = list(c.vars);
        List keys;
        for (const auto& br : c.branches) {  // sorted by key (compile.cpp, bytecode_io.cpp)
            if (br.first.size() != m) throw ExecError("match key length differs from the matched variables");
            keys.insert(keys.end(), br.first.begin(), br.first.end());
        }
        const uint32_t keys_off = list(keys), targets_off = list(List(n, X_NONE));
        if (many) x.b = keys_off;
        else x.a = keys_off, x.b = targets_off;
        p.ops.push_back(x);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t target = block(*c.branches[i].second);
            p.args[targets_off + i] = target;
        }
        if (c.def) {
            const uint32_t target = block(*c.def);
            p.ops[at].y = target;
        }
        return start;
    }
};

std::shared_ptr<const XProgram> program_of(const Toplevel& t) {
    std::lock_guard<std::mutex> lock(t.exec_cache.mu);
    if (!t.exec_cache.program) {
        auto p = std::make_shared<XProgram>();
        XBuilder b{*p, {}};
        for (const Func& f : t.funcs) p->entry.push_back(b.block(f.body));
        t.exec_cache.program = p;
    }
    return std::static_pointer_cast<const XProgram>(t.exec_cache.program);
}

// One suspended activation of a Func.  The reference keeps an exec-entry stack plus a caller stack (execute.rs:436-470); an
// operation cursor per activation is the same traversal.  Activations live in one grow-only array: multi-million-query
// executions recurse millions of frames deep.
struct Frame {
    const XOp* ip = nullptr;
    uint32_t func_index = 0;
    uint32_t nonce = 0;
    bool preimg = false;
    bool partial = false;
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
    const std::shared_ptr<const XProgram> prog = program_of(t);
    const XOp* const code = prog->ops.data();
    const uint32_t* const pargs = prog->args.data();
    {
        QueryResult top;
        top.provide.count = 1;
        q.func_queries[self.index].insert_full(args, top);
    }
    // frames[0 .. depth] are live, contiguous: a deep recursion returns through them in descending order, which the hardware
    // prefetchers follow in one array (a deque's 512-byte blocks were a cache miss per return) and the return below can name
    // the frames it will resume next.  The innermost activation `cur` is a local copy; frames[depth] is written when it
    // suspends.
    BigVec<Frame> frames;
    Arenas A;
    size_t depth = 0;
    frames.push_back(Frame());
    Frame cur;
    cur.ip = code + prog->entry[self.index];
    cur.func_index = self.index;
    cur.nonce = (uint32_t)q.func_queries[self.index].find(args);
    cur.partial = self.partial;
    A.map.append(args.data(), args.size());
    List key, inp, out;  // scratch, reused
    std::vector<Record> chip_requires;
    std::vector<uint8_t> by_scratch;

    for (;;) {
        const XOp& op = *cur.ip++;
        // the innermost activation's variable map is the top slice of A.map; `map` is re-read per operation because a push
        // may move the arena (values are fetched before they are pushed)
        const uint32_t* map = A.map.data() + cur.map0;
        const uint32_t nonce = cur.nonce;
        switch (op.kind) {
            case X_ASSERT_EQ: {
                const uint32_t *a = pargs + op.a, *b = pargs + op.b;
                for (uint32_t i = 0; i < op.n; i++)
                    if (map[a[i]] != map[b[i]]) throw ExecError("assert_eq! failed in " + t.funcs[cur.func_index].name);
                break;
            }
            case X_ASSERT_NE: {
                const uint32_t *a = pargs + op.a, *b = pargs + op.b;
                bool unequal = false;
                for (uint32_t i = 0; i < op.n; i++)
                    if (map[a[i]] != map[b[i]]) {
                        unequal = true;
                        break;
                    }
                if (!unequal) throw ExecError("assert_ne! failed in " + t.funcs[cur.func_index].name);
                break;
            }
            case X_CONTAINS: {
                const uint32_t* a = pargs + op.a;
                bool found = false;
                for (uint32_t i = 0; i < op.n; i++) found = found || map[a[i]] == map[op.y];
                if (!found) throw ExecError("contains! failed in " + t.funcs[cur.func_index].name);
                break;
            }
            case X_CALL:
            case X_PREIMG: {
                const bool pre = op.kind == X_PREIMG;
                const uint32_t callee = op.x;
                const uint32_t* a = pargs + op.a;
                key.resize(op.n);
                for (uint32_t i = 0; i < op.n; i++) key[i] = map[a[i]];
                const List* input = &key;
                if (pre) {
                    auto& inv = q.inv_func_queries[callee];
                    if (!inv) throw ExecError("Missing inverse map");
                    auto it = inv->find(key);
                    if (it == inv->end()) throw ExecError("Preimg not found");
                    inp = it->second;
                    input = &inp;
                }
                QueryMap& cqm = q.func_queries[callee];
                const uint32_t n_in = (uint32_t)input->size();
                const uint64_t inp_hash = QueryMap::hash(input->data(), n_in);
                if (!cqm.vals.empty() && n_in != cqm.key_len) throw ExecError("query table key length changed");
                const int idx = cqm.find_hashed(input->data(), n_in, inp_hash);
                const Func& cf = t.funcs[callee];
                if (idx >= 0) {
                    QueryResult& res = cqm.vals[idx];
                    if (!res.has_output) throw ExecError("Loop detected");
                    const uint32_t n_out = cf.output_size;
                    const uint32_t* res_out = cqm.output(res);
                    if (pre && (key.size() != n_out || memcmp(res_out, key.data(), (size_t)n_out * 4) != 0))
                        throw ExecError("memoized output differs from preimage key");
                    const uint32_t* ext = pre ? inp.data() : res_out;
                    const size_t n_ext = pre ? inp.size() : n_out;
                    A.map.append(ext, n_ext);
                    A.hints.append(ext, n_ext);
                    A.requires_.push_back(res.provide.new_lookup(nonce));
                    if (cf.partial) A.hints.push_back(res.depth);
                    if (cur.partial && cf.partial) A.depths.push_back(res.depth);
                } else {
                    // the lookup missed: the key is absent, its hash known.  Suspend this activation, start the callee's.
                    const uint32_t callee_nonce = cqm.push_hashed(input->data(), n_in, QueryResult(), inp_hash);
                    frames[depth] = cur;
                    depth++;
                    if (depth == frames.size()) frames.push_back(Frame());
                    cur.ip = code + prog->entry[callee];
                    cur.preimg = pre;
                    cur.partial = cf.partial;
                    cur.func_index = callee;
                    cur.nonce = callee_nonce;
                    cur.map0 = A.map.size();
                    cur.req0 = A.requires_.size();
                    cur.dep0 = A.depths.size();
                    cur.dreq0 = A.depth_requires.size();
                    cur.hint0 = A.hints.size();
                    A.map.append(input->data(), n_in);
                }
                break;
            }
            case X_CONST:
                A.map.push_back(op.c);
                break;
            case X_CONST_RUN:
                A.map.append(pargs + op.a, op.n);
                break;
            case X_ADD:
                A.map.push_back(fadd_c(map[op.x], map[op.y]));
                break;
            case X_SUB:
                A.map.push_back(fsub_c(map[op.x], map[op.y]));
                break;
            case X_MUL:
                A.map.push_back(fmul(map[op.x], map[op.y]));
                break;
            case X_INV:
                A.map.push_back(finv(map[op.x]));
                break;
            case X_NOT:
                A.map.push_back(map[op.x] == 0 ? 1u : 0u);
                break;
            case X_STORE: {
                if (op.x == X_NONE) (void)mem_index_from_len(op.n);  // throws
                const uint32_t* a = pargs + op.a;
                uint32_t kbuf[8];  // memory tables are at most 8 wide (execute.rs:243-244)
                for (uint32_t i = 0; i < op.n; i++) kbuf[i] = map[a[i]];
                QueryMap& mm = q.mem_queries[op.x];
                const uint64_t key_hash = QueryMap::hash(kbuf, op.n);
                int i = mm.find_hashed(kbuf, op.n, key_hash);
                if (i < 0) i = (int)mm.push_hashed(kbuf, op.n, QueryResult(), key_hash);
                const uint32_t ptr = (uint32_t)(i + 1);
                A.map.push_back(ptr);
                A.hints.push_back(ptr);
                A.requires_.push_back(mm.vals[i].provide.new_lookup(nonce));
                break;
            }
            case X_LOAD: {
                if (op.x == X_NONE) (void)mem_index_from_len(op.n);  // throws
                const uint32_t ptr = map[op.y];
                QueryMap& mm = q.mem_queries[op.x];
                if (ptr == 0 || ptr > mm.size()) throw ExecError("Unbound pointer");
                const uint32_t* vals = mm.key(ptr - 1);
                A.map.append(vals, op.n);
                A.hints.append(vals, op.n);
                A.requires_.push_back(mm.vals[ptr - 1].provide.new_lookup(nonce));
                break;
            }
            case X_EXTERN: {
                const uint32_t* a = pargs + op.a;
                key.resize(op.n);
                for (uint32_t i = 0; i < op.n; i++) key[i] = map[a[i]];
                chip_requires.clear();
                List res = t.chips[op.x].execute(key, nonce, q.bytes, chip_requires);
                A.requires_.append(chip_requires.data(), chip_requires.size());
                A.map.append(res.data(), res.size());
                break;
            }
            case X_EMIT: {
                const uint32_t* a = pargs + op.a;
                List v;
                for (uint32_t i = 0; i < op.n; i++) v.push_back(map[a[i]]);
                q.emitted.push_back(v);
                break;
            }
            case X_RANGE_U8: {
                const uint32_t* a = pargs + op.a;
                by_scratch.clear();
                for (uint32_t i = 0; i < op.n; i++) {
                    if (map[a[i]] > 255) throw ExecError("Variable not in u8 range");
                    by_scratch.push_back((uint8_t)map[a[i]]);
                }
                chip_requires.clear();
                q.bytes.range_check_u8_iter(by_scratch.data(), by_scratch.size(), nonce, chip_requires);
                A.requires_.append(chip_requires.data(), chip_requires.size());
                break;
            }
            case X_CHOOSE: {
                const uint32_t v = map[op.x];
                const uint32_t *keys = pargs + op.a, *targets = pargs + op.b;
                const uint32_t* it = std::lower_bound(keys, keys + op.n, v);
                const uint32_t target = it != keys + op.n && *it == v ? targets[it - keys] : op.y;
                if (target == X_NONE) throw ExecError("No match");
                cur.ip = code + target;
                break;
            }
            case X_CHOOSE_MANY: {
                const uint32_t m = op.x;
                const uint32_t *vars = pargs + op.a, *keys = pargs + op.b, *targets = keys + (size_t)op.n * m;
                key.resize(m);
                for (uint32_t i = 0; i < m; i++) key[i] = map[vars[i]];
                uint32_t lo = 0, hi = op.n;  // first case >= key (lexicographic)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) / 2;
                    if (std::lexicographical_compare(keys + (size_t)mid * m, keys + (size_t)(mid + 1) * m, key.begin(), key.end())) lo = mid + 1;
                    else hi = mid;
                }
                const uint32_t target = lo < op.n && std::equal(key.begin(), key.end(), keys + (size_t)lo * m) ? targets[lo] : op.y;
                if (target == X_NONE) throw ExecError("No match");
                cur.ip = code + target;
                break;
            }
            case X_RETURN: {
                const uint32_t* rv = pargs + op.a;
                out.resize(op.n);
                for (uint32_t i = 0; i < op.n; i++) out[i] = map[rv[i]];
                const uint32_t func_index = cur.func_index;
                QueryMap& qm = q.func_queries[func_index];
                QueryResult& result = qm.vals[nonce];
                if (result.has_output) throw ExecError("query evaluated twice");
                const bool need_input = cur.preimg || q.inv_func_queries[func_index];
                if (need_input) inp.assign(qm.key(nonce), qm.key(nonce) + qm.key_len);
                if (q.inv_func_queries[func_index]) (*q.inv_func_queries[func_index])[out] = inp;
                if (cur.partial) {
                    uint32_t d_self = 0;
                    for (size_t i = cur.dep0; i < A.depths.size(); i++) d_self = std::max(d_self, A.depths[i] + 1);
                    uint8_t by[4] = {(uint8_t)d_self, (uint8_t)(d_self >> 8), (uint8_t)(d_self >> 16), (uint8_t)(d_self >> 24)};
                    chip_requires.clear();
                    q.bytes.range_check_u8_iter(by, 4, nonce, chip_requires);
                    for (size_t i = cur.dep0; i < A.depths.size(); i++) depth_less_than_populate(A.depths[i], d_self, q.bytes, nonce, chip_requires);
                    A.depth_requires.append(chip_requires.data(), chip_requires.size());
                    result.depth = d_self;
                }
                // finalize: the activation's slices move into the table's pools
                result.out_off = (uint32_t)qm.pool.size();
                qm.pool.append(out.data(), out.size());
                result.hint_off = (uint32_t)qm.pool.size();
                result.n_hints = (uint32_t)(A.hints.size() - cur.hint0);
                qm.pool.append(A.hints.data() + cur.hint0, A.hints.size() - cur.hint0);
                result.req_off = (uint32_t)qm.rec_pool.size();
                result.n_requires = (uint32_t)(A.requires_.size() - cur.req0);
                result.n_depth_requires = (uint32_t)(A.depth_requires.size() - cur.dreq0);
                qm.rec_pool.append(A.requires_.data() + cur.req0, A.requires_.size() - cur.req0);
                qm.rec_pool.append(A.depth_requires.data() + cur.dreq0, A.depth_requires.size() - cur.dreq0);
                if (qm.pool.size() > 0xfffffff0ull || qm.rec_pool.size() > 0xfffffff0ull) throw ExecError("query table pools exceed 2^32 words");
                result.has_output = true;
                if (depth == 0) {
                    uint32_t d_top = 0;
                    for (size_t i = cur.dep0; i < A.depths.size(); i++) d_top = std::max(d_top, A.depths[i] + 1);
                    return {out, d_top};
                }
                // pop: the caller's slices are the arenas' tops again
                A.map.resize(cur.map0);
                A.requires_.resize(cur.req0);
                A.depths.resize(cur.dep0);
                A.depth_requires.resize(cur.dreq0);
                A.hints.resize(cur.hint0);
                const bool callee_partial = cur.partial, callee_preimg = cur.preimg;
                const uint32_t callee_depth = result.depth;
                depth--;
                cur = frames[depth];
                if (depth >= 2) {
                    // the activation resumed after this one was suspended long ago when the recursion is deep: ask for its
                    // variables and its table entry now
                    const Frame& up = frames[depth - 1];
                    const QueryMap& uq = q.func_queries[up.func_index];
                    __builtin_prefetch(A.map.data() + up.map0);
                    __builtin_prefetch(A.map.data() + up.map0 + 16);
                    __builtin_prefetch(&uq.vals[up.nonce]);
                    __builtin_prefetch(uq.key(up.nonce));
                    __builtin_prefetch(&frames[depth - 2]);
                }
                const List& ext = callee_preimg ? inp : out;
                A.map.append(ext.data(), ext.size());
                A.hints.append(ext.data(), ext.size());
                // `result` still refers to the callee's table slot: nothing was inserted between the Return and here
                A.requires_.push_back(result.provide.new_lookup(cur.nonce));
                if (callee_partial) A.hints.push_back(callee_depth);
                if (cur.partial && callee_partial) A.depths.push_back(callee_depth);
                break;
            }
            default:
                throw ExecError("corrupt interpreter program");
        }
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
