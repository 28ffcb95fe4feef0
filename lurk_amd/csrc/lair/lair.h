// Lair on the host: IR, bytecode, toplevel, interpreter state and column layout.
//
// Host-side mirror (C++, because the reference's Rust toolchain is absent here) of the parts of
// /root/reference/src/lair/ that feed the trace kernels:
//   expr.rs      -> FuncE / BlockE / OpE / CtrlE          (named IR)
//   macros.rs    -> parse_funcs(): a text form of the `func!` DSL with the same surface syntax
//   toplevel.rs  -> Toplevel: expand + compile to index-based bytecode (toplevel.rs:241-879)
//   bytecode.rs  -> Func / Block / Op / Ctrl
//   execute.rs   -> QueryRecord, execute() (memoising interpreter, execute.rs:436-784), sharding
//   func_chip.rs -> LayoutSizes, compute_layout_sizes (func_chip.rs:90-276)
// Values are canonical BabyBear u32 on the host; the device converts at its boundary.
#pragma once
#include <stdint.h>

#include <string.h>

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "trace_program.h"

namespace lair {

constexpr uint32_t P = 2013265921u;
inline uint32_t fadd(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % P); }
inline uint32_t fsub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + P - b) % P); }
inline uint32_t fmul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
uint32_t finv(uint32_t a);  // panics (throws) on zero like p3's inverse()
inline uint32_t field_from_i64(int64_t v) {
    int64_t m = v % (int64_t)P;
    if (m < 0) m += P;
    return (uint32_t)m;
}

struct ExecError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

using List = std::vector<uint32_t>;

// Allocator of the multi-gigabyte tables of a big execution (query pools, interpreter arenas).  Blocks of 4 MiB and more are
// mapped directly, 2 MiB aligned and advised as transparent huge pages: first-touch page faults of 4 KiB pages cost more than
// the interpreter's own work (measured: 3.5 s of system time in a 5.5 s execution of 0.8 M queries).
void* huge_alloc(size_t bytes);
void huge_free(void* p, size_t bytes);
template <class T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U>
    HugeAlloc(const HugeAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(huge_alloc(n * sizeof(T))); }
    void deallocate(T* p, size_t n) { huge_free(p, n * sizeof(T)); }
    template <class U>
    bool operator==(const HugeAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const HugeAlloc<U>&) const { return false; }
};
template <class T>
using BigVec = std::vector<T, HugeAlloc<T>>;

// ---------------------------------------------------------------- IR (expr.rs)
struct Var {
    std::string name;  // user names as written; internal names start with '$'
    int size = 1;
    bool operator==(const Var& o) const { return name == o.name && size == o.size; }
};

enum class OpEKind {
    AssertEq, AssertNe, Contains, Const, Array, Add, Sub, Mul, Div, Inv, Not, Eq,
    Call, PreImg, Store, Load, Slice, ExternCall, Emit, RangeU8, Breakpoint, Debug
};

struct OpE {
    OpEKind kind;
    std::vector<Var> out;   // targets (Const/Array/arith: 1; Call/Load/...: n)
    std::vector<Var> in;    // operands
    std::string name;       // callee / chip name
    List consts;            // Const (1) / Array (n)
};

enum class CaseType { Constrained, Unconstrained };

struct BlockE;
struct CaseE {
    List keys;  // Match: all values mapping to this branch; MatchMany: one pattern
    std::shared_ptr<BlockE> block;
    CaseType constrained = CaseType::Constrained;
};

enum class CtrlEKind { Match, MatchMany, Choose, ChooseMany, If, Return };

struct CtrlE {
    CtrlEKind kind = CtrlEKind::Return;
    Var var;                        // scrutinee
    std::vector<CaseE> branches;    // for If: unused
    std::shared_ptr<BlockE> def;    // default case
    CaseType def_constrained = CaseType::Constrained;
    std::shared_ptr<BlockE> t, f;   // If
    std::vector<Var> ret;           // Return
};

struct BlockE {
    std::vector<OpE> ops;
    CtrlE ctrl;
};

struct FuncE {
    std::string name;
    bool invertible = false;
    bool partial = false;
    std::vector<Var> input_params;
    int output_size = 0;
    BlockE body;
};

// Parses one or more `fn` definitions written in the func! surface syntax (macros.rs).
// `symbols` resolves identifiers used as match patterns / constants (e.g. Lurk tags).
std::vector<FuncE> parse_funcs(const std::string& src, const std::map<std::string, uint32_t>* symbols = nullptr);

// ---------------------------------------------------------------- bytecode (bytecode.rs)
enum class OpKind {
    AssertEq, AssertNe, Contains, Const, Add, Sub, Mul, Inv, Not, Call, PreImg, Store, Load, ExternCall,
    Emit, RangeU8, Breakpoint, Debug
};

struct Op {
    OpKind kind;
    std::vector<uint32_t> a;  // AssertEq/Ne: lhs; Contains: array; Call/PreImg/Store/Extern/Emit/RangeU8: args
    std::vector<uint32_t> b;  // AssertEq/Ne: rhs
    uint32_t x = 0, y = 0;    // Add/Sub/Mul: operands; Inv/Not: x; Contains: y = needle; Load: x = len, y = ptr;
                              // Call/PreImg/ExternCall: x = callee / chip index
    uint32_t c = 0;           // Const value
};

struct Block;
struct Ctrl {
    enum Kind { Choose, ChooseMany, Return } kind = Return;
    uint32_t var = 0;                 // Choose
    std::vector<uint32_t> vars;       // ChooseMany
    // sorted (key -> branch); several keys may point at the same block (shared_ptr identity)
    std::vector<std::pair<List, std::shared_ptr<Block>>> branches;
    std::vector<std::shared_ptr<Block>> unique_branches;  // Choose only: one per source branch
    std::shared_ptr<Block> def;
    uint32_t ident = 0;               // Return: selector index
    std::vector<uint32_t> ret;        // Return: output vars
    const Block* match_case(const List& key) const;
};

struct Block {
    std::vector<Op> ops;
    Ctrl ctrl;
    std::vector<uint32_t> return_idents;
};

struct Func {
    std::string name;
    bool invertible = false, partial = false;
    uint32_t index = 0, input_size = 0, output_size = 0;
    Block body;
};

// ---------------------------------------------------------------- chipsets (chipset.rs, core/chipset.rs)
struct Record {  // air/builder.rs:135-150
    uint32_t nonce = 0, count = 0;
    Record new_lookup(uint32_t n) {
        Record r = *this;
        nonce = n;
        count += 1;
        return r;
    }
};

struct BytesInputRecord {  // gadgets/bytes/record.rs:50-71, order fixed by iter_records()
    Record range_u8, range_u16, less_than, and_, xor_, or_;
};

struct BytesRecord {  // gadgets/bytes/record.rs:14-17
    std::map<uint16_t, BytesInputRecord> records;   // ordered: the trace rows are emitted in key order
    std::vector<BytesInputRecord*> slot;            // direct index into `records` (map nodes never move)
    BytesRecord() = default;
    BytesRecord(const BytesRecord& o) : records(o.records) {}  // the index points into the source's nodes: rebuilt on demand
    BytesRecord& operator=(const BytesRecord& o) {
        records = o.records;
        slot.clear();
        return *this;
    }
    BytesInputRecord& at(uint16_t key) {
        if (slot.empty()) slot.assign(65536, nullptr);
        BytesInputRecord*& p = slot[key];
        if (!p) p = &records[key];
        return *p;
    }
    void clear() {
        records.clear();
        slot.clear();
    }
    void range_check_u8_pair(uint8_t i1, uint8_t i2, uint32_t nonce, std::vector<Record>& requires_);
    void range_check_u8_iter(const uint8_t* bytes, size_t n, uint32_t nonce, std::vector<Record>& requires_);
    bool less_than(uint8_t i1, uint8_t i2, uint32_t nonce, std::vector<Record>& requires_);
    void range_check_u16(uint16_t v, uint32_t nonce, std::vector<Record>& requires_);
};

struct Chip {
    std::string name;
    ChipKind kind = CHIP_NONE;
    uint32_t input_size = 0, output_size = 0, witness_size = 0, require_size = 0;
    // number of values populate_witness returns (pushed to the variable map during trace
    // generation; the reference's Poseidon chip returns the whole state, core/poseidon.rs:65-72)
    uint32_t witness_return_size = 0;
    // host execution: returns outputs and appends byte-lookup requires
    List execute(const List& input, uint32_t nonce, BytesRecord& bytes, std::vector<Record>& requires_) const;
};

std::vector<Chip> lurk_chip_map();  // core/chipset.rs:28-63, in that order

// ---------------------------------------------------------------- toplevel (toplevel.rs)
struct Toplevel {
    std::vector<Func> funcs;          // insertion order = index
    std::unordered_map<std::string, uint32_t> func_index;
    std::vector<Chip> chips;
    std::unordered_map<std::string, uint32_t> chip_index;

    static Toplevel build(const std::vector<FuncE>& funcs, const std::vector<Chip>& chips);
    const Func& func_by_name(const std::string& n) const;
};

// Flat u32 serialisation of the compiled functions (format "LBC1", bytecode_io.cpp): export of this compiler's bytecode and
// import of another compiler's (the reference's Rust Toplevel), validated.
std::vector<uint32_t> toplevel_to_bytecode(const Toplevel& t);
Toplevel toplevel_from_bytecode(const uint32_t* words, size_t n_words);

// ---------------------------------------------------------------- layout (func_chip.rs)
struct LayoutSizes {
    uint32_t nonce = 1, input = 0, output = 0, aux = 0, sel = 0;
    uint32_t total() const { return nonce + input + output + aux + sel; }
};
LayoutSizes compute_layout_sizes(const Toplevel& t, const Func& f);

constexpr int DEPTH_W = 4;                 // provenance.rs:11
constexpr int DEPTH_LESS_THAN_SIZE = 6;    // provenance.rs:121
constexpr int DEPTH_LT_REQUIRES = 1;

// ---------------------------------------------------------------- execution (execute.rs)
// Result of one query.  Everything of variable length lives in the owning QueryMap's pools (one allocation per
// table instead of four per query: the interpreter is memory-bound on multi-million-query executions).
struct QueryResult {
    bool has_output = false;
    uint32_t depth = 0;
    Record provide;
    uint32_t out_off = 0;                 // QueryMap::pool: the output (length = the function's output size)
    uint32_t hint_off = 0, n_hints = 0;   // QueryMap::pool: values the row's trace needs that the reference re-derives by
                                          // hash-map lookups at trace time (callee outputs, preimages, pointers, loaded
                                          // values, callee depths), in bytecode order
    uint32_t req_off = 0, n_requires = 0, n_depth_requires = 0;  // QueryMap::rec_pool: requires, then depth requires
};

struct VecHash {
    size_t operator()(const List& v) const {
        uint64_t h = 0x9e3779b97f4a7c15ull ^ v.size();
        for (uint32_t x : v) {
            h = (h ^ x) * 0xff51afd7ed558ccdull;
            h ^= h >> 29;
        }
        return (size_t)h;
    }
};

// Insertion-ordered map key -> QueryResult (indexmap::IndexMap): flat key pool (all keys of a table have the same
// length: a function's inputs, a memory table's width), open-addressing index.
struct QueryMap {
    uint32_t key_len = 0;
    BigVec<uint32_t> key_pool;        // [n][key_len]
    BigVec<QueryResult> vals;
    BigVec<uint32_t> pool;            // outputs and hints
    BigVec<Record> rec_pool;          // require records
    BigVec<uint32_t> slots;           // entry index + 1, 0 = empty; power-of-two size
    size_t size() const { return vals.size(); }
    const uint32_t* key(size_t i) const { return key_pool.data() + i * key_len; }
    // Four independent multiply chains over the key words, folded at the end (round 3: the one-chain hash -- xor, 64-bit
    // multiply, shift-xor per word, each waiting for the last -- was 35 % of the interpreter under gprof: keys are 8 .. 40 words
    // and every Call hashes one, every new query three).
    static uint64_t hash(const uint32_t* k, uint32_t n) {
        constexpr uint64_t C = 0xff51afd7ed558ccdull;
        uint64_t h0 = 0x9e3779b97f4a7c15ull ^ n, h1 = 0xc2b2ae3d27d4eb4full, h2 = 0x165667b19e3779f9ull, h3 = 0x85ebca77c2b2ae63ull;
        uint32_t i = 0;
        for (; i + 4 <= n; i += 4) {
            h0 = (h0 ^ k[i]) * C;
            h1 = (h1 ^ k[i + 1]) * C;
            h2 = (h2 ^ k[i + 2]) * C;
            h3 = (h3 ^ k[i + 3]) * C;
        }
        for (; i < n; i++) h0 = ((h0 ^ k[i]) * C) ^ (h0 >> 31);
        uint64_t h = (h0 ^ (h1 >> 17) ^ (h1 << 47)) * C;
        h = (h ^ h2 ^ (h3 >> 29) ^ (h3 << 35)) * C;
        return h ^ (h >> 32);
    }
    int find_hashed(const uint32_t* k, uint32_t n, uint64_t h) const {
        if (slots.empty() || n != key_len) return -1;
        const size_t mask = slots.size() - 1;
        for (size_t s = h & mask;; s = (s + 1) & mask) {
            const uint32_t e = slots[s];
            if (!e) return -1;
            if (memcmp(key(e - 1), k, (size_t)n * 4) == 0) return (int)(e - 1);
        }
    }
    int find(const uint32_t* k, uint32_t n) const { return slots.empty() || n != key_len ? -1 : find_hashed(k, n, hash(k, n)); }
    int find(const List& k) const { return find(k.data(), (uint32_t)k.size()); }
    // appends a new entry (the key must be absent); returns its index.  push_hashed: with the key's hash already at hand (a
    // lookup that missed is followed by the insertion of the same key).
    uint32_t push(const uint32_t* k, uint32_t n, const QueryResult& v) { return push_hashed(k, n, v, hash(k, n)); }
    uint32_t push_hashed(const uint32_t* k, uint32_t n, const QueryResult& v, uint64_t h);
    // IndexMap::insert_full: replaces the value when the key exists
    uint32_t insert_full(const List& k, const QueryResult& v) {
        int i = find(k);
        if (i >= 0) {
            vals[i] = v;
            return (uint32_t)i;
        }
        return push(k.data(), (uint32_t)k.size(), v);
    }
    const uint32_t* output(const QueryResult& r) const { return pool.data() + r.out_off; }
    const uint32_t* hints(const QueryResult& r) const { return pool.data() + r.hint_off; }
    const Record* requires_of(const QueryResult& r) const { return rec_pool.data() + r.req_off; }
    void clear() {
        key_pool.clear();
        vals.clear();
        pool.clear();
        rec_pool.clear();
        slots.clear();
    }

   private:
    void grow();
};

constexpr int NUM_MEM_TABLES = 6;
extern const uint32_t MEM_TABLE_SIZES[NUM_MEM_TABLES];  // {2,3,4,5,6,8}, execute.rs:243-244
int mem_index_from_len(uint32_t len);

struct QueryRecord {
    bool has_public_values = false;
    List public_values;
    std::vector<QueryMap> func_queries;
    std::vector<std::unique_ptr<std::unordered_map<List, List, VecHash>>> inv_func_queries;  // null if not invertible
    std::vector<QueryMap> mem_queries;
    BytesRecord bytes;
    std::vector<List> emitted;

    explicit QueryRecord(const Toplevel& t);
    void clean();
    void inject_inv_query(uint32_t func_idx, const List& inp, const List& out);
};

// Toplevel::execute (execute.rs:375-392): runs `func` on `args`, fills `record`, sets public values.
List execute(const Toplevel& t, const Func& func, const List& args, QueryRecord& record);

struct ShardingConfig {
    uint32_t max_shard_size = 1u << 22;  // execute.rs:231-241
};
inline std::pair<size_t, size_t> shard_range(size_t num_queries, uint32_t shard_index, uint32_t max_shard_size) {
    size_t start = (size_t)shard_index * max_shard_size;
    size_t end = std::min((size_t)(shard_index + 1) * max_shard_size, num_queries);
    // like Rust's `start..end` with start > end: empty, but `start` keeps its value (it seeds the nonces)
    if (end < start) end = start;
    return {start, end};
}
size_t num_shards(const QueryRecord& r, uint32_t max_shard_size);  // execute.rs:186-216

// ---------------------------------------------------------------- device program (emit.cpp)
// Flattened, degree-resolved micro-program of one Func for the trace kernel; see trace_program.h.
std::vector<uint32_t> build_trace_program(const Toplevel& t, const Func& f, uint32_t* max_vars);

}  // namespace lair
