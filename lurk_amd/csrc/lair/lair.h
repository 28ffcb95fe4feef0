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
#include <mutex>
#include <stdexcept>
#include <cstring>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "trace_program.h"

namespace lair {

constexpr uint32_t P = 2013265921u;
inline uint32_t fadd(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % P); }
inline uint32_t fsub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + P - b) % P); }
inline uint32_t fmul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
// the same on operands known to be canonical (the interpreter's variables always are: arguments are checked at the boundary,
// constants are reduced by the compiler, everything else is a result): a compare instead of a 64-bit division
inline uint32_t fadd_c(uint32_t a, uint32_t b) {
    const uint32_t s = a + b;  // < 2^32: both operands are below p < 2^31
    return s >= P ? s - P : s;
}
inline uint32_t fsub_c(uint32_t a, uint32_t b) { return a >= b ? a - b : a + P - b; }
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
// Growable array of trivially copyable T on huge_alloc blocks (round 3: a std::vector with a custom allocator copies a range
// element by element through allocator_traits::construct -- the appends of a query's hints and requires were a scalar loop).
template <class T>
class BigVec {
    static_assert(std::is_trivially_copyable<T>::value, "BigVec holds plain data");
    T* d_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    void grow_to(size_t want) {
        size_t cap = cap_ ? cap_ * 2 : 64;
        if (cap < want) cap = want;
        T* nd = static_cast<T*>(huge_alloc(cap * sizeof(T)));
        if (n_) memcpy(static_cast<void*>(nd), d_, n_ * sizeof(T));
        if (d_) huge_free(d_, cap_ * sizeof(T));
        d_ = nd;
        cap_ = cap;
    }

   public:
    BigVec() = default;
    BigVec(const BigVec& o) { append(o.d_, o.n_); }
    BigVec(BigVec&& o) noexcept : d_(o.d_), n_(o.n_), cap_(o.cap_) { o.d_ = nullptr, o.n_ = o.cap_ = 0; }
    BigVec& operator=(const BigVec& o) {
        if (this != &o) n_ = 0, append(o.d_, o.n_);
        return *this;
    }
    BigVec& operator=(BigVec&& o) noexcept {
        if (this != &o) {
            release();
            d_ = o.d_, n_ = o.n_, cap_ = o.cap_;
            o.d_ = nullptr, o.n_ = o.cap_ = 0;
        }
        return *this;
    }
    ~BigVec() { release(); }
    void release() {
        if (d_) huge_free(d_, cap_ * sizeof(T));
        d_ = nullptr, n_ = cap_ = 0;
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T* data() { return d_; }
    const T* data() const { return d_; }
    T& operator[](size_t i) { return d_[i]; }
    const T& operator[](size_t i) const { return d_[i]; }
    T* begin() { return d_; }
    T* end() { return d_ + n_; }
    const T* begin() const { return d_; }
    const T* end() const { return d_ + n_; }
    void clear() { n_ = 0; }
    void reserve(size_t n) {
        if (n > cap_) grow_to(n);
    }
    void push_back(const T& v) {
        if (n_ == cap_) {
            const T copy = v;  // `v` may live in this array
            grow_to(n_ + 1);
            d_[n_++] = copy;
            return;
        }
        d_[n_++] = v;
    }
    // appends p[0 .. n); p may point into this array
    void append(const T* p, size_t n) {
        if (!n) return;
        if (n_ + n > cap_) {
            const bool inside = p >= d_ && p < d_ + cap_;
            const size_t off = inside ? (size_t)(p - d_) : 0;
            grow_to(n_ + n);
            if (inside) p = d_ + off;
        }
        memcpy(static_cast<void*>(d_ + n_), p, n * sizeof(T));
        n_ += n;
    }
    // shrinks, or grows with value-initialised elements
    void resize(size_t n) {
        if (n > n_) {
            if (n > cap_) grow_to(n);
            for (size_t i = n_; i < n; i++) d_[i] = T();
        }
        n_ = n;
    }
    void assign(size_t n, const T& v) {
        n_ = 0;
        if (n > cap_) grow_to(n);
        for (size_t i = 0; i < n; i++) d_[i] = v;
        n_ = n;
    }
};

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

enum BytesKind { BYTES_RANGE_U8 = 0, BYTES_RANGE_U16, BYTES_LESS_THAN, BYTES_AND, BYTES_XOR, BYTES_OR, BYTES_KINDS };  // gadgets/bytes/record.rs:50-71, order fixed by iter_records()

// gadgets/bytes/record.rs:14-17: a BTreeMap from the two input bytes to six records.  Every key of the map is a 16-bit number,
// so the map is a table: one array of 65536 records per kind (round 3: the std::map's nodes were a cache miss per lookup,
// a quarter of the interpreter's time on partial functions whose depth bytes walk through every key; a kind's array is
// 512 KiB) and a bitmap of the keys the reference's map would hold.
struct BytesRecord {
    std::vector<Record> table;      // [BYTES_KINDS][65536], empty until the first lookup
    std::vector<uint64_t> touched;  // bit per key
    uint32_t n_touched = 0;
    Record& at(uint16_t key, BytesKind k) {
        if (table.empty()) {
            table.assign((size_t)BYTES_KINDS * 65536, Record());
            touched.assign(65536 / 64, 0);
        }
        uint64_t& w = touched[key >> 6];
        const uint64_t bit = 1ull << (key & 63);
        if (!(w & bit)) w |= bit, n_touched++;
        return table[(size_t)k * 65536 + key];
    }
    // number of keys with a record (the reference map's len()) / a record as stored (zeroes when the key was never looked up)
    size_t size() const { return n_touched; }
    bool empty() const { return n_touched == 0; }
    Record get(uint16_t key, int k) const { return table.empty() ? Record() : table[(size_t)k * 65536 + key]; }
    void clear() {
        table.clear();
        touched.clear();
        n_touched = 0;
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
// The interpreter's pre-decoded form of a toplevel's functions (execute.cpp: XProgram), built on first execution.  It points
// into the toplevel's blocks, so a copy of the toplevel starts without one.
struct ExecCacheSlot {
    mutable std::mutex mu;
    mutable std::shared_ptr<const void> program;
    ExecCacheSlot() = default;
    ExecCacheSlot(const ExecCacheSlot&) {}
    ExecCacheSlot& operator=(const ExecCacheSlot&) {
        program.reset();
        return *this;
    }
};

struct Toplevel {
    std::vector<Func> funcs;          // insertion order = index
    std::unordered_map<std::string, uint32_t> func_index;
    std::vector<Chip> chips;
    std::unordered_map<std::string, uint32_t> chip_index;
    ExecCacheSlot exec_cache;

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
    BigVec<uint32_t> hashes;          // per entry: the low half of its key's hash (re-seats entries when the index grows
                                      // without reading the keys again, and screens probes before a key comparison)
    size_t size() const { return vals.size(); }
    const uint32_t* key(size_t i) const { return key_pool.data() + i * key_len; }
    // Keys are 2 .. 40 words and every Call / Store hashes one.  Round 3, first: four independent multiply chains instead of
    // one (the one-chain hash was 35 % of the interpreter under gprof); then this: 64 x 64 -> 128-bit multiply-folds over four
    // words at a time on two alternating accumulators and one fold at the end -- two multiplies for a 3-wide memory key.
    static uint64_t mum(uint64_t a, uint64_t b) {
        const unsigned __int128 r = (unsigned __int128)a * b;
        return (uint64_t)r ^ (uint64_t)(r >> 64);
    }
    static uint64_t hash(const uint32_t* k, uint32_t n) {
        constexpr uint64_t S0 = 0x9e3779b97f4a7c15ull, S1 = 0xc2b2ae3d27d4eb4full, S2 = 0x165667b19e3779f9ull, S3 = 0x85ebca77c2b2ae63ull,
                           S4 = 0xff51afd7ed558ccdull;
        uint64_t h0 = S0 ^ n, h1 = S1;
        uint32_t i = 0;
        for (; i + 8 <= n; i += 8) {
            h0 = mum((k[i] | (uint64_t)k[i + 1] << 32) ^ S2, (k[i + 2] | (uint64_t)k[i + 3] << 32) ^ h0);
            h1 = mum((k[i + 4] | (uint64_t)k[i + 5] << 32) ^ S3, (k[i + 6] | (uint64_t)k[i + 7] << 32) ^ h1);
        }
        if (i + 4 <= n) {
            h0 = mum((k[i] | (uint64_t)k[i + 1] << 32) ^ S2, (k[i + 2] | (uint64_t)k[i + 3] << 32) ^ h0);
            i += 4;
        }
        if (i < n) {
            const uint64_t a = k[i] | (i + 1 < n ? (uint64_t)k[i + 1] << 32 : 0), b = i + 2 < n ? k[i + 2] : 0;
            h1 = mum(a ^ S3, (b | (uint64_t)(n - i) << 32) ^ h1);
        }
        return mum(h0 ^ S4, h1 ^ S2);
    }
    // Absent keys are the common lookup (every new query, every fresh memory cell), and the index of a multi-million-entry
    // table is a DRAM access per probe.  So (round 3): a blocked Bloom filter beside the index -- three bits in one 64-byte
    // block per key, 8 .. 16 bits per entry, small enough to stay in the last-level cache -- answers "absent" without
    // touching the index, and a new entry's seat in the index is taken a few insertions later (`pending`), when the line
    // requested at insertion time has arrived.
    static constexpr uint32_t MAX_PENDING = 8;
    struct Pending {
        uint32_t index, h32;
    };
    BigVec<uint64_t> bloom;           // 8 words per block; bits = 4 x slots
    uint32_t bloom_shift = 32;        // block = (h32 * K) >> bloom_shift
    Pending pending[MAX_PENDING];
    uint32_t pending_head = 0, n_pending = 0;
    bool bloom_maybe(uint32_t h32) const {
        const uint64_t* blk = bloom.data() + ((size_t)((h32 * 0x9e3779b1u) >> bloom_shift) << 3);
        const uint32_t m = h32 * 0x85ebca6bu;
        const uint32_t b0 = m & 511, b1 = (m >> 9) & 511, b2 = (m >> 18) & 511;
        return ((blk[b0 >> 6] >> (b0 & 63)) & (blk[b1 >> 6] >> (b1 & 63)) & (blk[b2 >> 6] >> (b2 & 63)) & 1) != 0;
    }
    void bloom_add(uint32_t h32) {
        uint64_t* blk = bloom.data() + ((size_t)((h32 * 0x9e3779b1u) >> bloom_shift) << 3);
        const uint32_t m = h32 * 0x85ebca6bu;
        const uint32_t b0 = m & 511, b1 = (m >> 9) & 511, b2 = (m >> 18) & 511;
        blk[b0 >> 6] |= 1ull << (b0 & 63);
        blk[b1 >> 6] |= 1ull << (b1 & 63);
        blk[b2 >> 6] |= 1ull << (b2 & 63);
    }
    int find_hashed(const uint32_t* k, uint32_t n, uint64_t h) const {
        if (slots.empty() || n != key_len) return -1;
        if (!bloom_maybe((uint32_t)h)) return -1;
        for (uint32_t j = 0; j < n_pending; j++) {
            const Pending& p = pending[(pending_head + j) % MAX_PENDING];
            if (p.h32 == (uint32_t)h && (n == 0 || memcmp(key(p.index), k, (size_t)n * 4) == 0)) return (int)p.index;
        }
        const size_t mask = slots.size() - 1;
        for (size_t s = h & mask;; s = (s + 1) & mask) {
            const uint32_t e = slots[s];
            if (!e) return -1;
            if (hashes[e - 1] == (uint32_t)h && (n == 0 || memcmp(key(e - 1), k, (size_t)n * 4) == 0)) return (int)(e - 1);
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
        hashes.clear();
        bloom.clear();
        bloom_shift = 32;
        pending_head = n_pending = 0;
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
