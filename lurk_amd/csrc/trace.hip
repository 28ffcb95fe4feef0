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
#include "babybear.h"
#include "ctx.h"
#include "lair/trace_program.h"
#include "poseidon2_dev.h"

namespace {

using namespace lair;

constexpr int TBLOCK = 64;

struct TraceArgs {
    const uint32_t* prog;
    const uint32_t* args;      // [n][input]
    const uint32_t* outputs;   // [n][output]
    const uint32_t* provides;  // [n][2]  (last_nonce, last_count)
    const uint32_t* depths;    // [n] or null
    const RowMeta* meta;       // [n]
    const uint32_t* stream;
    uint32_t* out;             // [height][width]
    uint32_t n_real;
    uint32_t height;
    uint32_t nonce_start;
    int canonical_out;
};

// Column col of the lane's row lives at base[e + (e >> sh)], e = e0 + col.  Staged (the workgroup's rows go through LDS and
// leave with coalesced stores): base = the LDS tile, e0 = lane * width, sh = 5 (one pad word per 32 keeps a column of 64
// rows off a single bank whatever the width).  Unstaged (rows wider than the tile budget): base = the row in global
// memory, e0 = 0, sh = 31.
struct RowWriter {
    uint32_t* row;
    uint32_t aux0;   // column of aux[0]
    uint32_t aux;    // aux cursor
    bool canonical;
    uint32_t e0 = 0, sh = 31;
    __device__ __forceinline__ uint32_t& at(uint32_t col) {
        const uint32_t e = e0 + col;
        return row[e + (e >> sh)];
    }
    __device__ __forceinline__ void put(uint32_t col, uint32_t v_m) { at(col) = canonical ? bb::from_monty(v_m) : v_m; }
    __device__ __forceinline__ void push_aux(uint32_t v_m) { put(aux0 + aux++, v_m); }
    // small non-negative integers (bytes, nonces, counts) given as plain integers
    __device__ __forceinline__ void put_int(uint32_t col, uint32_t v) { at(col) = canonical ? v : bb::to_monty(v); }
    __device__ __forceinline__ void push_aux_int(uint32_t v) { put_int(aux0 + aux++, v); }
};

// Inverses of the small integers (Montgomery form), built at compile time: the lookup counts whose successors a require
// record inverts (air/builder.rs:162) are almost always a handful, and a Fermat ladder is 40 products per record.
constexpr int INV_TABLE = 1024;
struct InvTable {
    uint32_t v[INV_TABLE];
};
constexpr uint32_t c_pow(uint32_t a, uint32_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1u) r = bb::cmulmod(r, a);
        a = bb::cmulmod(a, a);
        e >>= 1;
    }
    return r;
}
constexpr InvTable make_inv_table() {
    InvTable t{};
    t.v[0] = 0;
    for (int i = 1; i < INV_TABLE; i++) t.v[i] = bb::c_to_monty(c_pow((uint32_t)i, bb::P - 2));
    return t;
}
__constant__ const InvTable kInvSmall = make_inv_table();

// RequireRecord: prev_nonce, prev_count, (prev_count + 1)^-1   (air/builder.rs:159-168)
__device__ __forceinline__ void push_require(RowWriter& w, const uint32_t* rec) {
    uint32_t nonce = rec[0], count = rec[1];
    w.push_aux_int(nonce);
    w.push_aux_int(count);
    const uint32_t c1 = count + 1;
    w.push_aux(c1 < (uint32_t)INV_TABLE ? kInvSmall.v[c1] : bb::inv(bb::to_monty(c1)));
}

// Poseidon2Cols recorder writing straight into the row (core/poseidon.rs:65-72: 8 outputs first)
template <int W, int RP>
struct RowRec {
    RowWriter* w;
    uint32_t base;  // column of the first witness lane (the 8 outputs)
    __device__ __forceinline__ void ext_state(int r, int i, uint32_t v) { w->put(base + 8 + r * W + i, v); }
    __device__ __forceinline__ void end_ext_state(int) {}
    __device__ __forceinline__ void ext_sbox(int r, int i, uint32_t v) { w->put(base + 8 + 8 * W + r * W + i, v); }
    __device__ __forceinline__ void end_ext_sbox(int) {}
    __device__ __forceinline__ void int_init(int i, uint32_t v) { w->put(base + 8 + 16 * W + i, v); }
    __device__ __forceinline__ void end_int_init() {}
    __device__ __forceinline__ void int_state0(int r, uint32_t v) { w->put(base + 8 + 17 * W + r, v); }
    __device__ __forceinline__ void int_sbox(int r, uint32_t v) { w->put(base + 8 + 17 * W + (RP - 1) + r, v); }
    __device__ __forceinline__ void end_internal() {}
};

template <int W, class MapT>
__device__ __noinline__ void extern_hasher(RowWriter& w, MapT& map, uint32_t& sp, const uint32_t* ins) {
    constexpr int RP = p2::Cfg<W>::RP;
    uint32_t s[W];
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = map[ins[i]];
    RowRec<W, RP> rec{&w, w.aux0 + w.aux};
    const auto& p = p2::Cfg<W>::params();
    p2::permute_core<W>(s, RP, p.ext_rc, p.int_rc, p.diag, p.ext_rc_mp, p.int_rc_mp, p.diag_c, rec);
#pragma unroll
    for (int i = 0; i < 8; i++) w.put(rec.base + i, s[i]);
    w.aux += 8 + p2::Cfg<W>::NUM_COLS;
    // populate_witness returns the whole state (core/poseidon.rs:71) and trace.rs:393-396 pushes all of it
#pragma unroll
    for (int i = 0; i < W; i++) map[sp++] = s[i];
}

template <class MapT>
__device__ __forceinline__ uint64_t map_u64(MapT& map, const uint32_t* ins) {
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r |= (uint64_t)(bb::from_monty(map[ins[i]]) & 0xff) << (8 * i);
    return r;
}

// LessThanWitness<_, 4> for depths (unsigned/less_than.rs:12-41): is_comp[4], lhs_limb, rhs_limb
__device__ __forceinline__ void push_depth_less_than(RowWriter& w, uint32_t lhs, uint32_t rhs) {
    int idx = -1;
    for (int i = 3; i >= 0; i--) {
        if (((lhs >> (8 * i)) & 0xff) != ((rhs >> (8 * i)) & 0xff)) {
            idx = i;
            break;
        }
    }
    for (int i = 0; i < 4; i++) w.push_aux_int(i == idx ? 1u : 0u);
    w.push_aux_int(idx >= 0 ? (lhs >> (8 * idx)) & 0xff : 0u);
    w.push_aux_int(idx >= 0 ? (rhs >> (8 * idx)) & 0xff : 0u);
}

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
            const uint32_t kind = prog[pc + 1], nin = prog[pc + 2], wit = prog[pc + 3], nreq = prog[pc + 4];
            const uint32_t* ins_v = prog + pc + 6;
            if (kind == CHIP_HASHER3) extern_hasher<24>(w, map, sp, ins_v);
            else if (kind == CHIP_HASHER4) extern_hasher<32>(w, map, sp, ins_v);
            else if (kind == CHIP_HASHER5) extern_hasher<40>(w, map, sp, ins_v);
            else if (kind == CHIP_U64_ADD || kind == CHIP_U64_SUB) {
                uint64_t x = map_u64(map, ins_v), y = map_u64(map, ins_v + 8);
                uint64_t z = kind == CHIP_U64_ADD ? x + y : x - y;
                for (int i = 0; i < 8; i++) {
                    uint32_t b = (uint32_t)(z >> (8 * i)) & 0xff;
                    w.push_aux_int(b);
                    map[sp++] = bb::to_monty(b);
                }
            } else if (kind == CHIP_U64_MUL) {
                uint64_t x = map_u64(map, ins_v), y = map_u64(map, ins_v + 8);
                uint32_t carry = 0;
                uint32_t res[8];
                for (int k = 0; k < 8; k++) {
                    uint32_t prod = 0;
                    for (int i = 0; i <= k; i++) prod += (uint32_t)((x >> (8 * i)) & 0xff) * (uint32_t)((y >> (8 * (k - i))) & 0xff);
                    uint32_t o = prod + carry;
                    res[k] = o & 0xff;
                    carry = (o >> 8) & 0xffff;
                    w.push_aux_int(carry);
                }
                for (int k = 0; k < 8; k++) {
                    w.push_aux_int(res[k]);
                    map[sp++] = bb::to_monty(res[k]);
                }
            } else if (kind == CHIP_U64_LESSTHAN) {
                // CompareWitness<_, 8>: is_comp[8], lhs_limb, rhs_limb, diff_inv, is_less_than
                uint64_t x = map_u64(map, ins_v), y = map_u64(map, ins_v + 8);
                int idx = -1;
                for (int i = 7; i >= 0; i--)
                    if (((x >> (8 * i)) & 0xff) != ((y >> (8 * i)) & 0xff)) {
                        idx = i;
                        break;
                    }
                uint32_t l = idx >= 0 ? (uint32_t)(x >> (8 * idx)) & 0xff : 0, rr = idx >= 0 ? (uint32_t)(y >> (8 * idx)) & 0xff : 0;
                for (int i = 0; i < 8; i++) w.push_aux_int(i == idx ? 1u : 0u);
                w.push_aux_int(l);
                w.push_aux_int(rr);
                w.push_aux(idx >= 0 ? bb::inv(bb::sub(bb::to_monty(l), bb::to_monty(rr))) : 0u);
                uint32_t lt = (idx >= 0 && l < rr) ? 1u : 0u;
                w.push_aux_int(lt);
                map[sp++] = bb::to_monty(lt);
            } else if (kind == CHIP_U64_ISZERO) {
                // IsZero<_, 8>: inverses[8] (only the first non-zero limb), result
                uint64_t x = map_u64(map, ins_v);
                bool found = false;
                for (int i = 0; i < 8; i++) {
                    uint32_t limb = (uint32_t)(x >> (8 * i)) & 0xff;
                    if (!found && limb) {
                        w.push_aux(bb::inv(bb::to_monty(limb)));
                        found = true;
                    } else {
                        w.push_aux(0u);
                    }
                }
                uint32_t z = x == 0 ? 1u : 0u;
                w.push_aux_int(z);
                map[sp++] = bb::to_monty(z);
            } else if (kind == CHIP_U64_DIVREM) {
                // DivRem<_, 8> (unsigned/div_rem.rs:16-62): b_non_zero.inverses[8], q[8], qb { carry[8], result[8] },
                // r[8], r_lt_b { is_comp[8], lhs, rhs }, qb_cmp_a { is_comp[8], lhs, rhs, diff_inv, is_less_than }
                const uint64_t x = map_u64(map, ins_v), y = map_u64(map, ins_v + 8);
                const uint64_t qv = y ? x / y : 0, qb = qv * y, rem = x - qb;
                bool found = false;
                for (int i = 0; i < 8; i++) {
                    const uint32_t limb = (uint32_t)(y >> (8 * i)) & 0xff;
                    if (!found && limb) {
                        w.push_aux(bb::inv(bb::to_monty(limb)));
                        found = true;
                    } else {
                        w.push_aux(0u);
                    }
                }
                for (int i = 0; i < 8; i++) w.push_aux_int((uint32_t)(qv >> (8 * i)) & 0xff);
                {
                    uint32_t carry = 0, res[8];
                    for (int k = 0; k < 8; k++) {
                        uint32_t prod = 0;
                        for (int i = 0; i <= k; i++) prod += (uint32_t)((qv >> (8 * i)) & 0xff) * (uint32_t)((y >> (8 * (k - i))) & 0xff);
                        const uint32_t o = prod + carry;
                        res[k] = o & 0xff;
                        carry = (o >> 8) & 0xffff;
                        w.push_aux_int(carry);
                    }
                    for (int k = 0; k < 8; k++) w.push_aux_int(res[k]);
                }
                for (int i = 0; i < 8; i++) w.push_aux_int((uint32_t)(rem >> (8 * i)) & 0xff);
                auto msb_diff = [](uint64_t l, uint64_t r) {
                    for (int i = 7; i >= 0; i--)
                        if (((l >> (8 * i)) & 0xff) != ((r >> (8 * i)) & 0xff)) return i;
                    return -1;
                };
                {  // LessThanWitness(rem, y)
                    const int idx = msb_diff(rem, y);
                    for (int i = 0; i < 8; i++) w.push_aux_int(i == idx ? 1u : 0u);
                    w.push_aux_int(idx >= 0 ? (uint32_t)(rem >> (8 * idx)) & 0xff : 0u);
                    w.push_aux_int(idx >= 0 ? (uint32_t)(y >> (8 * idx)) & 0xff : 0u);
                }
                {  // CompareWitness(qb, x)
                    const int idx = msb_diff(qb, x);
                    const uint32_t l = idx >= 0 ? (uint32_t)(qb >> (8 * idx)) & 0xff : 0, rr = idx >= 0 ? (uint32_t)(x >> (8 * idx)) & 0xff : 0;
                    for (int i = 0; i < 8; i++) w.push_aux_int(i == idx ? 1u : 0u);
                    w.push_aux_int(l);
                    w.push_aux_int(rr);
                    w.push_aux(idx >= 0 ? bb::inv(bb::sub(bb::to_monty(l), bb::to_monty(rr))) : 0u);
                    w.push_aux_int((idx >= 0 && l < rr) ? 1u : 0u);
                }
                for (int i = 0; i < 8; i++) map[sp++] = bb::to_monty((uint32_t)(qv >> (8 * i)) & 0xff);
                for (int i = 0; i < 8; i++) map[sp++] = bb::to_monty((uint32_t)(rem >> (8 * i)) & 0xff);
            } else if (kind == CHIP_BIGNUM_LESSTHAN) {
                // BigNumCompareWitness (big_num/cmp.rs:13-49): is_comp[8], lhs_limb, rhs_limb, lhs_word { is_msb_lt, bytes[4] },
                // rhs_word { .. }, CompareWitness<_, 4> { is_comp[4], lhs, rhs, diff_inv, is_less_than }
                int idx = -1;
                uint32_t lm = 0, rm = 0;
                for (int i = 7; i >= 0; i--)
                    if (map[ins_v[i]] != map[ins_v[8 + i]]) {
                        idx = i;
                        lm = map[ins_v[i]];
                        rm = map[ins_v[8 + i]];
                        break;
                    }
                const uint32_t l = idx >= 0 ? bb::from_monty(lm) : 0u, r = idx >= 0 ? bb::from_monty(rm) : 0u;
                for (int i = 0; i < 8; i++) w.push_aux_int(i == idx ? 1u : 0u);
                w.push_aux_int(l);
                w.push_aux_int(r);
                for (int side = 0; side < 2; side++) {
                    const uint32_t v = side ? r : l;
                    w.push_aux_int((v >> 24) < 0x78 ? 1u : 0u);
                    for (int i = 0; i < 4; i++) w.push_aux_int((v >> (8 * i)) & 0xff);
                }
                int j = -1;
                for (int i = 3; i >= 0; i--)
                    if (((l >> (8 * i)) & 0xff) != ((r >> (8 * i)) & 0xff)) {
                        j = i;
                        break;
                    }
                const uint32_t lb = j >= 0 ? (l >> (8 * j)) & 0xff : 0, rb = j >= 0 ? (r >> (8 * j)) & 0xff : 0;
                for (int i = 0; i < 4; i++) w.push_aux_int(i == j ? 1u : 0u);
                w.push_aux_int(lb);
                w.push_aux_int(rb);
                w.push_aux(j >= 0 ? bb::inv(bb::sub(bb::to_monty(lb), bb::to_monty(rb))) : 0u);
                const uint32_t lt = (j >= 0 && lb < rb) ? 1u : 0u;
                w.push_aux_int(lt);
                map[sp++] = bb::to_monty(lt);
            } else {
                // unsupported chips are rejected on the host before launch
                w.aux += wit;
            }
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

// One row per lane.  STAGED: the workgroup's 64 rows are built in a zero-filled LDS tile and leave as one contiguous run
// of 64 * width words with coalesced stores (a lane writing its own row straight to HBM touches 64 lines per store
// instruction: 4x write amplification measured); the output needs no memset then.
template <int CAP, bool STAGED>
__global__ __launch_bounds__(TBLOCK) void k_trace_func(TraceArgs a) {
    extern __shared__ uint32_t tile[];
    const uint32_t row0 = blockIdx.x * TBLOCK, row_i = row0 + threadIdx.x;
    const uint32_t width = a.prog[TH_WIDTH], n_in = a.prog[TH_INPUT], n_out = a.prog[TH_OUTPUT];
    if constexpr (STAGED) {
        const uint32_t rows = a.height - row0 < (uint32_t)TBLOCK ? a.height - row0 : (uint32_t)TBLOCK;
        const uint32_t words = rows * width, padded = TBLOCK * width + ((TBLOCK * width) >> 5) + 1;
        for (uint32_t e = threadIdx.x; e < padded; e += TBLOCK) tile[e] = 0;
        __syncthreads();
        if (row_i < a.height) {
            RowWriter w{tile, 1 + n_in + n_out, 0, a.canonical_out != 0, threadIdx.x * width, 5};
            trace_row<CAP>(a, row_i, w);
        }
        __syncthreads();
        uint32_t* __restrict__ dst = a.out + (size_t)row0 * width;
        for (uint32_t e = threadIdx.x; e < words; e += TBLOCK) dst[e] = tile[e + (e >> 5)];
    } else {
        if (row_i >= a.height) return;
        RowWriter w{a.out + (size_t)row_i * width, 1 + n_in + n_out, 0, a.canonical_out != 0};
        trace_row<CAP>(a, row_i, w);
    }
}

// ---- MemChip (memory.rs:30-69): [is_real = 1, ptr = i + 1, last_nonce, last_count, values...] -----------
__global__ void k_trace_mem(const uint32_t* __restrict__ values, const uint32_t* __restrict__ provides, uint32_t len,
                            uint32_t n_real, uint32_t height, uint32_t* __restrict__ out, int canonical) {
    const uint32_t width = 4 + len;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)height * width) return;
    uint32_t row = (uint32_t)(e / width), c = (uint32_t)(e - (size_t)row * width);
    uint32_t v = 0;
    if (row < n_real) {
        if (c == 0) v = 1;
        else if (c == 1) v = row + 1;
        else if (c < 4) v = provides[2 * (size_t)row + (c - 2)];
        else v = values[(size_t)row * len + (c - 4)];
    }
    out[e] = canonical ? v : bb::to_monty(v);
}

// ---- BytesChip main trace (bytes/trace.rs:75-101): [is_real, 6 x (last_nonce, last_count)] ----------------
// records: [65536][12] (range_u8, range_u16, less_than, and, xor, or) x (nonce, count); all-zero rows = never required
__global__ void k_trace_bytes(const uint32_t* __restrict__ records, int is_real, uint32_t* __restrict__ out, int canonical) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)65536 * 13) return;
    uint32_t row = (uint32_t)(e / 13), c = (uint32_t)(e - (size_t)row * 13);
    uint32_t v = 0;
    if (is_real) v = c == 0 ? 1u : records[(size_t)row * 12 + (c - 1)];
    out[e] = canonical ? v : bb::to_monty(v);
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

extern "C" {

int32_t lurkhip_trace_func_dev(lurkhip_ctx* ctx, const uint32_t* program_dev, const uint32_t* program_host_header,
                               uint32_t n_real, uint32_t height, uint32_t nonce_start, const uint32_t* args_dev,
                               const uint32_t* outputs_dev, const uint32_t* provides_dev, const uint32_t* depths_dev,
                               const void* meta_dev, const uint32_t* stream_dev, uint32_t* out_dev, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, program_dev && program_host_header && out_dev, "null argument");
    LH_ARG(ctx, program_host_header[TH_MAGIC] == TRACE_PROGRAM_MAGIC, "bad trace program header");
    LH_ARG(ctx, n_real <= height, "n_real exceeds height");
    LH_ARG(ctx, repr == LURKHIP_REPR_CANONICAL || repr == LURKHIP_REPR_MONTY, "bad repr %d", repr);
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t width = program_host_header[TH_WIDTH];
    const uint32_t max_vars = program_host_header[TH_MAX_VARS];
    LH_ARG(ctx, max_vars <= 4096, "function needs %u variables, more than the kernel's map capacity", max_vars);
    if (height == 0) return LURKHIP_OK;
    const size_t tile_words = (size_t)TBLOCK * width + (((size_t)TBLOCK * width) >> 5) + 1;
    const bool staged = tile_words * 4 <= 64 * 1024;
    if (!staged) LH_HIP(ctx, hipMemsetAsync(out_dev, 0, (size_t)height * width * sizeof(uint32_t), ctx->stream));
    TraceArgs a{program_dev, args_dev, outputs_dev, provides_dev, depths_dev, (const RowMeta*)meta_dev, stream_dev, out_dev,
                n_real, height, nonce_start, repr == LURKHIP_REPR_CANONICAL};
    dim3 grid((height + TBLOCK - 1) / TBLOCK), block(TBLOCK);
    lurkhip::span_begin(ctx, "trace_func");
    const size_t lds = staged ? tile_words * 4 : 0;
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
    lurkhip::span_end(ctx, "trace_func");
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t lurkhip_trace_mem_dev(lurkhip_ctx* ctx, uint32_t len, uint32_t n_real, uint32_t height, const uint32_t* values_dev,
                              const uint32_t* provides_dev, uint32_t* out_dev, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out_dev && (n_real == 0 || (values_dev && provides_dev)), "null argument");
    LH_ARG(ctx, n_real <= height, "n_real exceeds height");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    size_t total = (size_t)height * (4 + len);
    if (total == 0) return LURKHIP_OK;
    hipLaunchKernelGGL(k_trace_mem, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, values_dev, provides_dev, len,
                       n_real, height, out_dev, repr == LURKHIP_REPR_CANONICAL);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t lurkhip_trace_bytes_dev(lurkhip_ctx* ctx, const uint32_t* records_dev, int32_t is_real, uint32_t* out_dev, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out_dev && (!is_real || records_dev), "null argument");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    size_t total = (size_t)65536 * 13;
    hipLaunchKernelGGL(k_trace_bytes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, records_dev, is_real, out_dev,
                       repr == LURKHIP_REPR_CANONICAL);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
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
