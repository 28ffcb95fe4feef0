// Device side of the FuncChip trace generator, shared by the row interpreter (trace.hip) and the per-function compiled
// kernels (trace_jit.cpp generates their row functions; hiprtc compiles them against this header).
//
// Replaces (T2/T3/T4/T7 in SURVEY.md 8a):
//   Func/Block/Ctrl/Op::populate_row                 /root/reference/src/lair/trace.rs:145-418
//   RequireRecord / ProvideRecord population          /root/reference/src/air/builder.rs:152-214
//   u64 / depth / big-num gadget witnesses            /root/reference/src/gadgets/unsigned/{add,mul,cmp,less_than,is_zero,div_rem}.rs,
//                                                     /root/reference/src/gadgets/big_num/cmp.rs
#pragma once
#include "babybear.h"
#include "lair/trace_program.h"
#include "poseidon2_dev.h"

// the interpreter keeps the hashers out of line (three widths behind one dispatch); compiled row functions inline the one they use
#if defined(LURKHIP_COMPILED_TRACE)
#define LURK_TRACE_HASHER_INLINE __forceinline__
#else
#define LURK_TRACE_HASHER_INLINE __noinline__
#endif

namespace lurkhip_trace {

using namespace lair;

constexpr int TBLOCK = 64;
constexpr int EXTERN_MAX_IO = 48;  // inputs / returned values of one extern chip call (hasher5: 40 lanes)

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
    uint32_t out_pitch;        // words between rows of `out` (>= width; round 5: a column range of an aligned group buffer)
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

// in: the W input lanes (Montgomery); out: the whole state after the permutation (populate_witness returns it, core/poseidon.rs:71,
// and trace.rs:393-396 pushes all of it)
template <int W>
__device__ LURK_TRACE_HASHER_INLINE void extern_hasher(RowWriter& w, const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    constexpr int RP = p2::Cfg<W>::RP;
    uint32_t s[W];
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = in[i];
    RowRec<W, RP> rec{&w, w.aux0 + w.aux};
    const auto& p = p2::Cfg<W>::params();
    p2::permute_core<W>(s, RP, p.ext_rc, p.int_rc, p.diag, p.ext_rc_mp, p.int_rc_mp, p.diag_c, rec);
#pragma unroll
    for (int i = 0; i < 8; i++) w.put(rec.base + i, s[i]);
    w.aux += 8 + p2::Cfg<W>::NUM_COLS;
#pragma unroll
    for (int i = 0; i < W; i++) out[i] = s[i];
}

// the u64 of eight byte lanes (Montgomery form)
__device__ __forceinline__ uint64_t vals_u64(const uint32_t* __restrict__ in) {
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r |= (uint64_t)(bb::from_monty(in[i]) & 0xff) << (8 * i);
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

// One extern chip's witness (core/chipset.rs:28-63): `in` = its n_in input values, `out` = the values it returns (both
// Montgomery), columns pushed through `w`.  `wit` = the chip's witness size (skipped for a kind the host would have rejected).
__device__ __forceinline__ void extern_op(RowWriter& w, const uint32_t kind, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                          const uint32_t wit) {
    uint32_t no = 0;
    if (kind == CHIP_HASHER3) extern_hasher<24>(w, in, out);
    else if (kind == CHIP_HASHER4) extern_hasher<32>(w, in, out);
    else if (kind == CHIP_HASHER5) extern_hasher<40>(w, in, out);
    else if (kind == CHIP_U64_ADD || kind == CHIP_U64_SUB) {
        uint64_t x = vals_u64(in), y = vals_u64(in + 8);
        uint64_t z = kind == CHIP_U64_ADD ? x + y : x - y;
        for (int i = 0; i < 8; i++) {
            uint32_t b = (uint32_t)(z >> (8 * i)) & 0xff;
            w.push_aux_int(b);
            out[no++] = bb::to_monty(b);
        }
    } else if (kind == CHIP_U64_MUL) {
        uint64_t x = vals_u64(in), y = vals_u64(in + 8);
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
            out[no++] = bb::to_monty(res[k]);
        }
    } else if (kind == CHIP_U64_LESSTHAN) {
        // CompareWitness<_, 8>: is_comp[8], lhs_limb, rhs_limb, diff_inv, is_less_than
        uint64_t x = vals_u64(in), y = vals_u64(in + 8);
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
        out[no++] = bb::to_monty(lt);
    } else if (kind == CHIP_U64_ISZERO) {
        // IsZero<_, 8>: inverses[8] (only the first non-zero limb), result
        uint64_t x = vals_u64(in);
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
        out[no++] = bb::to_monty(z);
    } else if (kind == CHIP_U64_DIVREM) {
        // DivRem<_, 8> (unsigned/div_rem.rs:16-62): b_non_zero.inverses[8], q[8], qb { carry[8], result[8] },
        // r[8], r_lt_b { is_comp[8], lhs, rhs }, qb_cmp_a { is_comp[8], lhs, rhs, diff_inv, is_less_than }
        const uint64_t x = vals_u64(in), y = vals_u64(in + 8);
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
        for (int i = 0; i < 8; i++) out[no++] = bb::to_monty((uint32_t)(qv >> (8 * i)) & 0xff);
        for (int i = 0; i < 8; i++) out[no++] = bb::to_monty((uint32_t)(rem >> (8 * i)) & 0xff);
    } else if (kind == CHIP_BIGNUM_LESSTHAN) {
        // BigNumCompareWitness (big_num/cmp.rs:13-49): is_comp[8], lhs_limb, rhs_limb, lhs_word { is_msb_lt, bytes[4] },
        // rhs_word { .. }, CompareWitness<_, 4> { is_comp[4], lhs, rhs, diff_inv, is_less_than }
        int idx = -1;
        uint32_t lm = 0, rm = 0;
        for (int i = 7; i >= 0; i--)
            if (in[i] != in[8 + i]) {
                idx = i;
                lm = in[i];
                rm = in[8 + i];
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
        out[no++] = bb::to_monty(lt);
    } else {
        // unsupported chips are rejected on the host before launch
        w.aux += wit;
    }
    (void)no;
}

// RowMeta, hints / requires / depth requires of a row; the prologue every row shares (trace.rs:82-131) is in the callers
// One row per lane.  STAGED: the workgroup's 64 rows are built in a zero-filled LDS tile and leave as one contiguous run
// of 64 * width words with coalesced stores (a lane writing its own row straight to HBM touches 64 lines per store
// instruction: 4x write amplification measured); the output needs no memset then.
template <bool STAGED, class RowFn>
__device__ __forceinline__ void trace_kernel_body(const TraceArgs& a, RowFn&& row_fn) {
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
            row_fn(a, row_i, w);
        }
        __syncthreads();
        uint32_t* __restrict__ dst = a.out + (size_t)row0 * a.out_pitch;
        if (a.out_pitch == width) {
            for (uint32_t e = threadIdx.x; e < words; e += TBLOCK) dst[e] = tile[e + (e >> 5)];
        } else {
            // pitched rows: lanes still run along the rows (the row of word e by a float quotient, exact after one correction: e < 2^24)
            const float inv_w = 1.0f / (float)width;
            for (uint32_t e = threadIdx.x; e < words; e += TBLOCK) {
                uint32_t r = (uint32_t)((float)e * inv_w);
                r += (r + 1u) * width <= e ? 1u : 0u;
                r -= r * width > e ? 1u : 0u;
                dst[(size_t)r * a.out_pitch + (e - r * width)] = tile[e + (e >> 5)];
            }
        }
    } else {
        if (row_i >= a.height) return;
        RowWriter w{a.out + (size_t)row_i * a.out_pitch, 1 + n_in + n_out, 0, a.canonical_out != 0};
        row_fn(a, row_i, w);
    }
}

}  // namespace lurkhip_trace
