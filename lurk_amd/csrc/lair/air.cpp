// Symbolic AIR builder for the Lair chips + lowering to the device register program.  See air.h.
#include <cstdlib>
#include "air.h"

#include <algorithm>
#include <array>
#include <functional>
#include <tuple>

#include "../p2_params.h"

namespace lair {

// ------------------------------------------------------------------ builder
E Builder::intern(NodeKind k, uint32_t a, uint32_t b, uint8_t degree) {
    auto key = std::make_tuple((uint8_t)k, a, b);
    auto it = memo_.find(key);
    if (it != memo_.end()) return it->second;
    E id = (E)air_.nodes.size();
    air_.nodes.push_back(Node{k, a, b, degree});
    memo_.emplace(key, id);
    return id;
}
E Builder::leaf(NodeKind k, uint32_t a, uint8_t degree) { return intern(k, a, 0, degree); }
E Builder::cst(uint32_t canonical) { return intern(N_CONST, canonical % P, 0, 0); }

bool Builder::is_const(E e, uint32_t* v) const {
    const Node& n = air_.nodes[e];
    if (n.kind != N_CONST) return false;
    if (v) *v = n.a;
    return true;
}

E Builder::add(E a, E b) {
    uint32_t x, y;
    const bool ca = is_const(a, &x), cb = is_const(b, &y);
    if (ca && cb) return cst(fadd(x, y));
    if (ca && x == 0) return b;
    if (cb && y == 0) return a;
    if (a > b) std::swap(a, b);  // commutative: one node per unordered pair
    return intern(N_ADD, a, b, std::max(air_.nodes[a].degree, air_.nodes[b].degree));
}
E Builder::sub(E a, E b) {
    uint32_t x, y;
    const bool ca = is_const(a, &x), cb = is_const(b, &y);
    if (ca && cb) return cst(fsub(x, y));
    if (cb && y == 0) return a;
    return intern(N_SUB, a, b, std::max(air_.nodes[a].degree, air_.nodes[b].degree));
}
E Builder::mul(E a, E b) {
    uint32_t x, y;
    const bool ca = is_const(a, &x), cb = is_const(b, &y);
    if (ca && cb) return cst(fmul(x, y));
    if ((ca && x == 0) || (cb && y == 0)) return cst(0);
    if (ca && x == 1) return b;
    if (cb && y == 1) return a;
    if (a > b) std::swap(a, b);
    return intern(N_MUL, a, b, (uint8_t)(air_.nodes[a].degree + air_.nodes[b].degree));
}

void Builder::assert_zero(E x, E cond) { air_.constraints.push_back(cond == NONE ? x : mul(cond, x)); }

void Builder::receive(const std::vector<E>& values, E is_real) {
    air_.receives.push_back(Interaction{false, INTERACTION_KIND_MEMORY, values, is_real});
}
void Builder::send(const std::vector<E>& values, E is_real) {
    air_.sends.push_back(Interaction{true, INTERACTION_KIND_MEMORY, values, is_real});
}

// air/builder.rs:42-72
void Builder::provide(const std::vector<E>& relation, E last_nonce, E last_count, E is_real) {
    std::vector<E> r{last_nonce, last_count};
    r.insert(r.end(), relation.begin(), relation.end());
    receive(r, is_real);
    std::vector<E> s{zero(), zero()};
    s.insert(s.end(), relation.begin(), relation.end());
    send(s, is_real);
}

// air/builder.rs:75-104
void Builder::require(const std::vector<E>& relation, E nonce, E prev_nonce, E prev_count, E count_inv, E is_real) {
    E count = add(prev_count, one());
    assert_one(mul(count, count_inv), is_real);
    std::vector<E> r{prev_nonce, prev_count};
    r.insert(r.end(), relation.begin(), relation.end());
    receive(r, is_real);
    std::vector<E> s{nonce, count};
    s.insert(s.end(), relation.begin(), relation.end());
    send(s, is_real);
}

uint32_t ChipAir::max_constraint_degree() const {
    uint32_t d = 0;
    for (E c : constraints) d = std::max<uint32_t>(d, nodes[c].degree);
    return d;
}
uint32_t ChipAir::log_quotient_degree() const {
    uint32_t d = max_constraint_degree();
    if (!sends.empty() || !receives.empty()) d = std::max(d, 3u);
    if (d < 2) d = 2;  // log2_ceil(d - 1) with at least one quotient chunk
    uint32_t l = 0;
    while ((1u << l) < d - 1) l++;
    return l;
}
uint32_t ChipAir::permutation_width() const {
    const uint32_t batch = 1u << log_quotient_degree();
    return (num_interactions() + batch - 1) / batch + 1;
}

// ------------------------------------------------------------------ byte-lookup records (gadgets/bytes/builder.rs)
namespace {

constexpr uint32_t CALL_TAG = 0, MEMORY_TAG = 1, BYTE_TAG = 3;

struct ByteAirRecord {
    struct Rec {
        std::vector<E> relation;
        E is_real;
    };
    std::vector<Rec> records;
    Builder& b;
    explicit ByteAirRecord(Builder& bb) : b(bb) {}
    void range_check_u8_pair(E i1, E i2, E is_real) { records.push_back({{b.cst(BYTE_TAG), b.cst(1), i1, i2}, is_real}); }
    void range_check_u8_iter(const std::vector<E>& xs, E is_real) {
        for (size_t i = 0; i < xs.size(); i += 2) range_check_u8_pair(xs[i], i + 1 < xs.size() ? xs[i + 1] : b.zero(), is_real);
    }
    void range_check_u16(E i, E is_real) { records.push_back({{b.cst(BYTE_TAG), b.cst(2), i}, is_real}); }
    void less_than(E i1, E i2, E r, E is_real) { records.push_back({{b.cst(BYTE_TAG), b.cst(3), i1, i2, r}, is_real}); }
    // require_all (builder.rs:27-37): zip_eq with the row's RequireRecords
    void require_all(E nonce, const std::vector<std::array<E, 3>>& requires_) {
        if (requires_.size() != records.size()) throw ExecError("byte air record / require count mismatch");
        for (size_t i = 0; i < records.size(); i++)
            b.require(records[i].relation, nonce, requires_[i][0], requires_[i][1], requires_[i][2], records[i].is_real);
    }
};


// ------------------------------------------------------------------ extern chips (core/chipset.rs dispatch)
struct P2Consts {
    int w, rp;
    const uint32_t *diag, *ext_rc, *int_rc;
};
P2Consts p2_consts(int w) {
    for (int i = 0; i < LURK_P2_NUM_WIDTHS; i++)
        if (LURK_P2_PARAMS[i].width == w) return {w, LURK_P2_PARAMS[i].rounds_p, LURK_P2_PARAMS[i].diag, LURK_P2_PARAMS[i].ext_rc, LURK_P2_PARAMS[i].int_rc};
    throw ExecError("no Poseidon2 AIR for width " + std::to_string(w));
}

// p3 Poseidon2ExternalMatrixGeneral on expressions: M4 = circ(2,3,1,1) per 4-chunk, then add the column sums
void external_linear_layer(Builder& b, std::vector<E>& s) {
    const size_t w = s.size();
    for (size_t i = 0; i < w; i += 4) {
        E x0 = s[i], x1 = s[i + 1], x2 = s[i + 2], x3 = s[i + 3];
        E t01 = b.add(x0, x1), t23 = b.add(x2, x3), t0123 = b.add(t01, t23);
        E t01123 = b.add(t0123, x1), t01233 = b.add(t0123, x3);
        s[i + 3] = b.add(t01233, b.add(x0, x0));
        s[i + 1] = b.add(t01123, b.add(x2, x2));
        s[i] = b.add(t01123, t01);
        s[i + 2] = b.add(t01233, t23);
    }
    E sums[4];
    for (size_t k = 0; k < 4; k++) {
        E acc = s[k];
        for (size_t j = k + 4; j < w; j += 4) acc = b.add(acc, s[j]);
        sums[k] = acc;
    }
    for (size_t i = 0; i < w; i++) s[i] = b.add(s[i], sums[i & 3]);
}

// InternalDiffusion::permute_mut (poseidon/config.rs:109-118)
void internal_linear_layer(Builder& b, std::vector<E>& s, const uint32_t* diag) {
    E sum = b.zero();
    for (E x : s) sum = b.add(sum, x);
    for (size_t i = 0; i < s.size(); i++) s[i] = b.add(b.mul(s[i], b.cst(diag[i])), sum);
}

E cube(Builder& b, E x) { return b.mul(b.mul(x, x), x); }

// Poseidon2Cols::eval (poseidon/wide/air.rs:15-124); cols = external_rounds_state[8][W], external_rounds_sbox[8][W],
// internal_rounds_state_init[W], internal_rounds_state0[RP - 1], internal_rounds_sbox[RP] (wide/columns.rs:16-32)
void poseidon2_wide_eval(Builder& b, int width, const std::vector<E>& input, const std::vector<E>& output, const std::vector<E>& cols,
                         E is_real) {
    const P2Consts pc = p2_consts(width);
    const int W = pc.w, RP = pc.rp;
    auto ext_state = [&](int r, int i) { return cols[(size_t)r * W + i]; };
    auto ext_sbox = [&](int r, int i) { return cols[(size_t)8 * W + (size_t)r * W + i]; };
    auto int_init = [&](int i) { return cols[(size_t)16 * W + i]; };
    auto int_state0 = [&](int r) { return cols[(size_t)17 * W + r]; };
    auto int_sbox = [&](int r) { return cols[(size_t)17 * W + (RP - 1) + r]; };
    std::vector<E> state(W);
    for (int i = 0; i < W; i++) state[i] = b.mul(is_real, input[i]);
    external_linear_layer(b, state);
    auto external_round = [&](int round) {
        for (int i = 0; i < W; i++) {
            b.assert_eq(state[i], ext_state(round, i));
            state[i] = ext_state(round, i);
        }
        for (int i = 0; i < W; i++) state[i] = b.add(state[i], b.mul(is_real, b.cst(pc.ext_rc[round * W + i])));
        for (int i = 0; i < W; i++) {
            E s3 = ext_sbox(round, i);
            b.assert_eq(cube(b, state[i]), s3);
            state[i] = b.mul(state[i], b.mul(s3, s3));
        }
        external_linear_layer(b, state);
    };
    for (int r = 0; r < 4; r++) external_round(r);
    for (int r = 0; r < RP; r++) {
        if (r == 0) {
            for (int i = 0; i < W; i++) {
                b.assert_eq(state[i], int_init(i));
                state[i] = int_init(i);
            }
        } else {
            b.assert_eq(state[0], int_state0(r - 1));
            state[0] = int_state0(r - 1);
        }
        state[0] = b.add(state[0], b.mul(is_real, b.cst(pc.int_rc[r])));
        E s3 = int_sbox(r);
        b.assert_eq(cube(b, state[0]), s3);
        state[0] = b.mul(state[0], b.mul(s3, s3));
        internal_linear_layer(b, state, pc.diag);
    }
    for (int r = 4; r < 8; r++) external_round(r);
    for (size_t i = 0; i < output.size() && i < state.size(); i++) b.assert_eq(state[i], b.mul(is_real, output[i]));
}

// AddWitness::assert_add (gadgets/unsigned/add.rs:16-58): lhs + rhs = out limb-wise with boolean carries
void assert_add(Builder& b, const std::vector<E>& lhs, const std::vector<E>& rhs, const std::vector<E>& out, E is_real) {
    const E base_inv = b.cst(finv(256));
    E carry = b.zero();
    for (size_t i = 0; i < out.size(); i++) {
        E sum = b.add(b.add(lhs[i], rhs[i]), carry);
        carry = b.mul(b.sub(sum, out[i]), base_inv);
        b.assert_bool(carry, is_real);
    }
}

// ------------------------------------------------------------------ Func AIR (lair/air.rs:158-552)
struct Val {
    bool is_const;
    uint32_t c;
    E e;
};

struct FuncAirWalk {
    const Toplevel& t;
    const Func& f;
    Builder& b;
    LayoutSizes ls;
    uint32_t in_cur = 0, aux_cur = 0, out_cur = 0;  // ColumnIndex
    std::vector<Val> map;
    E nonce;
    std::vector<E> depth;  // own depth bytes (partial functions)

    FuncAirWalk(const Toplevel& tl, const Func& fn, Builder& bb) : t(tl), f(fn), b(bb), ls(compute_layout_sizes(tl, fn)) {}

    uint32_t col_input(uint32_t i) const { return 1 + i; }
    uint32_t col_output(uint32_t i) const { return 1 + ls.input + i; }
    uint32_t col_aux(uint32_t i) const { return 1 + ls.input + ls.output + i; }
    uint32_t col_sel(uint32_t i) const { return 1 + ls.input + ls.output + ls.aux + i; }

    E next_aux() {
        if (aux_cur >= ls.aux) throw ExecError("AIR walk ran past the aux columns of " + f.name);
        return b.main(col_aux(aux_cur++));
    }
    std::array<E, 3> next_require() {
        E pn = next_aux(), pc = next_aux(), ci = next_aux();
        return {pn, pc, ci};
    }
    E val_expr(const Val& v) { return v.is_const ? b.cst(v.c) : v.e; }
    E var(uint32_t i) { return val_expr(map.at(i)); }
    void push_expr(E e) { map.push_back(Val{false, 0, e}); }

    E return_sel(const Block& blk) {
        E s = b.zero();
        for (uint32_t i : blk.return_idents) s = b.add(s, b.main(col_sel(i)));
        return s;
    }

    // provenance.rs DepthLessThan = LessThanWitness<_, 4>; unsigned/less_than.rs:44-99
    void assert_less_than(const std::array<E, 6>& wit, const std::vector<E>& lhs, const std::vector<E>& rhs, ByteAirRecord& rec,
                          E is_real) {
        const int W = DEPTH_W;
        E is_equal = b.zero();
        for (int i = 0; i < W; i++) {
            if (i > 0) b.assert_eq(lhs[i], rhs[i], b.both(is_real, is_equal));
            E is_comp = wit[i];
            b.assert_bool(is_comp, is_real);
            is_equal = b.add(is_equal, is_comp);
        }
        b.assert_one(is_equal, is_real);
        auto select_limb = [&](const std::vector<E>& w) {
            E s = b.zero();
            for (int i = 0; i < W; i++) s = b.add(s, b.mul(w[i], wit[i]));
            return s;
        };
        b.assert_eq(select_limb(lhs), wit[4], is_real);
        b.assert_eq(select_limb(rhs), wit[5], is_real);
        rec.less_than(wit[4], wit[5], b.one(), is_real);
    }

    // air.rs:103-133
    void eval_depth(E sel, std::vector<E>& out) {
        std::vector<E> dep_depth;
        for (int i = 0; i < DEPTH_W; i++) dep_depth.push_back(next_aux());
        std::array<E, 6> wit;
        for (int i = 0; i < DEPTH_LESS_THAN_SIZE; i++) wit[i] = next_aux();
        ByteAirRecord rec(b);
        assert_less_than(wit, dep_depth, depth, rec, sel);
        std::vector<std::array<E, 3>> reqs;
        for (int i = 0; i < DEPTH_LT_REQUIRES; i++) reqs.push_back(next_require());
        rec.require_all(nonce, reqs);
        out.insert(out.end(), dep_depth.begin(), dep_depth.end());
    }

    // LurkChip::eval (core/chipset.rs:122-171) -> PoseidonChipset::eval (core/poseidon.rs:74-93), U64::eval (core/u64.rs:173-229)
    std::vector<E> eval_chip(const Chip& chip, E is_real, const std::vector<E>& ins, const std::vector<E>& wit,
                             const std::vector<std::array<E, 3>>& reqs) {
        ByteAirRecord rec(b);
        std::vector<E> out;
        auto word = [&](size_t from) { return std::vector<E>(ins.begin() + from, ins.begin() + from + 8); };
        switch (chip.kind) {
            case CHIP_HASHER3:
            case CHIP_HASHER4:
            case CHIP_HASHER5: {
                std::vector<E> output(wit.begin(), wit.begin() + 8), cols(wit.begin() + 8, wit.end());
                poseidon2_wide_eval(b, (int)chip.input_size, ins, output, cols, is_real);
                out = output;
                break;
            }
            case CHIP_U64_ADD:
            case CHIP_U64_SUB: {
                // Sum / Diff (gadgets/unsigned/add.rs:80-98,135-154): result bytes range-checked, then assert_add
                std::vector<E> result(wit.begin(), wit.begin() + 8);
                rec.range_check_u8_iter(result, is_real);
                if (chip.kind == CHIP_U64_ADD) assert_add(b, word(0), word(8), result, is_real);
                else assert_add(b, result, word(8), word(0), is_real);
                out = result;
                break;
            }
            case CHIP_U64_MUL: {
                // Product { MulWitness { carry[8] }, result[8] } (gadgets/unsigned/mul.rs:66-108,141-164)
                std::vector<E> carry(wit.begin(), wit.begin() + 8), result(wit.begin() + 8, wit.begin() + 16);
                const std::vector<E> lhs = word(0), rhs = word(8);
                std::vector<E> products(8, b.zero());
                for (int i = 0; i < 8; i++)
                    for (int j = 0; j + i < 8; j++) products[i + j] = b.add(products[i + j], b.mul(lhs[i], rhs[j]));
                E carry_prev = b.zero();
                for (int k = 0; k < 8; k++) {
                    rec.range_check_u16(carry[k], is_real);
                    E o = b.add(result[k], b.mul(carry[k], b.cst(256)));
                    b.assert_eq(b.add(products[k], carry_prev), o, is_real);
                    carry_prev = carry[k];
                }
                rec.range_check_u8_iter(result, is_real);
                out = result;
                break;
            }
            case CHIP_U64_LESSTHAN: {
                // CompareWitness<_, 8> { is_comp[8], lhs_comp_limb, rhs_comp_limb, comp_diff_inv, is_less_than } (cmp.rs:48-118)
                const std::vector<E> lhs = word(0), rhs = word(8);
                E is_equal = b.one();
                for (int i = 7; i >= 0; i--) {
                    b.assert_bool(wit[i], is_real);
                    is_equal = b.sub(is_equal, wit[i]);
                    b.assert_eq(lhs[i], rhs[i], b.both(is_real, is_equal));
                }
                b.assert_bool(is_equal, is_real);
                auto select_limb = [&](const std::vector<E>& w) {
                    E s = b.zero();
                    for (int i = 0; i < 8; i++) s = b.add(s, b.mul(w[i], wit[i]));
                    return s;
                };
                b.assert_eq(select_limb(lhs), wit[8], is_real);
                b.assert_eq(select_limb(rhs), wit[9], is_real);
                E is_different = b.sub(b.one(), is_equal);
                E comp_diff = b.sub(wit[8], wit[9]);
                b.assert_eq(b.mul(comp_diff, wit[10]), is_different, is_real);
                rec.less_than(wit[8], wit[9], wit[11], is_real);
                out = {wit[11]};
                break;
            }
            case CHIP_U64_ISZERO: {
                // IsZeroOrEqual<_, 8> { IsZeroWitness { inverses[8] }, result } (is_zero.rs:69-92,142-157)
                const E is_zero = wit[8];
                b.assert_bool(is_zero, is_real);
                E lc = b.zero();
                for (int i = 0; i < 8; i++) {
                    b.assert_zero(ins[i], b.both(is_real, is_zero));
                    lc = b.add(lc, b.mul(ins[i], wit[i]));
                }
                b.assert_eq(lc, b.sub(b.one(), is_zero), is_real);
                out = {is_zero};
                break;
            }
            case CHIP_U64_DIVREM: {
                // DivRem::eval (gadgets/unsigned/div_rem.rs:65-110)
                const std::vector<E> a = word(0), bw = word(8);
                std::vector<E> inverses(wit.begin(), wit.begin() + 8), qv(wit.begin() + 8, wit.begin() + 16),
                    carry(wit.begin() + 16, wit.begin() + 24), qb(wit.begin() + 24, wit.begin() + 32), r(wit.begin() + 32, wit.begin() + 40);
                // b != 0: 1 = sum w_i * b_i (is_zero.rs:48-65)
                {
                    E lc = b.zero();
                    for (int i = 0; i < 8; i++) lc = b.add(lc, b.mul(bw[i], inverses[i]));
                    b.assert_one(lc, is_real);
                }
                rec.range_check_u8_iter(qv, is_real);
                {  // qb = q * b (Product::eval)
                    std::vector<E> products(8, b.zero());
                    for (int i = 0; i < 8; i++)
                        for (int j = 0; j + i < 8; j++) products[i + j] = b.add(products[i + j], b.mul(qv[i], bw[j]));
                    E carry_prev = b.zero();
                    for (int k = 0; k < 8; k++) {
                        rec.range_check_u16(carry[k], is_real);
                        b.assert_eq(b.add(products[k], carry_prev), b.add(qb[k], b.mul(carry[k], b.cst(256))), is_real);
                        carry_prev = carry[k];
                    }
                    rec.range_check_u8_iter(qb, is_real);
                }
                // r = a - qb (Diff::eval): r + qb = a
                rec.range_check_u8_iter(r, is_real);
                assert_add(b, r, qb, a, is_real);
                // r < b (LessThanWitness<_, 8>::assert_less_than, less_than.rs:44-99)
                {
                    const E* lw = &wit[40];
                    E is_equal = b.zero();
                    for (int i = 0; i < 8; i++) {
                        if (i > 0) b.assert_eq(r[i], bw[i], b.both(is_real, is_equal));
                        b.assert_bool(lw[i], is_real);
                        is_equal = b.add(is_equal, lw[i]);
                    }
                    b.assert_one(is_equal, is_real);
                    E sl = b.zero(), sr = b.zero();
                    for (int i = 0; i < 8; i++) {
                        sl = b.add(sl, b.mul(r[i], lw[i]));
                        sr = b.add(sr, b.mul(bw[i], lw[i]));
                    }
                    b.assert_eq(sl, lw[8], is_real);
                    b.assert_eq(sr, lw[9], is_real);
                    rec.less_than(lw[8], lw[9], b.one(), is_real);
                }
                // qb <= a (CompareWitness<_, 8>::eval)
                {
                    const E* cw = &wit[50];
                    E is_equal = b.one();
                    for (int i = 7; i >= 0; i--) {
                        b.assert_bool(cw[i], is_real);
                        is_equal = b.sub(is_equal, cw[i]);
                        b.assert_eq(qb[i], a[i], b.both(is_real, is_equal));
                    }
                    b.assert_bool(is_equal, is_real);
                    E sl = b.zero(), sr = b.zero();
                    for (int i = 0; i < 8; i++) {
                        sl = b.add(sl, b.mul(qb[i], cw[i]));
                        sr = b.add(sr, b.mul(a[i], cw[i]));
                    }
                    b.assert_eq(sl, cw[8], is_real);
                    b.assert_eq(sr, cw[9], is_real);
                    b.assert_eq(b.mul(b.sub(cw[8], cw[9]), cw[10]), b.sub(b.one(), is_equal), is_real);
                    rec.less_than(cw[8], cw[9], cw[11], is_real);
                    b.assert_one(b.add(cw[11], is_equal), is_real);  // is_less_than_or_equal
                }
                out = qv;
                out.insert(out.end(), r.begin(), r.end());
                break;
            }
            case CHIP_BIGNUM_LESSTHAN: {
                // BigNumCompareWitness::eval (gadgets/big_num/cmp.rs:52-135)
                const std::vector<E> lhs(ins.begin(), ins.begin() + 8), rhs(ins.begin() + 8, ins.begin() + 16);
                E is_equal = b.one();
                for (int i = 7; i >= 0; i--) {
                    b.assert_bool(wit[i], is_real);
                    is_equal = b.sub(is_equal, wit[i]);
                    b.assert_eq(lhs[i], rhs[i], b.both(is_real, is_equal));
                }
                b.assert_bool(is_equal, is_real);
                E sl = b.zero(), sr = b.zero();
                for (int i = 0; i < 8; i++) {
                    sl = b.add(sl, b.mul(lhs[i], wit[i]));
                    sr = b.add(sr, b.mul(rhs[i], wit[i]));
                }
                b.assert_eq(sl, wit[8], is_real);
                b.assert_eq(sr, wit[9], is_real);
                // FieldToWord32::eval (gadgets/unsigned/field.rs:34-84,120-139): witness { is_msb_less_than, bytes[4] }
                auto field_to_word = [&](E field, const E* fw) {
                    const E is_msb_lt = fw[0];
                    const E* wd = fw + 1;
                    b.assert_bool(is_msb_lt, is_real);
                    E recomposed = b.zero();
                    for (int i = 3; i >= 0; i--) recomposed = b.add(b.mul(recomposed, b.cst(256)), wd[i]);
                    b.assert_eq(field, recomposed, is_real);
                    rec.less_than(wd[3], b.cst(0x78), is_msb_lt, is_real);
                    const E when_eq = b.mul(is_real, b.sub(b.one(), is_msb_lt));
                    b.assert_eq(wd[3], b.cst(0x78), when_eq);
                    for (int i = 0; i < 3; i++) b.assert_eq(wd[i], b.zero(), when_eq);
                    rec.range_check_u8_iter({wd[0], wd[1], wd[2], wd[3]}, is_real);
                };
                field_to_word(wit[8], &wit[10]);
                field_to_word(wit[9], &wit[15]);
                // CompareWitness<_, 4> on the two words
                const E* lwd = &wit[11];
                const E* rwd = &wit[16];
                const E* cw = &wit[20];
                E w_equal = b.one();
                for (int i = 3; i >= 0; i--) {
                    b.assert_bool(cw[i], is_real);
                    w_equal = b.sub(w_equal, cw[i]);
                    b.assert_eq(lwd[i], rwd[i], b.both(is_real, w_equal));
                }
                b.assert_bool(w_equal, is_real);
                E wl = b.zero(), wr = b.zero();
                for (int i = 0; i < 4; i++) {
                    wl = b.add(wl, b.mul(lwd[i], cw[i]));
                    wr = b.add(wr, b.mul(rwd[i], cw[i]));
                }
                b.assert_eq(wl, cw[4], is_real);
                b.assert_eq(wr, cw[5], is_real);
                b.assert_eq(b.mul(b.sub(cw[4], cw[5]), cw[6]), b.sub(b.one(), w_equal), is_real);
                rec.less_than(cw[4], cw[5], cw[7], is_real);
                b.assert_eq(is_equal, w_equal, is_real);
                out = {cw[7]};
                break;
            }
            default:
                throw ExecError("AIR of extern chip " + chip.name + " is not available in this build");
        }
        rec.require_all(nonce, reqs);
        return out;
    }

    void run() {
        nonce = b.main(0);
        E next_nonce = b.main_next(0);
        // nonces are unique, even for dummy rows
        b.assert_eq(next_nonce, b.add(nonce, b.one()), b.is_transition());
        std::vector<E> call_inp;
        for (uint32_t i = 0; i < f.input_size; i++) {
            E v = b.main(col_input(in_cur++));
            push_expr(v);
            call_inp.push_back(v);
        }
        E toplevel_sel = return_sel(f.body);
        b.assert_bool(toplevel_sel);
        E last_nonce = next_aux(), last_count = next_aux();
        std::vector<E> out;
        for (uint32_t i = 0; i < f.output_size; i++) out.push_back(b.main(col_output(i)));
        if (f.partial) {
            for (int i = 0; i < DEPTH_W; i++) depth.push_back(next_aux());
            const int num_requires = (DEPTH_W / 2) + (DEPTH_W % 2);
            std::vector<std::array<E, 3>> reqs;
            for (int i = 0; i < num_requires; i++) reqs.push_back(next_require());
            ByteAirRecord rec(b);
            rec.range_check_u8_iter(depth, toplevel_sel);
            rec.require_all(nonce, reqs);
            out.insert(out.end(), depth.begin(), depth.end());
        }
        std::vector<E> rel{b.cst(CALL_TAG), b.cst(f.index)};
        rel.insert(rel.end(), call_inp.begin(), call_inp.end());
        rel.insert(rel.end(), out.begin(), out.end());
        b.provide(rel, last_nonce, last_count, toplevel_sel);
        eval_block(f.body, toplevel_sel);
    }

    void eval_block(const Block& blk, E sel) {
        for (const Op& op : blk.ops) eval_op(op, sel);
        eval_ctrl(blk.ctrl);
    }

    void eval_ctrl(const Ctrl& c) {
        if (c.kind == Ctrl::Return) {
            E sel = b.main(col_sel(c.ident));
            for (uint32_t v : c.ret) {
                if (out_cur >= ls.output) throw ExecError("AIR walk ran past the output columns");
                E out_var = b.main(col_output(out_cur++));
                b.assert_eq(var(v), out_var, sel);
            }
            return;
        }
        const size_t map_len = map.size();
        const uint32_t s_aux = aux_cur, s_out = out_cur;
        auto process = [&](const Block& blk) {
            E sel = return_sel(blk);
            eval_block(blk, sel);
            map.resize(map_len);
            aux_cur = s_aux;
            out_cur = s_out;
        };
        if (c.kind == Ctrl::Choose) {
            for (const auto& blk : c.unique_branches) process(*blk);
        } else {
            for (const auto& kv : c.branches) process(*kv.second);
        }
        if (c.def) process(*c.def);
    }

    void eval_op(const Op& op, E sel) {
        switch (op.kind) {
            case OpKind::AssertNe: {
                // constrain_inequality_witness (air.rs:540-552)
                std::vector<E> coeffs;
                for (size_t i = 0; i < op.a.size(); i++) coeffs.push_back(next_aux());
                E acc = b.zero();
                for (size_t i = 0; i < op.a.size(); i++) acc = b.add(acc, b.mul(coeffs[i], b.sub(var(op.a[i]), var(op.b[i]))));
                b.assert_one(acc, sel);
                break;
            }
            case OpKind::AssertEq:
                for (size_t i = 0; i < op.a.size(); i++) b.assert_eq(var(op.a[i]), var(op.b[i]), sel);
                break;
            case OpKind::Contains: {
                E y = var(op.y);
                E acc = b.sub(var(op.a[0]), y);
                for (size_t i = 1; i < op.a.size(); i++) {
                    E diff = b.sub(var(op.a[i]), y);
                    E aux = next_aux();
                    b.assert_eq(b.mul(acc, diff), aux, sel);
                    acc = aux;
                }
                b.assert_zero(acc, sel);
                break;
            }
            case OpKind::Const:
                map.push_back(Val{true, op.c % P, 0});
                break;
            case OpKind::Add:
            case OpKind::Sub: {
                const Val &x = map.at(op.x), &y = map.at(op.y);
                const bool add = op.kind == OpKind::Add;
                if (x.is_const && y.is_const) map.push_back(Val{true, add ? fadd(x.c, y.c) : fsub(x.c, y.c), 0});
                else {
                    E ex = val_expr(x), ey = val_expr(y);
                    push_expr(add ? b.add(ex, ey) : b.sub(ex, ey));
                }
                break;
            }
            case OpKind::Mul: {
                const Val x = map.at(op.x), y = map.at(op.y);
                if (x.is_const && y.is_const) map.push_back(Val{true, fmul(x.c, y.c), 0});
                else {
                    // air.rs:345-358: an aux column whenever not both operands are constants (the layout
                    // only allocates one when both have degree 1, func_chip.rs:202-211 -- upstream disagreement
                    // preserved)
                    E c = next_aux();
                    b.assert_eq(b.mul(val_expr(x), val_expr(y)), c, sel);
                    push_expr(c);
                }
                break;
            }
            case OpKind::Inv: {
                const Val x = map.at(op.x);
                if (x.is_const) map.push_back(Val{true, finv(x.c), 0});
                else {
                    E c = next_aux();
                    b.assert_one(b.mul(val_expr(x), c), sel);
                    push_expr(c);
                }
                break;
            }
            case OpKind::Not: {
                const Val x = map.at(op.x);
                if (x.is_const) map.push_back(Val{true, x.c == 0 ? 1u : 0u, 0});
                else {
                    E d = next_aux(), r = next_aux();
                    E a = val_expr(x);
                    b.assert_zero(b.mul(a, r), sel);
                    b.assert_one(b.add(b.mul(a, d), r), sel);
                    push_expr(r);
                }
                break;
            }
            case OpKind::Call: {
                const Func& callee = t.funcs.at(op.x);
                std::vector<E> out;
                for (uint32_t i = 0; i < callee.output_size; i++) {
                    E o = next_aux();
                    push_expr(o);
                    out.push_back(o);
                }
                std::vector<E> inp;
                for (uint32_t v : op.a) inp.push_back(var(v));
                auto rec = next_require();
                if (callee.partial) eval_depth(sel, out);
                std::vector<E> rel{b.cst(CALL_TAG), b.cst(op.x)};
                rel.insert(rel.end(), inp.begin(), inp.end());
                rel.insert(rel.end(), out.begin(), out.end());
                b.require(rel, nonce, rec[0], rec[1], rec[2], sel);
                break;
            }
            case OpKind::PreImg: {
                const Func& callee = t.funcs.at(op.x);
                std::vector<E> inp;
                for (uint32_t i = 0; i < callee.input_size; i++) {
                    E v = next_aux();
                    push_expr(v);
                    inp.push_back(v);
                }
                std::vector<E> out;
                for (uint32_t v : op.a) out.push_back(var(v));
                auto rec = next_require();
                if (callee.partial) eval_depth(sel, out);
                std::vector<E> rel{b.cst(CALL_TAG), b.cst(op.x)};
                rel.insert(rel.end(), inp.begin(), inp.end());
                rel.insert(rel.end(), out.begin(), out.end());
                b.require(rel, nonce, rec[0], rec[1], rec[2], sel);
                break;
            }
            case OpKind::Store: {
                E ptr = next_aux();
                push_expr(ptr);
                std::vector<E> rel{b.cst(MEMORY_TAG), ptr};
                for (uint32_t v : op.a) rel.push_back(var(v));
                auto rec = next_require();
                b.require(rel, nonce, rec[0], rec[1], rec[2], sel);
                break;
            }
            case OpKind::Load: {
                E ptr = var(op.y);
                std::vector<E> rel{b.cst(MEMORY_TAG), ptr};
                for (uint32_t i = 0; i < op.x; i++) {
                    E o = next_aux();
                    push_expr(o);
                    rel.push_back(o);
                }
                auto rec = next_require();
                b.require(rel, nonce, rec[0], rec[1], rec[2], sel);
                break;
            }
            case OpKind::RangeU8: {
                const size_t num_requires = (op.a.size() / 2) + (op.a.size() % 2);
                std::vector<std::array<E, 3>> reqs;
                for (size_t i = 0; i < num_requires; i++) reqs.push_back(next_require());
                ByteAirRecord rec(b);
                std::vector<E> xs;
                for (uint32_t v : op.a) xs.push_back(var(v));
                rec.range_check_u8_iter(xs, sel);
                rec.require_all(nonce, reqs);
                break;
            }
            case OpKind::ExternCall: {
                // air.rs:453-472: witness columns, then the chip's requires, then Chipset::eval
                const Chip& chip = t.chips.at(op.x);
                std::vector<E> input;
                for (uint32_t v : op.a) input.push_back(var(v));
                std::vector<E> wit;
                for (uint32_t i = 0; i < chip.witness_size; i++) wit.push_back(next_aux());
                std::vector<std::array<E, 3>> reqs;
                for (uint32_t i = 0; i < chip.require_size; i++) reqs.push_back(next_require());
                for (E o : eval_chip(chip, sel, input, wit, reqs)) push_expr(o);
                break;
            }
            case OpKind::Emit:
            case OpKind::Breakpoint:
            case OpKind::Debug:
                break;
        }
    }
};

}  // namespace

ChipAir build_func_air(const Toplevel& t, const Func& f) {
    ChipAir air;
    air.name = "Func[" + f.name + "]";
    Builder b(air);
    FuncAirWalk w(t, f, b);
    air.width = w.ls.total();
    w.run();
    return air;
}

// lair/memory.rs:71-109
ChipAir build_mem_air(uint32_t len) {
    ChipAir air;
    air.name = "Mem[" + std::to_string(len) + "-wide]";
    air.width = 4 + len;
    Builder b(air);
    E is_real = b.main(0), ptr_local = b.main(1), last_nonce = b.main(2), last_count = b.main(3);
    E is_real_next = b.main_next(0), ptr_next = b.main_next(1);
    b.assert_bool(is_real);
    E is_real_transition = b.mul(is_real_next, b.is_transition());
    b.assert_one(is_real, is_real_transition);
    b.assert_one(ptr_local, b.both(b.is_first_row(), is_real));
    b.assert_eq(b.add(ptr_local, b.one()), ptr_next, is_real_transition);
    std::vector<E> rel{b.cst(MEMORY_TAG), ptr_local};
    for (uint32_t i = 0; i < len; i++) rel.push_back(b.main(4 + i));
    b.provide(rel, last_nonce, last_count, is_real);
    return air;
}

// gadgets/bytes/trace.rs:117-143; columns: preprocessed [i1, i2, less_than, and, xor, or],
// main [is_real, 6 x (last_nonce, last_count)]
ChipAir build_bytes_air() {
    ChipAir air;
    air.name = "CPU";  // the name sphinx requires of the byte chip (lair_chip.rs:90-92)
    air.width = 13;
    air.prep_width = 6;
    Builder b(air);
    E is_real = b.main(0);
    b.assert_bool(is_real);
    E i1 = b.prep(0), i2 = b.prep(1);
    E input_u16 = b.add(i1, b.mul(i2, b.cst(256)));
    std::vector<std::vector<E>> relations = {
        {b.cst(BYTE_TAG), b.cst(1), i1, i2},        {b.cst(BYTE_TAG), b.cst(2), input_u16},
        {b.cst(BYTE_TAG), b.cst(3), i1, i2, b.prep(2)}, {b.cst(BYTE_TAG), b.cst(4), i1, i2, b.prep(3)},
        {b.cst(BYTE_TAG), b.cst(5), i1, i2, b.prep(4)}, {b.cst(BYTE_TAG), b.cst(6), i1, i2, b.prep(5)},
    };
    for (size_t k = 0; k < relations.size(); k++) b.provide(relations[k], b.main(1 + 2 * (uint32_t)k), b.main(2 + 2 * (uint32_t)k), is_real);
    return air;
}

// lair/lair_chip.rs:166-191
ChipAir build_entrypoint_air(uint32_t func_idx, uint32_t num_public_values) {
    ChipAir air;
    air.name = "Entrypoint[" + std::to_string(func_idx) + "]";
    air.width = num_public_values;
    air.num_public = num_public_values;
    Builder b(air);
    std::vector<E> rel{b.cst(CALL_TAG), b.cst(func_idx)};
    for (uint32_t i = 0; i < num_public_values; i++) {
        b.assert_eq(b.main(i), b.pub(i));
        rel.push_back(b.main(i));
    }
    b.require(rel, b.zero(), b.zero(), b.zero(), b.one(), b.one());
    return air;
}

// Poseidon2Chip::eval, one row per round (poseidon/air.rs:21-165); columns input[W] | is_init | rounds[R] | add_rc[W] |
// sbox_deg_3[W] | sbox_deg_7[W] | output[W] (poseidon/columns.rs:16-25); round constants in the order of
// poseidon/config.rs:59-72.  No lookups.
ChipAir build_poseidon2_air(uint32_t width) {
    const P2Consts pc = p2_consts((int)width);
    const uint32_t W = width, RP = (uint32_t)pc.rp, R = 8 + RP;
    ChipAir air;
    air.name = "Poseidon2[" + std::to_string(width) + "]";
    air.width = 5 * W + 1 + R;
    Builder b(air);
    const uint32_t o_init = W, o_rounds = W + 1, o_rc = W + 1 + R, o_s3 = o_rc + W, o_s7 = o_s3 + W, o_out = o_s7 + W;
    auto flag_sum = [&](uint32_t from, uint32_t to) {
        E acc = b.zero();
        for (uint32_t r = from; r < to; r++) acc = b.add(acc, b.main(o_rounds + r));
        return acc;
    };
    const E is_init = b.main(o_init);
    const E is_external_first = flag_sum(0, 4), is_internal = flag_sum(4, 4 + RP), is_external_second = flag_sum(4 + RP, R);
    const E is_external = b.add(is_external_first, is_external_second);
    const E is_linear = b.add(is_init, is_external);
    b.assert_bool(is_init);
    for (uint32_t r = 0; r < R; r++) b.assert_bool(b.main(o_rounds + r));
    const E is_real = b.add(b.add(is_init, is_internal), is_external);
    b.assert_bool(is_real);
    std::vector<E> add_rc(W);
    for (uint32_t i = 0; i < W; i++) add_rc[i] = b.main(i);
    for (uint32_t r = 0; r < R; r++) {
        const E flag = b.main(o_rounds + r);
        if (r >= 4 && r < 4 + RP) {
            add_rc[0] = b.add(add_rc[0], b.mul(flag, b.cst(pc.int_rc[r - 4])));
        } else {
            const uint32_t* rc = pc.ext_rc + (size_t)(r < 4 ? r : r - RP) * W;
            for (uint32_t i = 0; i < W; i++) add_rc[i] = b.add(add_rc[i], b.mul(flag, b.cst(rc[i])));
        }
    }
    for (uint32_t i = 0; i < W; i++) b.assert_eq(add_rc[i], b.main(o_rc + i), is_real);
    for (uint32_t i = 0; i < W; i++) {
        const E x = b.main(o_rc + i), s3 = b.main(o_s3 + i), s7 = b.main(o_s7 + i);
        b.assert_eq(b.mul(b.mul(x, x), x), s3);
        b.assert_eq(b.mul(b.mul(s3, s3), x), s7);
    }
    std::vector<E> sbox_result(W);
    for (uint32_t i = 0; i < W; i++) {
        const E x = b.main(o_rc + i), s7 = b.main(o_s7 + i);
        sbox_result[i] = i == 0 ? b.add(b.mul(is_init, x), b.mul(b.add(is_internal, is_external), s7))
                                : b.add(b.mul(b.add(is_init, is_internal), x), b.mul(is_external, s7));
    }
    {
        std::vector<E> state = sbox_result;
        external_linear_layer(b, state);
        for (uint32_t i = 0; i < W; i++) b.assert_eq(state[i], b.main(o_out + i), is_linear);
    }
    {
        std::vector<E> state = sbox_result;
        internal_linear_layer(b, state, pc.diag);
        for (uint32_t i = 0; i < W; i++) b.assert_eq(state[i], b.main(o_out + i), is_internal);
    }
    const E is_not_last_round = b.sub(is_real, b.main(o_rounds + R - 1));
    for (uint32_t i = 0; i < W; i++) b.assert_eq(b.main(o_out + i), b.main_next(i), is_not_last_round);
    return air;
}

// ------------------------------------------------------------------ lowering
namespace {

struct Lowerer {
    const ChipAir& air;
    std::vector<uint32_t> code;     // 2 words per instruction
    std::vector<uint32_t> consts;   // Montgomery
    std::map<uint32_t, uint32_t> const_index;
    // scheduling
    std::vector<E> order;                  // non-leaf nodes in emission order
    std::vector<int> pos;                  // node -> index in `order` (-1: not emitted)
    std::vector<int> last_use;             // node -> index of the last consuming step
    std::vector<int> reg_of;
    uint32_t n_regs = 0;

    explicit Lowerer(const ChipAir& a) : air(a), pos(a.nodes.size(), -1), last_use(a.nodes.size(), -1), reg_of(a.nodes.size(), -1) {}

    bool is_leaf(E e) const {
        NodeKind k = air.nodes[e].kind;
        return k != N_ADD && k != N_SUB && k != N_MUL;
    }
    static uint32_t to_monty(uint32_t x) { return (uint32_t)((((uint64_t)x) << 32) % P); }

    uint32_t const_slot(uint32_t canonical) {
        auto it = const_index.find(canonical);
        if (it != const_index.end()) return it->second;
        uint32_t i = (uint32_t)consts.size();
        consts.push_back(to_monty(canonical));
        const_index.emplace(canonical, i);
        return i;
    }

    uint32_t operand_of(E e) {
        const Node& n = air.nodes[e];
        uint32_t idx = n.a, type;
        switch (n.kind) {
            case N_CONST: type = airp::S_CONST; idx = const_slot(n.a); break;
            case N_MAIN: type = airp::S_MAIN; break;
            case N_MAIN_NEXT: type = airp::S_MAIN_NEXT; break;
            case N_PREP: type = airp::S_PREP; break;
            case N_PREP_NEXT: type = airp::S_PREP_NEXT; break;
            case N_PUBLIC: type = airp::S_PUBLIC; break;
            case N_IS_FIRST: type = airp::S_SEL; idx = 0; break;
            case N_IS_LAST: type = airp::S_SEL; idx = 1; break;
            case N_IS_TRANS: type = airp::S_SEL; idx = 2; break;
            default: type = airp::S_REG; idx = (uint32_t)reg_of[e]; break;
        }
        if (idx > airp::SRC_MASK) throw ExecError("AIR program operand index out of range");
        return airp::operand(type, idx);
    }

    // iterative post-order scheduling of the non-leaf nodes under `root`
    void schedule(E root) {
        if (is_leaf(root) || pos[root] >= 0) return;
        std::vector<std::pair<E, int>> st{{root, 0}};
        while (!st.empty()) {
            auto& [e, phase] = st.back();
            const Node& n = air.nodes[e];
            if (phase == 0) {
                phase = 1;
                if (!is_leaf(n.a) && pos[n.a] < 0) st.push_back({n.a, 0});
            } else if (phase == 1) {
                phase = 2;
                if (!is_leaf(n.b) && pos[n.b] < 0) st.push_back({n.b, 0});
            } else {
                if (pos[e] < 0) {
                    pos[e] = (int)order.size();
                    order.push_back(e);
                }
                st.pop_back();
            }
        }
    }

    void emit(uint32_t op, uint32_t dst, uint32_t a, uint32_t b) {
        code.push_back(op | (dst << 8));
        code.push_back(a | (b << 16));
    }

    static bool immediate_only(uint32_t op) { return op == airp::OP_IBEGIN || op == airp::OP_IVALS; }

    // `steps`: the root uses in program order; each step is (kind, node or immediate)
    struct Step {
        uint32_t op;
        E node;            // ASSERT / IVAL / IVALT / IEND operand
        uint32_t dst = 0, a = 0, b = 0;  // IBEGIN / IVALS immediates, IVALT position
    };

    std::vector<uint32_t> lower(const std::vector<Step>& steps, uint32_t n_asserts, uint32_t n_interactions, uint32_t n_sends) {
        // 1. schedule every node needed by a step right before the step's position: emission order is
        //    the concatenation over steps of the not-yet-emitted nodes
        std::vector<size_t> step_after(steps.size());  // number of compute nodes emitted before step i
        for (size_t i = 0; i < steps.size(); i++) {
            if (!immediate_only(steps[i].op)) schedule(steps[i].node);
            step_after[i] = order.size();
        }
        // 2. last use, in a merged timeline: time of compute node k = 2 * ... simpler: walk the final stream
        //    and record for each node the last stream position that reads it
        struct Item { bool is_step; size_t idx; };
        std::vector<Item> stream;
        {
            size_t k = 0;
            for (size_t i = 0; i < steps.size(); i++) {
                while (k < step_after[i]) stream.push_back({false, k++});
                stream.push_back({true, i});
            }
        }
        for (size_t s = 0; s < stream.size(); s++) {
            if (stream[s].is_step) {
                const Step& st = steps[stream[s].idx];
                if (!immediate_only(st.op) && !is_leaf(st.node)) last_use[st.node] = (int)s;
            } else {
                const Node& n = air.nodes[order[stream[s].idx]];
                if (!is_leaf(n.a)) last_use[n.a] = (int)s;
                if (!is_leaf(n.b)) last_use[n.b] = (int)s;
            }
        }
        // 3. emit with a free list
        std::vector<uint32_t> free_regs;
        auto alloc = [&]() {
            if (!free_regs.empty()) {
                uint32_t r = free_regs.back();
                free_regs.pop_back();
                return r;
            }
            return n_regs++;
        };
        auto release_if_dead = [&](E e, int s) {
            if (!is_leaf(e) && last_use[e] == s && reg_of[e] >= 0) {
                free_regs.push_back((uint32_t)reg_of[e]);
                reg_of[e] = -2;  // dead
            }
        };
        uint32_t n_instr = 0;
        for (size_t s = 0; s < stream.size(); s++) {
            if (stream[s].is_step) {
                const Step& st = steps[stream[s].idx];
                if (immediate_only(st.op)) emit(st.op, st.dst, st.a, st.b);
                else {
                    emit(st.op, st.op == airp::OP_IVALT ? st.dst : 0, operand_of(st.node), 0);
                    release_if_dead(st.node, (int)s);
                }
            } else {
                E e = order[stream[s].idx];
                const Node& n = air.nodes[e];
                uint32_t oa = operand_of(n.a), ob = operand_of(n.b);
                // operands may die here: their registers can be reused for the result
                release_if_dead(n.a, (int)s);
                if (n.b != n.a) release_if_dead(n.b, (int)s);
                uint32_t r = alloc();
                reg_of[e] = (int)r;
                uint32_t op = n.kind == N_ADD ? airp::OP_ADD : (n.kind == N_SUB ? airp::OP_SUB : airp::OP_MUL);
                emit(op, r, oa, ob);
                if (last_use[e] < 0) {  // never read (cannot happen for scheduled nodes, but keep the pool sound)
                    free_regs.push_back(r);
                    reg_of[e] = -2;
                }
            }
            n_instr++;
        }
        if (n_regs > airp::SRC_MASK) throw ExecError("AIR program needs too many registers");
        while (n_instr % 4) {  // the VM fetches four instructions at a time
            emit(airp::OP_NOP, 0, 0, 0);
            n_instr++;
        }
        static_assert(airp::H_WORDS % 4 == 0, "code must start 16-byte aligned");
        std::vector<uint32_t> out(airp::H_WORDS, 0);
        out[airp::H_MAGIC] = airp::MAGIC;
        out[airp::H_N_INSTR] = n_instr;
        out[airp::H_N_REGS] = std::max<uint32_t>(n_regs, 1);
        out[airp::H_N_CONSTS] = (uint32_t)consts.size();
        out[airp::H_N_ASSERTS] = n_asserts;
        out[airp::H_N_INTERACTIONS] = n_interactions;
        out[airp::H_N_SENDS] = n_sends;
        out[airp::H_CODE_OFF] = airp::H_WORDS;
        out[airp::H_CONST_OFF] = airp::H_WORDS + (uint32_t)code.size();
        out.insert(out.end(), code.begin(), code.end());
        out.insert(out.end(), consts.begin(), consts.end());
        out[airp::H_TOTAL_WORDS] = (uint32_t)out.size();
        return out;
    }
};

}  // namespace

AirPrograms lower_air(const ChipAir& air) {
    AirPrograms p;
    {
        Lowerer lw(air);
        std::vector<Lowerer::Step> steps;
        for (E c : air.constraints) steps.push_back({airp::OP_ASSERT, c});
        p.constraints = lw.lower(steps, (uint32_t)air.constraints.size(), 0, 0);
    }
    {
        const size_t n_cons = air.constraints.size();
        const size_t n_instr = p.constraints[airp::H_N_INSTR];
        // (LURKHIP_CONS_INSTR_PER_PART: instructions per constraint piece, A/B hook.  1024 in rounds 2-4.  Round 5: the interaction
        // waves of a quotient workgroup skip their dead batches, so the constraint waves became the longest of the workgroup:
        // pieces of 256 instructions and 24 interactions measure quotient_all 5.44 -> 4.30 ms on the fib-mix step (512 / 24: 4.83,
        // 384 / 24: 4.48, 256 / 32: 4.32, 192 / 24: 4.70; ten constraint pieces instead of eight: no gain))
        static const size_t per_piece = getenv("LURKHIP_CONS_INSTR_PER_PART") ? (size_t)std::max(64, atoi(getenv("LURKHIP_CONS_INSTR_PER_PART"))) : 256;
        size_t n_parts = std::min<size_t>({(size_t)airp::MAX_CONSTRAINT_PARTS, std::max<size_t>(1, (n_instr + per_piece - 1) / per_piece), std::max<size_t>(n_cons, 1)});
        // every piece keeps its register file regs[n_regs][64] in LDS next to the other pieces': stay within 128 KiB of the CU's 160
        for (;; n_parts--) {
            p.constraint_parts.clear();
            size_t reg_words = 0;
            for (size_t j = 0; j < n_parts; j++) {
                const size_t k0 = n_cons * j / n_parts, k1 = n_cons * (j + 1) / n_parts;
                Lowerer lw(air);
                std::vector<Lowerer::Step> steps;
                for (size_t k = k0; k < k1; k++) steps.push_back({airp::OP_ASSERT, air.constraints[k]});
                std::vector<uint32_t> prog = lw.lower(steps, (uint32_t)(k1 - k0), 0, 0);
                prog[airp::H_FIRST_COLUMN] = (uint32_t)k0;
                reg_words += (size_t)prog[airp::H_N_REGS] * 64;
                p.constraint_parts.push_back(std::move(prog));
            }
            if (n_parts == 1 || reg_words * 4 <= (size_t)96 * 1024) break;
        }
    }
    std::vector<const Interaction*> all;
    for (const auto& it : air.sends) all.push_back(&it);
    for (const auto& it : air.receives) all.push_back(&it);
    auto lower_range = [&](size_t i0, size_t i1, uint32_t first_column, bool compact) {
        Lowerer lw(air);
        std::vector<Lowerer::Step> steps;
        uint32_t n_sends = 0;
        for (size_t i = i0; i < i1; i++) {
            const Interaction& it = *all[i];
            Lowerer::Step s{airp::OP_IBEGIN, 0};
            s.dst = it.kind;
            s.a = it.is_send ? 1 : 0;
            s.b = compact ? (uint32_t)i : (uint32_t)it.values.size();
            steps.push_back(s);
            if (!compact) {
                for (E v : it.values) steps.push_back({airp::OP_IVAL, v});
            } else {
                // constants are folded into the start value; runs of consecutive main columns become one instruction
                for (size_t k = 0; k < it.values.size();) {
                    const Node& n = air.nodes[it.values[k]];
                    if (n.kind == N_CONST) {
                        k++;
                        continue;
                    }
                    if (n.kind == N_MAIN) {
                        size_t run = 1;
                        while (k + run < it.values.size() && run < 255) {
                            const Node& m = air.nodes[it.values[k + run]];
                            if (m.kind != N_MAIN || m.a != n.a + run) break;
                            run++;
                        }
                        Lowerer::Step r{airp::OP_IVALS, 0};
                        r.dst = (uint32_t)run;
                        r.a = n.a;
                        r.b = (uint32_t)k + 1;
                        steps.push_back(r);
                        k += run;
                        continue;
                    }
                    Lowerer::Step v{airp::OP_IVALT, it.values[k]};
                    v.dst = (uint32_t)k + 1;
                    steps.push_back(v);
                    k++;
                }
            }
            steps.push_back({airp::OP_IEND, it.mult});
            if (it.is_send) n_sends++;
        }
        std::vector<uint32_t> prog = lw.lower(steps, 0, (uint32_t)(i1 - i0), n_sends);
        prog[airp::H_FIRST_COLUMN] = first_column;
        return prog;
    };
    p.interactions = lower_range(0, all.size(), 0, false);
    for (size_t i = 0; i < all.size(); i++) {
        p.interaction_kinds.push_back(all[i]->kind);
        for (size_t k = 0; k < all[i]->values.size(); k++) {
            const Node& n = air.nodes[all[i]->values[k]];
            if (n.kind == N_CONST && n.a != 0) p.const_terms.push_back({(uint32_t)i, (uint32_t)k + 1, n.a});
        }
    }
    // pieces of about `per_part` interactions, at most six, each a whole number of batches
    const size_t batch = (size_t)1 << air.log_quotient_degree();
    const size_t n_batches = (all.size() + batch - 1) / batch;
    auto cut = [&](size_t per_part, std::vector<std::vector<uint32_t>>& out) {
        size_t n_parts = std::min<size_t>(6, std::max<size_t>(1, (all.size() + per_part - 1) / per_part));
        n_parts = std::min(n_parts, std::max<size_t>(n_batches, 1));
        for (size_t j = 0; j < n_parts; j++) {
            const size_t b0 = n_batches * j / n_parts, b1 = n_batches * (j + 1) / n_parts;
            out.push_back(lower_range(std::min(b0 * batch, all.size()), std::min(b1 * batch, all.size()), (uint32_t)b0, true));
        }
    };
    static const size_t perm_per_part = getenv("LURKHIP_PERM_PER_PART") ? (size_t)std::max(2, atoi(getenv("LURKHIP_PERM_PER_PART"))) : 12;  // (A/B hook)
    cut(perm_per_part, p.interaction_parts);
    // (LURKHIP_QUOT_PER_PART: interactions per quotient piece, A/B hook; round 2 settled on 24 -- more pieces were slower then: the per-lane reads of the permutation row thrashed L1; round 4, with the sinks' arithmetic a third cheaper, 12 measures 3.02 -> 2.83 ms for the quotient stage, 8 and 16 2.9)
    // (round 5, with the dead-batch skip: 24 again, see the constraint pieces above)
    static const size_t quot_per_part = getenv("LURKHIP_QUOT_PER_PART") ? (size_t)std::max(2, atoi(getenv("LURKHIP_QUOT_PER_PART"))) : 24;
    cut(quot_per_part, p.interaction_parts_coarse);
    return p;
}

}  // namespace lair
