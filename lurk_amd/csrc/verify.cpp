// Host verifier of machine proofs: the consumer on the other side of the proving path.
//
// Replaces (third-party, source absent from /root/reference; [UPSTREAM-RECALL], parity unpinned like the prover's):
//   sphinx StarkMachine::verify / Verifier::verify_shard, p3 TwoAdicFriPcs::verify and p3_fri::verifier::verify
//   (call sites: `machine.verify(&vk, &proof, &mut challenger)`, /root/reference/benches/fib.rs:105-133,
//    /root/reference/src/lair/lair_chip.rs:246-276, /root/reference/src/core/cli/repl.rs verify path).
// Nothing here touches a device: the transcript, the Merkle paths, the FRI queries and the constraint identity at zeta are a
// few thousand width-16 permutations and a few hundred extension-field operations per chip.  A chip's constraints are
// evaluated on the opened values straight from its symbolic AIR (lair::ChipAir: the node list is in topological order).
// Every choice the prover takes from the protocol profile (include/lurkhip.h) is taken from the same profile here.
#include <array>
#include <cstdarg>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lurkhip.h"
#include "babybear.h"
#include "challenger.h"
#include "commit.h"
#include "stark.h"

namespace lurkhip {
P16Params p16_tables_of(const lurkhip_protocol_profile& p);  // merkle.hip
}

namespace {

using bb::ef;
using lurkhip::Challenger;
using lurkhip::P16Params;

constexpr uint32_t PROOF_MAGIC = 0x4652504cu;  // "LPRF" (prover.hip)

struct Reject : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void reject(const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Reject(buf);
}
#define NEED(cond, ...)                 \
    do {                                \
        if (!(cond)) reject(__VA_ARGS__); \
    } while (0)

uint32_t pow_m(uint32_t a_m, uint64_t e) {
    uint32_t r = bb::R1;
    while (e) {
        if (e & 1) r = bb::mul(r, a_m);
        a_m = bb::mul(a_m, a_m);
        e >>= 1;
    }
    return r;
}
uint32_t inv_m(uint32_t a_m) { return pow_m(a_m, bb::P - 2); }
ef ef_pow(ef a, uint64_t e) {
    ef r = bb::ef_one();
    while (e) {
        if (e & 1) r = bb::ef_mul(r, a);
        a = bb::ef_sqr(a);
        e >>= 1;
    }
    return r;
}
bool ef_eq(const ef& a, const ef& b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3]; }
uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

// ---- a parsed shard proof (layout: prover.hip's serialisation, lurk_amd/prover.py: parse_proof); values in Montgomery form
struct VChip {
    uint32_t machine_index, log_n, width, prep_width, perm_width, qd;
    int prep_index;
    ef cumsum;
    std::vector<ef> prep[2], main[2], perm[2];
    std::vector<std::array<ef, 4>> quotient;
};
struct VProof {
    uint32_t log_blowup, nq, pow_bits, n_public, n_layers, log_max, n_prep, n_chunks;
    std::vector<VChip> chips;
    std::vector<uint32_t> pub;  // canonical
    uint32_t main_root[8], perm_root[8], quot_root[8];
    std::vector<std::array<uint32_t, 8>> fri_roots;
    ef final_poly;
    uint32_t pow_witness;  // canonical
    std::vector<uint32_t> indices;
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> rounds, layers;  // (record words, nq records back to back; Montgomery)
};

struct Cursor {
    const uint32_t* w;
    uint64_t n, pos = 0;
    uint32_t u() {
        NEED(pos < n, "truncated proof");
        return w[pos++];
    }
    uint32_t f() {  // a field element: canonical on the wire
        const uint32_t v = u();
        NEED(v < bb::P, "proof holds a non-canonical field element");
        return bb::to_monty(v);
    }
    ef e() {
        ef r;
        for (int i = 0; i < 4; i++) r.c[i] = f();
        return r;
    }
    void efs(std::vector<ef>& out, uint32_t count) {
        NEED((uint64_t)count * 4 <= n - pos, "truncated proof");
        out.resize(count);
        for (uint32_t i = 0; i < count; i++) out[i] = e();
    }
};

VProof parse(const uint32_t* words, uint64_t n_words) {
    Cursor c{words, n_words};
    VProof p;
    NEED(c.u() == PROOF_MAGIC, "not a lurkhip proof (bad magic)");
    const uint32_t n_chips = c.u();
    p.log_blowup = c.u(), p.nq = c.u(), p.pow_bits = c.u(), p.n_public = c.u(), p.n_layers = c.u(), p.log_max = c.u(), p.n_prep = c.u(),
    p.n_chunks = c.u();
    NEED(n_chips >= 1 && n_chips <= 4096 && p.log_blowup >= 1 && p.log_blowup <= 4 && p.nq >= 1 && p.nq <= 1024 && p.pow_bits <= 30 &&
             p.n_layers <= 40 && p.n_public <= 4096 && p.n_prep <= n_chips && p.log_max <= 40,
         "implausible proof header");
    p.chips.resize(n_chips);
    for (VChip& ch : p.chips) {
        ch.machine_index = c.u(), ch.log_n = c.u(), ch.width = c.u(), ch.prep_width = c.u(), ch.perm_width = c.u(), ch.qd = c.u();
        ch.prep_index = (int)c.u() - 1;
        NEED(ch.log_n <= 32 && ch.width <= (1u << 20) && ch.prep_width <= (1u << 20) && ch.perm_width <= (1u << 20) && ch.perm_width % 4 == 0 &&
                 ch.perm_width >= 4 && ch.qd >= 1 && ch.qd <= 16 && (ch.qd & (ch.qd - 1)) == 0 && ch.prep_index < (int)p.n_prep,
             "implausible chip header");
        ch.cumsum = c.e();
    }
    p.pub.resize(p.n_public);
    for (uint32_t& v : p.pub) {
        v = c.u();
        NEED(v < bb::P, "public value is not canonical");
    }
    for (uint32_t* root : {p.main_root, p.perm_root, p.quot_root})
        for (int i = 0; i < 8; i++) root[i] = c.f();
    if (p.n_prep) {
        std::map<int, VChip*> by_idx;
        for (VChip& ch : p.chips)
            if (ch.prep_index >= 0) NEED(by_idx.emplace(ch.prep_index, &ch).second, "two chips share a preprocessed trace");
        for (uint32_t m = 0; m < p.n_prep; m++) {
            auto it = by_idx.find((int)m);
            NEED(it != by_idx.end(), "a preprocessed trace has no chip");
            for (int k = 0; k < 2; k++) c.efs(it->second->prep[k], it->second->prep_width);
        }
    }
    for (VChip& ch : p.chips)
        for (int k = 0; k < 2; k++) c.efs(ch.main[k], ch.width);
    for (VChip& ch : p.chips)
        for (int k = 0; k < 2; k++) c.efs(ch.perm[k], ch.perm_width);
    uint32_t chunks = 0;
    for (VChip& ch : p.chips) {
        ch.quotient.resize(ch.qd);
        for (auto& q : ch.quotient)
            for (int e = 0; e < 4; e++) q[e] = c.e();
        chunks += ch.qd;
    }
    NEED(chunks == p.n_chunks, "quotient chunk count mismatch");
    p.fri_roots.resize(p.n_layers);
    for (auto& r : p.fri_roots)
        for (int i = 0; i < 8; i++) r[i] = c.f();
    p.final_poly = c.e();
    p.pow_witness = c.u();
    p.indices.resize(p.nq);
    for (uint32_t& ix : p.indices) ix = c.u();
    auto records = [&](std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& out, uint32_t count) {
        out.resize(count);
        for (auto& r : out) {
            r.first = c.u();
            const uint64_t total = (uint64_t)r.first * p.nq;
            NEED(total <= c.n - c.pos, "truncated proof");
            r.second.resize(total);
            for (uint64_t k = 0; k < total; k++) r.second[k] = c.f();
        }
    };
    records(p.rounds, 3 + (p.n_prep ? 1 : 0));
    records(p.layers, p.n_layers);
    NEED(c.pos == c.n, "trailing words in proof");
    return p;
}

// ---- Merkle (p3 FieldMerkleTreeMmcs::verify_batch with PaddingFreeSponge<16, 8, 8> / TruncatedPermutation<16, 2, 8>)
struct Hasher {
    const P16Params& p;
    void sponge(const std::vector<std::pair<const uint32_t*, uint32_t>>& rows, uint32_t out[8]) const {
        uint32_t s[16] = {};
        int pos = 0;
        for (const auto& r : rows)
            for (uint32_t k = 0; k < r.second; k++) {
                s[pos++] = r.first[k];
                if (pos == 8) {
                    lurkhip::host_perm16(p, s);
                    pos = 0;
                }
            }
        if (pos) lurkhip::host_perm16(p, s);
        memcpy(out, s, 32);
    }
    void compress(const uint32_t* l, const uint32_t* r, uint32_t out[8]) const {
        uint32_t s[16];
        memcpy(s, l, 32);
        memcpy(s + 8, r, 32);
        lurkhip::host_perm16(p, s);
        memcpy(out, s, 32);
    }
    // rows: the opened rows of every matrix back to back (the prover's order); path: log_max sibling digests, leaf level first
    bool verify(const std::vector<uint32_t>& log_h, const std::vector<uint32_t>& widths, uint64_t index, const uint32_t* rows, const uint32_t* path,
                const uint32_t* root) const {
        uint32_t log_max = 0;
        for (uint32_t h : log_h) log_max = std::max(log_max, h);
        std::vector<const uint32_t*> ptr(log_h.size());
        size_t off = 0;
        for (size_t i = 0; i < log_h.size(); i++) ptr[i] = rows + off, off += widths[i];
        auto group = [&](uint32_t h, uint32_t out[8]) {
            std::vector<std::pair<const uint32_t*, uint32_t>> g;
            for (size_t i = 0; i < log_h.size(); i++)
                if (log_h[i] == h) g.push_back({ptr[i], widths[i]});
            if (g.empty()) return false;
            sponge(g, out);
            return true;
        };
        uint32_t cur[8];
        group(log_max, cur);
        for (uint32_t l = 0; l < log_max; l++) {
            uint32_t d[8], h[8];
            if (((index >> l) & 1) == 0) compress(cur, path + l * 8, d);
            else compress(path + l * 8, cur, d);
            if (group(log_max - l - 1, h)) compress(d, h, cur);
            else memcpy(cur, d, 32);
        }
        return memcmp(cur, root, 32) == 0;
    }
};

// ---- one opening round as the verifier sees it: (root, [(log_n, width, [(point, values)])])
struct VMat {
    uint32_t log_n, width;
    std::vector<std::pair<ef, const std::vector<ef>*>> pts;
};
struct VRound {
    const uint32_t* root;
    std::vector<VMat> mats;
};

void pcs_verify(const lurkhip_protocol_profile& prof, const Hasher& H, const std::vector<VRound>& rounds, const VProof& p, Challenger& ch) {
    const uint32_t gen_m = bb::to_monty(31);
    if (prof.observe_openings)
        for (const VRound& r : rounds)
            for (const VMat& m : r.mats)
                for (const auto& pt : m.pts)
                    for (const ef& v : *pt.second) ch.observe_ef_m(v);
    const ef alpha_fri = ch.sample_ef_m();
    std::vector<ef> betas;
    for (const auto& root : p.fri_roots) {
        ch.observe_digest_m(root.data());
        betas.push_back(ch.sample_ef_m());
    }
    ch.observe_ef_m(p.final_poly);
    NEED(ch.check_witness((int)p.pow_bits, p.pow_witness), "invalid proof-of-work witness");
    const uint32_t log_max = p.n_layers + p.log_blowup;
    NEED(log_max == p.log_max && log_max <= 31, "log_max_height");
    NEED(rounds.size() == p.rounds.size(), "number of opening rounds");
    for (uint32_t qi = 0; qi < p.nq; qi++) {
        const uint32_t index = ch.sample_bits((int)log_max);
        NEED(index == p.indices[qi], "query indices differ from the transcript's");
        std::map<uint32_t, ef> ro, alpha_pow;
        for (size_t ri = 0; ri < rounds.size(); ri++) {
            const VRound& r = rounds[ri];
            std::vector<uint32_t> log_hs, widths;
            uint32_t total_w = 0, log_batch_max = 0;
            for (const VMat& m : r.mats) {
                log_hs.push_back(m.log_n + p.log_blowup);
                widths.push_back(m.width);
                total_w += m.width;
                log_batch_max = std::max(log_batch_max, m.log_n + p.log_blowup);
            }
            NEED(log_batch_max <= log_max, "a matrix is taller than the FRI domain");
            const uint32_t rw = p.rounds[ri].first;
            NEED(rw == total_w + 8 * log_batch_max, "round record size");
            const uint32_t* rec = p.rounds[ri].second.data() + (size_t)qi * rw;
            const uint32_t reduced_index = index >> (log_max - log_batch_max);
            NEED(H.verify(log_hs, widths, reduced_index, rec, rec + total_w, r.root), "Merkle opening of query %u fails (round %zu)", qi, ri);
            size_t off = 0;
            for (size_t mi = 0; mi < r.mats.size(); mi++) {
                const VMat& m = r.mats[mi];
                const uint32_t log_h = log_hs[mi];
                const uint32_t* row = rec + off;
                off += m.width;
                const uint32_t rev = bitrev(index >> (log_max - log_h), (int)log_h);
                const uint32_t x = bb::mul(gen_m, pow_m(lurkhip::two_adic_generator_monty((int)log_h), rev));
                for (const auto& pt : m.pts) {
                    NEED(pt.second->size() == m.width, "opened values shape");
                    const ef d = bb::ef_sub(bb::ef_from_base(x), pt.first);
                    NEED(!bb::ef_is_zero(d), "an opening point lies in the evaluation domain");
                    const ef inv_d = bb::ef_inv(d);
                    const uint32_t key = prof.fri_alpha_global ? 0 : log_h;  // p3: one power offset per LDE height
                    ef ap = alpha_pow.count(key) ? alpha_pow[key] : bb::ef_one();
                    ef acc = ro.count(log_h) ? ro[log_h] : bb::ef_zero();
                    for (uint32_t k = 0; k < m.width; k++) {
                        const ef quotient = bb::ef_mul(bb::ef_sub(bb::ef_from_base(row[k]), (*pt.second)[k]), inv_d);
                        acc = bb::ef_add(acc, bb::ef_mul(ap, quotient));
                        ap = bb::ef_mul(ap, alpha_fri);
                    }
                    alpha_pow[key] = ap;
                    ro[log_h] = acc;
                }
            }
        }
        // ---- p3_fri verify_query
        ef folded = bb::ef_zero();
        uint32_t idx = index;
        uint32_t x = pow_m(lurkhip::two_adic_generator_monty((int)log_max), bitrev(index, (int)log_max));
        const uint32_t minus_one = bb::sub(0, bb::R1);
        for (uint32_t li = 0; li < p.n_layers; li++) {
            const uint32_t log_folded = log_max - 1 - li;
            if (ro.count(log_folded + 1)) folded = bb::ef_add(folded, ro[log_folded + 1]);
            const uint32_t rw = p.layers[li].first;
            NEED(rw == 8 + 8 * log_folded, "layer record size");
            const uint32_t* rec = p.layers[li].second.data() + (size_t)qi * rw;
            ef evals[2] = {ef{{rec[0], rec[1], rec[2], rec[3]}}, ef{{rec[4], rec[5], rec[6], rec[7]}}};
            NEED(ef_eq(evals[idx & 1], folded), "query %u: layer %u does not continue the fold", qi, li);
            NEED(H.verify({log_folded}, {8}, idx >> 1, rec, rec + 8, p.fri_roots[li].data()), "query %u: FRI layer %u opening fails", qi, li);
            uint32_t xs[2] = {x, x};
            xs[(idx ^ 1) & 1] = bb::mul(xs[(idx ^ 1) & 1], minus_one);  // times the generator of the order-2 subgroup
            // the line through (xs[0], evals[0]), (xs[1], evals[1]) at beta
            const ef slope = bb::ef_scale(bb::ef_sub(evals[1], evals[0]), inv_m(bb::sub(xs[1], xs[0])));
            folded = bb::ef_add(evals[0], bb::ef_mul(bb::ef_sub(betas[li], bb::ef_from_base(xs[0])), slope));
            idx >>= 1;
            x = bb::mul(x, x);
        }
        NEED(idx < (1u << p.log_blowup), "final index");
        NEED(ef_eq(folded, p.final_poly), "query %u: final polynomial mismatch", qi);
    }
}

// sphinx Verifier::eval_constraints on the opened values: the chip's constraints, then eval_permutation_constraints, folded
ef eval_constraints_at(const lair::ChipAir& air, const VChip& c, const ef sels[3], const ef& alpha, const ef& perm_alpha, const ef& perm_beta,
                       const std::vector<uint32_t>& pub, bool ascending) {
    std::vector<ef> v(air.nodes.size());
    for (size_t i = 0; i < air.nodes.size(); i++) {
        const lair::Node& n = air.nodes[i];
        auto col = [&](const std::vector<ef>& row, uint32_t k) {
            NEED(k < row.size(), "AIR of chip %u reads column %u of %zu", c.machine_index, k, row.size());
            return row[k];
        };
        switch (n.kind) {
            case lair::N_CONST: v[i] = bb::ef_from_base(bb::to_monty(n.a % bb::P)); break;
            case lair::N_MAIN: v[i] = col(c.main[0], n.a); break;
            case lair::N_MAIN_NEXT: v[i] = col(c.main[1], n.a); break;
            case lair::N_PREP: v[i] = col(c.prep[0], n.a); break;
            case lair::N_PREP_NEXT: v[i] = col(c.prep[1], n.a); break;
            case lair::N_PUBLIC:
                NEED(n.a < pub.size(), "AIR of chip %u reads public value %u", c.machine_index, n.a);
                v[i] = bb::ef_from_base(bb::to_monty(pub[n.a]));
                break;
            case lair::N_IS_FIRST: v[i] = sels[0]; break;
            case lair::N_IS_LAST: v[i] = sels[1]; break;
            case lair::N_IS_TRANS: v[i] = sels[2]; break;
            case lair::N_ADD: v[i] = bb::ef_add(v[n.a], v[n.b]); break;
            case lair::N_SUB: v[i] = bb::ef_sub(v[n.a], v[n.b]); break;
            case lair::N_MUL: v[i] = bb::ef_mul(v[n.a], v[n.b]); break;
        }
    }
    std::vector<ef> terms;
    for (lair::E e : air.constraints) terms.push_back(v[e]);
    // 4 opened base columns -> one extension element: sum_e x^e v_e
    auto unflatten = [&](const std::vector<ef>& flat) {
        std::vector<ef> out;
        for (size_t j = 0; j + 4 <= flat.size(); j += 4) {
            ef acc = bb::ef_zero();
            for (int e = 0; e < 4; e++) {
                ef mono = bb::ef_zero();
                mono.c[e] = bb::R1;
                acc = bb::ef_add(acc, bb::ef_mul(mono, flat[j + e]));
            }
            out.push_back(acc);
        }
        return out;
    };
    const std::vector<ef> perm_local = unflatten(c.perm[0]), perm_next = unflatten(c.perm[1]);
    struct It {
        const lair::Interaction* it;
        bool send;
    };
    std::vector<It> its;
    for (const auto& s : air.sends) its.push_back({&s, true});
    for (const auto& r : air.receives) its.push_back({&r, false});
    const size_t batch = c.qd, n_cols = perm_local.size();
    NEED(n_cols == (its.size() + batch - 1) / batch + 1, "permutation width of chip %u", c.machine_index);
    for (size_t col = 0, c0 = 0; c0 < its.size(); col++, c0 += batch) {
        const size_t n = std::min(batch, its.size() - c0);
        std::vector<ef> rlc(n), mult(n);
        for (size_t i = 0; i < n; i++) {
            const lair::Interaction& x = *its[c0 + i].it;
            ef d = bb::ef_add_base(perm_alpha, bb::to_monty(x.kind % bb::P)), bp = perm_beta;
            for (lair::E e : x.values) {
                d = bb::ef_add(d, bb::ef_mul(bp, v[e]));
                bp = bb::ef_mul(bp, perm_beta);
            }
            rlc[i] = d;
            mult[i] = its[c0 + i].send ? v[x.mult] : bb::ef_sub(bb::ef_zero(), v[x.mult]);
        }
        ef product = bb::ef_one(), numerator = bb::ef_zero();
        for (size_t i = 0; i < n; i++) {
            product = bb::ef_mul(product, rlc[i]);
            ef others = bb::ef_one();
            for (size_t j = 0; j < n; j++)
                if (j != i) others = bb::ef_mul(others, rlc[j]);
            numerator = bb::ef_add(numerator, bb::ef_mul(mult[i], others));
        }
        terms.push_back(bb::ef_sub(bb::ef_mul(product, perm_local[col]), numerator));
    }
    ef sum_local = bb::ef_zero(), sum_next = bb::ef_zero();
    for (size_t k = 0; k + 1 < n_cols; k++) sum_local = bb::ef_add(sum_local, perm_local[k]), sum_next = bb::ef_add(sum_next, perm_next[k]);
    const ef phi_local = perm_local[n_cols - 1], phi_next = perm_next[n_cols - 1];
    terms.push_back(bb::ef_mul(bb::ef_sub(phi_local, sum_local), sels[0]));
    terms.push_back(bb::ef_mul(bb::ef_sub(bb::ef_sub(phi_next, phi_local), sum_next), sels[2]));
    terms.push_back(bb::ef_mul(bb::ef_sub(phi_local, c.cumsum), sels[1]));
    ef acc = bb::ef_zero();
    if (ascending) {  // constraint k weighs alpha^k
        ef pw = bb::ef_one();
        for (const ef& t : terms) acc = bb::ef_add(acc, bb::ef_mul(pw, t)), pw = bb::ef_mul(pw, alpha);
    } else {  // sphinx's folders: accumulator = accumulator * alpha + constraint
        for (const ef& t : terms) acc = bb::ef_add(bb::ef_mul(acc, alpha), t);
    }
    return acc;
}

// sphinx Verifier::recompute_quotient: chunk i lives on the coset 31 * w_Q^i * H
ef recompute_quotient(const VChip& c, const ef& zeta) {
    uint32_t lqd = 0;
    while ((1u << lqd) < c.qd) lqd++;
    const uint64_t n = (uint64_t)1 << c.log_n;
    const uint32_t wq = lurkhip::two_adic_generator_monty((int)(c.log_n + lqd)), gen_m = bb::to_monty(31);
    std::vector<uint32_t> shifts(c.qd);
    for (uint32_t i = 0; i < c.qd; i++) shifts[i] = bb::mul(gen_m, pow_m(wq, i));
    auto zp_at = [&](uint32_t shift, const ef& x) {  // (x / shift)^n - 1
        return bb::ef_sub(bb::ef_scale(ef_pow(x, n), inv_m(pow_m(shift, n))), bb::ef_one());
    };
    ef total = bb::ef_zero();
    for (uint32_t i = 0; i < c.qd; i++) {
        ef zps = bb::ef_one();
        for (uint32_t j = 0; j < c.qd; j++)
            if (j != i) {
                const ef den = zp_at(shifts[j], bb::ef_from_base(shifts[i]));
                NEED(!bb::ef_is_zero(den), "degenerate quotient chunk domains");
                zps = bb::ef_mul(zps, bb::ef_mul(zp_at(shifts[j], zeta), bb::ef_inv(den)));
            }
        ef acc = bb::ef_zero();
        for (int e = 0; e < 4; e++) {
            ef mono = bb::ef_zero();
            mono.c[e] = bb::R1;
            acc = bb::ef_add(acc, bb::ef_mul(mono, c.quotient[i][e]));
        }
        total = bb::ef_add(total, bb::ef_mul(zps, acc));
    }
    return total;
}

struct VerifyInput {
    const lurkhip_protocol_profile& prof;
    const Hasher& H;
    const lurkhip_air* const* airs;
    uint32_t n_airs;
    const uint32_t* vk_root_m;
    const uint32_t *prep_log_heights, *prep_widths;
    uint32_t n_prep;
};

// sphinx Verifier::verify_shard; `ch` must be in the state the prover's transcript had when prove_shard started
void verify_shard(const VerifyInput& in, const VProof& p, Challenger ch, ef* sum) {
    const lurkhip_protocol_profile& prof = in.prof;
    NEED(p.n_prep == 0 || p.n_prep == in.n_prep, "the proof opens %u preprocessed traces, the verifying key has %u", p.n_prep, in.n_prep);
    for (const VChip& c : p.chips) {
        NEED(c.machine_index < in.n_airs && in.airs[c.machine_index], "chip with machine index %u is not part of the machine", c.machine_index);
        const lair::ChipAir& air = lurkhip::air_of(in.airs[c.machine_index]);
        NEED(air.width == c.width && air.prep_width == c.prep_width, "shape of chip %u", c.machine_index);
        NEED(c.qd == (1u << air.log_quotient_degree()), "quotient degree of chip %u", c.machine_index);
        NEED((air.prep_width != 0) == (c.prep_index >= 0), "preprocessed trace of chip %u", c.machine_index);
    }
    if (prof.observe_chip_meta)
        for (const VChip& c : p.chips) ch.observe(c.log_n), ch.observe(c.width), ch.observe((uint32_t)(c.prep_index + 1));
    const ef perm_alpha = ch.sample_ef_m(), perm_beta = ch.sample_ef_m();
    ch.observe_digest_m(p.perm_root);
    if (prof.observe_chip_meta)
        for (const VChip& c : p.chips) ch.observe_ef_m(c.cumsum);
    const ef alpha = ch.sample_ef_m();
    ch.observe_digest_m(p.quot_root);
    const ef zeta = ch.sample_ef_m();
    auto next_point = [&](uint32_t log_n) { return bb::ef_scale(zeta, lurkhip::two_adic_generator_monty((int)log_n)); };

    std::vector<VRound> rounds;
    std::vector<std::vector<ef>> chunk_values;  // the quotient chunks' four opened values, as rows
    if (p.n_prep) {
        VRound r{in.vk_root_m, {}};
        for (uint32_t m = 0; m < p.n_prep; m++) {
            const VChip* c = nullptr;
            for (const VChip& x : p.chips)
                if (x.prep_index == (int)m) c = &x;
            NEED(c && c->log_n == in.prep_log_heights[m] && c->prep_width == in.prep_widths[m], "preprocessed shape");
            r.mats.push_back(VMat{c->log_n, c->prep_width, {{zeta, &c->prep[0]}, {next_point(c->log_n), &c->prep[1]}}});
        }
        rounds.push_back(std::move(r));
    }
    {
        VRound r{p.main_root, {}}, s{p.perm_root, {}};
        for (const VChip& c : p.chips) {
            r.mats.push_back(VMat{c.log_n, c.width, {{zeta, &c.main[0]}, {next_point(c.log_n), &c.main[1]}}});
            s.mats.push_back(VMat{c.log_n, c.perm_width, {{zeta, &c.perm[0]}, {next_point(c.log_n), &c.perm[1]}}});
        }
        rounds.push_back(std::move(r));
        rounds.push_back(std::move(s));
    }
    {
        size_t total = 0;
        for (const VChip& c : p.chips) total += c.qd;
        chunk_values.reserve(total);
        VRound r{p.quot_root, {}};
        for (const VChip& c : p.chips)
            for (const auto& q : c.quotient) {
                chunk_values.push_back(std::vector<ef>(q.begin(), q.end()));
                r.mats.push_back(VMat{c.log_n, 4, {{zeta, &chunk_values.back()}}});
            }
        rounds.push_back(std::move(r));
    }
    pcs_verify(prof, in.H, rounds, p, ch);

    for (const VChip& c : p.chips) {
        const lair::ChipAir& air = lurkhip::air_of(in.airs[c.machine_index]);
        const uint64_t n = (uint64_t)1 << c.log_n;
        const ef zh = bb::ef_sub(ef_pow(zeta, n), bb::ef_one());
        const uint32_t w_inv = inv_m(lurkhip::two_adic_generator_monty((int)c.log_n));
        const ef d_first = bb::ef_sub(zeta, bb::ef_one()), d_last = bb::ef_sub(zeta, bb::ef_from_base(w_inv));
        NEED(!bb::ef_is_zero(zh) && !bb::ef_is_zero(d_first) && !bb::ef_is_zero(d_last), "zeta lies in the trace domain");
        const ef sels[3] = {bb::ef_mul(zh, bb::ef_inv(d_first)), bb::ef_mul(zh, bb::ef_inv(d_last)), d_last};
        const ef folded = eval_constraints_at(air, c, sels, alpha, perm_alpha, perm_beta, p.pub, prof.constraint_alpha_ascending != 0);
        const ef quotient = recompute_quotient(c, zeta);
        NEED(ef_eq(bb::ef_mul(folded, bb::ef_inv(zh)), quotient), "constraints of chip %u do not match the quotient at zeta", c.machine_index);
        *sum = bb::ef_add(*sum, c.cumsum);
    }
}

}  // namespace

extern "C" int32_t lurkhip_machine_verify(const lurkhip_protocol_profile* profile, const lurkhip_air* const* airs, uint32_t n_airs,
                                          const uint32_t* vk_root, const uint32_t* prep_log_heights, const uint32_t* prep_widths, uint32_t n_prep,
                                          const uint32_t* const* proofs, const uint64_t* proof_words, uint32_t n_proofs, char* err,
                                          uint32_t err_cap) {
    auto say = [&](const std::string& m) {
        if (err && err_cap) snprintf(err, err_cap, "%s", m.c_str());
    };
    if (err && err_cap) err[0] = 0;
    if (!airs || !n_airs || !vk_root || !proofs || !proof_words || !n_proofs || (n_prep && (!prep_log_heights || !prep_widths))) {
        say("null or empty argument");
        return LURKHIP_ERR_INVALID_ARG;
    }
    lurkhip_protocol_profile prof;
    if (profile) {
        prof = *profile;
        if (prof.struct_bytes != sizeof prof || prof.p16_rounds_p < 1 || prof.p16_rounds_p > (uint32_t)lurkhip::P16_MAX_RP ||
            prof.p16_internal_scale % bb::P == 0 || (prof.challenger_squeeze != 8 && prof.challenger_squeeze != 16) || prof.fri_log_arity != 1) {
            say("invalid protocol profile");
            return LURKHIP_ERR_INVALID_ARG;
        }
    } else if (lurkhip_protocol_profile_preset("default", &prof) != LURKHIP_OK) {
        say("no default profile");
        return LURKHIP_ERR_INVALID_ARG;
    }
    try {
        const P16Params tables = lurkhip::p16_tables_of(prof);
        const Hasher H{tables};
        uint32_t vk_m[8];
        for (int i = 0; i < 8; i++) {
            NEED(vk_root[i] < bb::P, "verifying-key root is not canonical");
            vk_m[i] = bb::to_monty(vk_root[i]);
        }
        std::vector<VProof> parsed;
        for (uint32_t s = 0; s < n_proofs; s++) {
            NEED(proofs[s] != nullptr, "null proof");
            parsed.push_back(parse(proofs[s], proof_words[s]));
        }
        // sphinx StarkMachine::verify: the verifying key, pc_start, then every shard's main root and public values
        Challenger ch;
        ch.params = &tables;
        ch.squeeze = (int)prof.challenger_squeeze;
        ch.pop_front = prof.challenger_pop_front != 0;
        ch.observe_digest_m(vk_m);
        ch.observe(0);
        for (const VProof& p : parsed) {
            ch.observe_digest_m(p.main_root);
            for (uint32_t v : p.pub) ch.observe(v);
        }
        const VerifyInput in{prof, H, airs, n_airs, vk_m, prep_log_heights, prep_widths, n_prep};
        ef total = bb::ef_zero();
        for (size_t s = 0; s < parsed.size(); s++) {
            try {
                verify_shard(in, parsed[s], ch, &total);
            } catch (const Reject& r) {
                reject("shard %zu: %s", s, r.what());
            }
        }
        NEED(bb::ef_is_zero(total), "cumulative sums do not cancel");
    } catch (const Reject& r) {
        say(r.what());
        return LURKHIP_ERR_VERIFY;
    } catch (const std::bad_alloc&) {
        say("host allocation failed");
        return LURKHIP_ERR_OOM;
    } catch (const std::exception& e) {
        say(std::string("internal error: ") + e.what());
        return LURKHIP_ERR_EXEC;
    }
    return LURKHIP_OK;
}
