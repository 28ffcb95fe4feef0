// Proof wire format: the flat proof words of lurkhip_shard_prove re-encoded as the reference's serialised proofs.
//
// /root/reference/src/core/cli/proofs.rs:22-35: `CryptoProof { shard_proofs: Vec<CryptoShardProof>, verifier_version: String,
// depth: u32 }`, `CryptoShardProof { commitment, opened_values, opening_proof, chip_ordering }`, written with
// `bincode::serialize` (/root/reference/src/core/cli/repl.rs:200-203); proofs.rs:137-143: `CachedProof { crypto_proof, expr,
// env, result, zdag }`.  The public values are NOT part of a CryptoProof: the verifier rebuilds the 44 lanes
// [expr flat 16 | env digest 8 | result flat 16 | depth as 4 LE bytes] (proofs.rs:46-56, stark_machine.rs:16-17).
//
// bincode 1.x defaults: little-endian fixed-width integers, usize and every length as u64, enum variant index as u32,
// String = length + UTF-8 bytes, fixed-size arrays and struct fields back to back.
//
// The inner types live in sphinx-core / Plonky3 (absent from /root/reference) and are restated from memory [UPSTREAM-RECALL]:
//   ShardCommitment<C>        { main_commit: C, permutation_commit: C, quotient_commit: C }        C = [F; 8]
//   ShardOpenedValues<T>      { chips: Vec<ChipOpenedValues<T>> }                                  T = [F; 4]
//   ChipOpenedValues<T>       { preprocessed, main, permutation: AirOpenedValues<T>, quotient: Vec<Vec<T>>, cumulative_sum: T,
//                               log_degree: usize }
//   AirOpenedValues<T>        { local: Vec<T>, next: Vec<T> }
//   TwoAdicFriPcsProof        { fri_proof: FriProof, query_openings: Vec<Vec<BatchOpening>> }      [query][round]
//   FriProof                  { commit_phase_commits: Vec<C>, query_proofs: Vec<QueryProof>, final_poly: T, pow_witness: F }
//   QueryProof                { commit_phase_openings: Vec<CommitPhaseProofStep> }
//   CommitPhaseProofStep      { sibling_value: T, opening_proof: Vec<[F; 8]> }
//   BatchOpening              { opened_values: Vec<Vec<F>>, opening_proof: Vec<[F; 8]> }
//   F (BabyBear) as one u32: canonical, or the Montgomery word when lurkhip_protocol_profile::serialize_montgomery is set.
// `chip_ordering` is a hashbrown HashMap upstream, whose iteration order is not deterministic; entries are written here in
// the order of their indices.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lurkhip.h"

namespace {

constexpr uint32_t P = 2013265921u;
constexpr uint32_t PROOF_MAGIC = 0x4652504cu;  // "LPRF" (prover.hip)

struct Out {
    std::vector<uint8_t> b;
    bool monty = false;
    void u32(uint32_t v) {
        for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void u64(uint64_t v) {
        for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void f(uint32_t canon) {
        if (canon >= P) throw std::runtime_error("field word is not canonical");  // upstream's deserialiser rejects it
        u32(monty ? (uint32_t)(((uint64_t)canon << 32) % P) : canon);
    }
    void fs(const uint32_t* p, size_t n) {
        for (size_t i = 0; i < n; i++) f(p[i]);
    }
    void str(const char* s) {
        size_t n = strlen(s);
        u64(n);
        b.insert(b.end(), s, s + n);
    }
};

struct Words {
    const uint32_t* p;
    uint64_t n, pos = 0;
    const uint32_t* take(uint64_t k) {
        if (k > n - pos) throw std::runtime_error("truncated proof");
        const uint32_t* r = p + pos;
        pos += k;
        return r;
    }
    uint32_t one() { return *take(1); }
};

struct Chip {
    uint32_t machine_index, log_n, width, prep_width, perm_width, quotient_degree, prep_index_plus1;
    const uint32_t* cumulative_sum;
    const uint32_t *prep[2] = {nullptr, nullptr}, *main[2], *perm[2], *quot;
};

void air_values(Out& o, const uint32_t* const v[2], uint32_t width) {
    for (int k = 0; k < 2; k++) {
        o.u64(v[k] ? width : 0);
        if (v[k]) o.fs(v[k], (size_t)width * 4);
    }
}

void digests(Out& o, const uint32_t* path, uint32_t levels) {
    o.u64(levels);
    o.fs(path, (size_t)levels * 8);
}

void shard_proof(Out& o, const uint32_t* words, uint64_t n_words, int32_t n_names, const char* const* names) {
    Words w{words, n_words};
    const uint32_t* h = w.take(10);
    if (h[0] != PROOF_MAGIC) throw std::runtime_error("not a lurkhip proof (bad magic)");
    const uint32_t n_chips = h[1], log_blowup = h[2], nq = h[3], n_public = h[5], n_layers = h[6], log_max = h[7], n_prep = h[8], n_chunks = h[9];
    std::vector<Chip> chips(n_chips);
    for (auto& c : chips) {
        const uint32_t* q = w.take(7);
        c.machine_index = q[0], c.log_n = q[1], c.width = q[2], c.prep_width = q[3], c.perm_width = q[4], c.quotient_degree = q[5], c.prep_index_plus1 = q[6];
        c.cumulative_sum = w.take(4);
        if (n_names < 0 || c.machine_index >= (uint32_t)n_names) throw std::runtime_error("chip without a name");
    }
    w.take(n_public);
    const uint32_t *main_root = w.take(8), *perm_root = w.take(8), *quot_root = w.take(8);
    for (uint32_t m = 0; m < n_prep; m++) {
        Chip* c = nullptr;
        for (auto& x : chips)
            if (x.prep_index_plus1 == m + 1) c = &x;
        if (!c) throw std::runtime_error("preprocessed matrix without a chip");
        c->prep[0] = w.take((uint64_t)c->prep_width * 4);
        c->prep[1] = w.take((uint64_t)c->prep_width * 4);
    }
    for (auto& c : chips) c.main[0] = w.take((uint64_t)c.width * 4), c.main[1] = w.take((uint64_t)c.width * 4);
    for (auto& c : chips) c.perm[0] = w.take((uint64_t)c.perm_width * 4), c.perm[1] = w.take((uint64_t)c.perm_width * 4);
    uint32_t chunks = 0;
    for (auto& c : chips) {
        c.quot = w.take((uint64_t)c.quotient_degree * 16);
        chunks += c.quotient_degree;
    }
    if (chunks != n_chunks) throw std::runtime_error("quotient chunk count mismatch");
    const uint32_t* fri_roots = w.take((uint64_t)n_layers * 8);
    const uint32_t* final_poly = w.take(4);
    const uint32_t pow_witness = w.one();
    const uint32_t* indices = w.take(nq);
    const uint32_t n_rounds = 3 + (n_prep ? 1 : 0);
    std::vector<uint32_t> rw(n_rounds), lw(n_layers);
    std::vector<const uint32_t*> rrec(n_rounds), lrec(n_layers);
    for (uint32_t r = 0; r < n_rounds; r++) {
        rw[r] = w.one();
        rrec[r] = w.take((uint64_t)nq * rw[r]);
    }
    for (uint32_t l = 0; l < n_layers; l++) {
        lw[l] = w.one();
        lrec[l] = w.take((uint64_t)nq * lw[l]);
    }
    if (w.pos != w.n) throw std::runtime_error("trailing words in proof");

    // commitment
    o.fs(main_root, 8);
    o.fs(perm_root, 8);
    o.fs(quot_root, 8);
    // opened_values
    o.u64(n_chips);
    for (const auto& c : chips) {
        air_values(o, c.prep, c.prep_width);
        air_values(o, c.main, c.width);
        air_values(o, c.perm, c.perm_width);
        o.u64(c.quotient_degree);
        for (uint32_t k = 0; k < c.quotient_degree; k++) {
            o.u64(4);
            o.fs(c.quot + (size_t)k * 16, 16);
        }
        o.fs(c.cumulative_sum, 4);
        o.u64(c.log_n);
    }
    // opening_proof.fri_proof
    o.u64(n_layers);
    o.fs(fri_roots, (size_t)n_layers * 8);
    o.u64(nq);
    for (uint32_t q = 0; q < nq; q++) {
        o.u64(n_layers);
        for (uint32_t l = 0; l < n_layers; l++) {
            const uint32_t log_folded = log_max - 1 - l;
            if (lw[l] != 8 + 8 * log_folded) throw std::runtime_error("layer record size");
            const uint32_t* rec = lrec[l] + (size_t)q * lw[l];
            const uint32_t idx = indices[q] >> l;
            o.fs(rec + 4 * ((idx ^ 1u) & 1u), 4);  // the sibling of the queried element of the pair
            digests(o, rec + 8, log_folded);
        }
    }
    o.fs(final_poly, 4);
    o.f(pow_witness);
    // opening_proof.query_openings[query][round]
    o.u64(nq);
    for (uint32_t q = 0; q < nq; q++) {
        o.u64(n_rounds);
        for (uint32_t r = 0; r < n_rounds; r++) {
            // matrices of the round in committed order: preprocessed (key order), main, permutation, quotient chunks
            std::vector<uint32_t> widths;
            uint32_t log_h_max = 0;
            const uint32_t kind = n_prep ? r : r + 1;  // 0 prep, 1 main, 2 perm, 3 quotient
            if (kind == 0) {
                for (uint32_t m = 0; m < n_prep; m++)
                    for (const auto& c : chips)
                        if (c.prep_index_plus1 == m + 1) widths.push_back(c.prep_width), log_h_max = std::max(log_h_max, c.log_n + log_blowup);
            } else {
                for (const auto& c : chips) {
                    const uint32_t reps = kind == 3 ? c.quotient_degree : 1;
                    for (uint32_t k = 0; k < reps; k++) widths.push_back(kind == 1 ? c.width : kind == 2 ? c.perm_width : 4);
                    log_h_max = std::max(log_h_max, c.log_n + log_blowup);
                }
            }
            uint64_t total = 0;
            for (uint32_t x : widths) total += x;
            if (rw[r] != total + 8 * (uint64_t)log_h_max) throw std::runtime_error("round record size");
            const uint32_t* rec = rrec[r] + (size_t)q * rw[r];
            o.u64(widths.size());
            for (uint32_t x : widths) {
                o.u64(x);
                o.fs(rec, x);
                rec += x;
            }
            digests(o, rec, log_h_max);
        }
    }
    // chip_ordering: name -> position in this shard's (height-sorted) chip list
    o.u64(n_chips);
    for (uint32_t i = 0; i < n_chips; i++) {
        o.str(names[chips[i].machine_index]);
        o.u64(i);
    }
}

int64_t finish(const Out& o, uint8_t* out, uint64_t capacity) {
    if (out && capacity >= o.b.size()) memcpy(out, o.b.data(), o.b.size());
    return (int64_t)o.b.size();
}

void zptr(Out& o, const uint32_t* z) {
    // Tag: a unit-variant enum of 15 variants (U64 = 0 ... Err = 14, /root/reference/src/core/tag.rs:23-39); bincode writes the
    // variant index and rejects any other on the way back in
    if (z[0] > 14) throw std::runtime_error("ZPtr tag out of range");
    o.u32(z[0]);
    o.fs(z + 1, 8);
}

}  // namespace

extern "C" {

int64_t lurkhip_crypto_proof_bincode(int32_t n_shards, const uint32_t* const* shard_words, const uint64_t* shard_n_words, int32_t n_chip_names,
                                     const char* const* chip_names, const char* verifier_version, int32_t serialize_montgomery, uint8_t* out,
                                     uint64_t capacity) {
    if (n_shards < 1 || !shard_words || !shard_n_words || n_chip_names < 1 || !chip_names || !verifier_version) return LURKHIP_ERR_INVALID_ARG;
    try {
        Out o;
        o.monty = serialize_montgomery != 0;
        o.u64((uint64_t)n_shards);
        uint32_t depth = 0;
        for (int32_t s = 0; s < n_shards; s++) {
            if (!shard_words[s]) return LURKHIP_ERR_INVALID_ARG;
            shard_proof(o, shard_words[s], shard_n_words[s], n_chip_names, chip_names);
            // depth = the last DEPTH_W = 4 public values as little-endian bytes (proofs.rs:115-124); the same in every shard
            const uint32_t n_chips = shard_words[s][1], n_public = shard_words[s][5];
            if (n_public < 4) return LURKHIP_ERR_INVALID_ARG;
            const uint32_t* pv = shard_words[s] + 10 + 11 * (uint64_t)n_chips;
            uint32_t d = 0;
            for (int k = 0; k < 4; k++) {
                if (pv[n_public - 4 + k] > 255) return LURKHIP_ERR_INVALID_ARG;  // `assert!(x <= u8::MAX)`
                d |= pv[n_public - 4 + k] << (8 * k);
            }
            if (s && d != depth) return LURKHIP_ERR_INVALID_ARG;  // "all shards have the same public values"
            depth = d;
        }
        o.str(verifier_version);
        o.u32(depth);
        return finish(o, out, capacity);
    } catch (const std::exception&) {
        return LURKHIP_ERR_INVALID_ARG;
    }
}

int64_t lurkhip_shard_proof_bincode(const uint32_t* words, uint64_t n_words, int32_t n_chip_names, const char* const* chip_names,
                                    int32_t serialize_montgomery, uint8_t* out, uint64_t capacity) {
    if (!words || n_chip_names < 1 || !chip_names) return LURKHIP_ERR_INVALID_ARG;
    try {
        Out o;
        o.monty = serialize_montgomery != 0;
        shard_proof(o, words, n_words, n_chip_names, chip_names);
        // sphinx's ShardProof ends with `public_values: Vec<Val>` (the field CryptoShardProof drops, proofs.rs:61-74)
        const uint32_t n_chips = words[1], n_public = words[5];
        const uint32_t* pv = words + 10 + 11 * (uint64_t)n_chips;
        o.u64(n_public);
        o.fs(pv, n_public);
        return finish(o, out, capacity);
    } catch (const std::exception&) {
        return LURKHIP_ERR_INVALID_ARG;
    }
}

int64_t lurkhip_cached_proof_bincode(const uint8_t* crypto_proof, uint64_t crypto_len, const uint32_t* expr, const uint32_t* env,
                                     const uint32_t* result, uint64_t n_dag_entries, const uint32_t* dag_entries, int32_t serialize_montgomery,
                                     uint8_t* out, uint64_t capacity) {
    if (!crypto_proof || !expr || !env || !result || (n_dag_entries && !dag_entries)) return LURKHIP_ERR_INVALID_ARG;
    try {
        Out o;
        o.monty = serialize_montgomery != 0;
        o.b.assign(crypto_proof, crypto_proof + crypto_len);
        zptr(o, expr);
        zptr(o, env);
        zptr(o, result);
        // ZDag(FxHashMap<ZPtr, ZPtrType>): length, then (key, value); ZPtrType = Atom | Tuple11(a, b) | Tuple110(a, b, c)
        o.u64(n_dag_entries);
        for (uint64_t i = 0; i < n_dag_entries; i++) {
            const uint32_t* e = dag_entries + i * 37;
            if (e[9] > 2) return LURKHIP_ERR_INVALID_ARG;
            zptr(o, e);
            o.u32(e[9]);
            for (uint32_t k = 0; k < (e[9] == 0 ? 0u : e[9] + 1); k++) zptr(o, e + 10 + 9 * k);
        }
        return finish(o, out, capacity);
    } catch (const std::exception&) {
        return LURKHIP_ERR_INVALID_ARG;
    }
}

}  // extern "C"
