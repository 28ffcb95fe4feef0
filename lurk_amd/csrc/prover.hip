// The shard prover: host orchestration of the device stages and of the Fiat-Shamir transcript.
//
// Replaces (third-party, source absent from /root/reference; [UPSTREAM-RECALL] sphinx-core @ 8a39b951 (SP1 v1
// lineage) over Plonky3 @ a0b92870; parity unpinned):
//   StarkMachine::setup            -> lurkhip_setup              (commit the preprocessed traces)
//   LocalProver::commit_shards     -> lurkhip_shard_commit       (sort chips by height, commit main traces)
//   LocalProver::prove_shard       -> lurkhip_shard_prove        (permutation traces, quotient, openings, FRI)
// reached from the reference at /root/reference/benches/fib.rs:114-124 and /root/reference/src/core/cli/repl.rs:196.
//
// prove_shard, as restated here:
//   sample 2 permutation challenges; per chip permutation trace; commit; observe; sample alpha;
//   per chip quotient values on 31 * <w_{N << lqd}>, split into 2^lqd chunks over cosets; commit; observe;
//   sample zeta; open {preprocessed, main, permutation} at zeta and zeta * w_N, quotient chunks at zeta:
//     p3 TwoAdicFriPcs::open: alpha_fri = sample; per matrix and point y = p(z) by barycentric evaluation on the
//     low coset; reduced_openings[log_height] += alpha_fri^k (p(x) - y) / (x - z) column by column;
//     FRI commit phase (fold by beta per layer, layers committed as width-2 extension matrices), final constant,
//     proof-of-work grind, query indices, Merkle openings of every round and every layer.
// The transcript is host state (challenger.h); roots, sums and opened values cross the boundary once each.
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <new>
#include <numeric>
#include <stdexcept>

#include "babybear.h"
#include "challenger.h"
#include "commit.h"
#include "ctx.h"
#include "fri.h"
#include "split.h"
#include "stark.h"

using namespace lurkhip;
using bb::ef;

struct lurkhip_challenger {
    Challenger ch;
};

struct lurkhip_pk {
    lurkhip_commitment* commit = nullptr;        // preprocessed traces (may be null when no chip has any)
    std::vector<const uint32_t*> traces;         // natural order, Montgomery, caller-owned
    std::vector<uint32_t> log_heights, widths;
    uint32_t root_m[8] = {};
    int log_blowup = 1;
    SplitEnv split;                              // lurkhip_setup_split: the key's commitment is this rank's part of it
};

struct lurkhip_shard {
    std::vector<lurkhip_air*> airs;              // sorted by height, tallest first (stable)
    std::vector<int> machine_index;              // position in the caller's chip list
    std::vector<uint32_t> log_n;
    std::vector<const uint32_t*> main;           // natural order, Montgomery, caller-owned
    std::vector<uint32_t> main_pitch;            // words between rows of main[i] (>= the chip's width: lurkhip_shard_commit_pitched)
    std::vector<int> prep_index;                 // index into the pk's matrices or -1
    lurkhip_commitment* main_commit = nullptr;
    uint32_t root_m[8] = {};
    int log_blowup = 1;
    SplitEnv split;                              // lurkhip_shard_commit_split: one shard proved by several ranks together
    bool main_row_blocks = false;                // ... main[i] of a cut chip holds this rank's block of rows only
};

struct lurkhip_proof {
    std::vector<uint32_t> words;
};

namespace {

constexpr uint32_t SIDE_LANES_SHORT_PROOF_LOG_N = 17;  // a shard whose tallest chip is below 2^17 rows (the byte chip is always 2^16) spreads its short chips over all side lanes
constexpr int SIDE_LANE_MAX_LOG_N = 13;          // chips below 2^13 rows are "short": their per-chip launches go to the side lane
constexpr uint32_t PROOF_MAGIC = 0x4652504cu;    // "LPRF"
constexpr uint32_t OPENING_MAGIC = 0x4e504f4cu;  // "LOPN"

// dst[4 k ..] = the extension element at src[k]: the chips' cumulative sums, gathered for one read-back
constexpr int GATHER_EF_MAX = 64;
struct GatherEfArgs {
    const uint32_t* src[GATHER_EF_MAX];
    uint32_t n;
};
// the query records leave the device as canonical words: the proof holds canonical values, and converting 300 k words one by one
// while serialising was 0.8 ms of host time at the end of every proof (a fifth of a 2^12-row proof's device time)
__global__ void k_records_canonical(uint32_t* __restrict__ rec, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) rec[i] = bb::from_monty(rec[i]);
}

__global__ void k_gather_ef(GatherEfArgs a, uint32_t* __restrict__ dst) {
    const uint32_t k = threadIdx.x >> 2, j = threadIdx.x & 3;
    if (k < a.n) dst[threadIdx.x] = a.src[k][j];
}

ef ef_pow_host(ef a, uint64_t e) {
    ef r = bb::ef_one();
    while (e) {
        if (e & 1) r = bb::ef_mul(r, a);
        a = bb::ef_sqr(a);
        e >>= 1;
    }
    return r;
}
uint32_t pow_host(uint32_t a_m, uint64_t e) {
    uint32_t r = bb::R1;
    while (e) {
        if (e & 1) r = bb::mul(r, a_m);
        a_m = bb::mul(a_m, a_m);
        e >>= 1;
    }
    return r;
}

void push_ef(std::vector<uint32_t>& out, const ef& e) {
    for (int i = 0; i < 4; i++) out.push_back(bb::from_monty(e.c[i]));
}

struct Round {
    lurkhip_commitment* c;
    // per matrix (committed order): the opening points (indices into the shard's point table)
    std::vector<std::vector<int>> points;
};

}  // namespace

extern "C" {

// ------------------------------------------------------------------ transcript
int32_t lurkhip_challenger_new(lurkhip_ctx* ctx, lurkhip_challenger** out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out != nullptr, "null argument");
    const P16Params* dev = nullptr;
    LH_TRY(get_merkle_params(ctx, &dev));
    auto* c = new lurkhip_challenger();
    c->ch.params = (const P16Params*)ctx->merkle_params_host;
    const lurkhip_protocol_profile& prof = profile_of(ctx);
    c->ch.squeeze = (int)prof.challenger_squeeze;
    c->ch.pop_front = prof.challenger_pop_front != 0;
    *out = c;
    return LURKHIP_OK;
}
int32_t lurkhip_challenger_clone(const lurkhip_challenger* src, lurkhip_challenger** out) {
    if (!src || !out) return LURKHIP_ERR_INVALID_ARG;
    *out = new lurkhip_challenger(*src);
    return LURKHIP_OK;
}
int32_t lurkhip_challenger_free(lurkhip_challenger* c) {
    delete c;
    return LURKHIP_OK;
}
int32_t lurkhip_challenger_observe(lurkhip_challenger* c, const uint32_t* values, uint32_t n) {
    if (!c || (n && !values)) return LURKHIP_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; i++) c->ch.observe(values[i]);
    return LURKHIP_OK;
}
int32_t lurkhip_challenger_sample(lurkhip_challenger* c, uint32_t* out, uint32_t n) {
    if (!c || (n && !out)) return LURKHIP_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; i++) out[i] = bb::from_monty(c->ch.sample_m());
    return LURKHIP_OK;
}
int32_t lurkhip_challenger_sample_bits(lurkhip_challenger* c, uint32_t bits, uint32_t* out) {
    if (!c || !out || bits > 30) return LURKHIP_ERR_INVALID_ARG;
    *out = c->ch.sample_bits((int)bits);
    return LURKHIP_OK;
}

// ------------------------------------------------------------------ setup
int32_t lurkhip_setup(lurkhip_ctx* ctx, int32_t n_prep, const uint32_t* const* prep_traces_dev, const uint32_t* log_heights,
                      const uint32_t* widths, int32_t log_blowup, lurkhip_pk** out, uint32_t* root) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out && n_prep >= 0 && (n_prep == 0 || (prep_traces_dev && log_heights && widths)), "bad setup arguments");
    auto* pk = new lurkhip_pk();
    pk->log_blowup = log_blowup;
    if (n_prep > 0) {
        // sphinx sorts the preprocessed traces by height, tallest first (stable)
        std::vector<int> order(n_prep);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return log_heights[a] > log_heights[b]; });
        for (int i : order) {
            pk->traces.push_back(prep_traces_dev[i]);
            pk->log_heights.push_back(log_heights[i]);
            pk->widths.push_back(widths[i]);
        }
        bool identity = true;
        for (int i = 0; i < n_prep; i++) identity = identity && order[i] == i;
        if (!identity) {
            delete pk;
            return set_error(ctx, LURKHIP_ERR_INVALID_ARG, "pass the preprocessed traces sorted by height, tallest first");
        }
        uint32_t canon[8];
        int32_t s = commit_impl(ctx, n_prep, pk->traces.data(), false, pk->log_heights.data(), pk->widths.data(), log_blowup,
                                LURKHIP_REPR_MONTY, 0, &pk->commit, canon);
        if (s != LURKHIP_OK) {
            delete pk;
            return s;
        }
        memcpy(pk->root_m, canon, sizeof canon);  // repr = MONTY: the root comes back in Montgomery form
    }
    if (root)
        for (int i = 0; i < 8; i++) root[i] = bb::from_monty(pk->root_m[i]);
    *out = pk;
    return LURKHIP_OK;
}

int32_t lurkhip_pk_free(lurkhip_ctx* ctx, lurkhip_pk* pk) {
    LH_CHECK_CTX_NOLOCK(ctx);
    if (!pk) return LURKHIP_OK;
    if (pk->commit) free_commitment(ctx, pk->commit);
    delete pk;
    return LURKHIP_OK;
}

// ------------------------------------------------------------------ main commitment of one shard
int32_t lurkhip_shard_commit(lurkhip_ctx* ctx, int32_t n_chips, lurkhip_air* const* airs, const uint32_t* log_heights,
                             const uint32_t* const* main_traces_dev, const int32_t* prep_indices, int32_t log_blowup,
                             lurkhip_shard** out, uint32_t* root) {
    return lurkhip_shard_commit_pitched(ctx, n_chips, airs, log_heights, main_traces_dev, nullptr, prep_indices, log_blowup, out, root);
}

int32_t lurkhip_shard_commit_pitched(lurkhip_ctx* ctx, int32_t n_chips, lurkhip_air* const* airs, const uint32_t* log_heights,
                                     const uint32_t* const* main_traces_dev, const uint32_t* main_pitches, const int32_t* prep_indices,
                                     int32_t log_blowup, lurkhip_shard** out, uint32_t* root) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, n_chips > 0 && airs && log_heights && main_traces_dev && out, "bad shard arguments");
    if (main_pitches)
        for (int i = 0; i < n_chips; i++)
            LH_ARG(ctx, airs[i] && main_pitches[i] >= air_of(airs[i]).width, "chip %d: row pitch %u below its width", i, main_pitches[i]);
    host_mark("shard_commit in");
    auto* sh = new lurkhip_shard();
    sh->log_blowup = log_blowup;
    std::vector<int> order(n_chips);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return log_heights[a] > log_heights[b]; });
    std::vector<uint32_t> widths;
    for (int i : order) {
        sh->airs.push_back(airs[i]);
        sh->machine_index.push_back(i);
        sh->log_n.push_back(log_heights[i]);
        sh->main.push_back(main_traces_dev[i]);
        sh->prep_index.push_back(prep_indices ? prep_indices[i] : -1);
        widths.push_back(air_of(airs[i]).width);
        sh->main_pitch.push_back(main_pitches ? main_pitches[i] : widths.back());
    }
    span_begin(ctx, "commit_main");
    // LURKHIP_MAIN_SPARSE_LDE=1 (opt-in, measured, not part of the headline): the main traces' identically-zero columns -- selectors and
    // auxiliary columns of never-taken branches, high bytes of small numbers: a third of a real `(fib N)` shard's main cells -- found by
    // one pass over the traces and left out of the LDE like the permutation traces' (the stand-in is sparser than the real
    // functions here, so the number it gives is not the real machine's)
    std::vector<ColumnRuns> main_runs;
    const char* main_sparse = getenv("LURKHIP_MAIN_SPARSE_LDE");
    int32_t s = LURKHIP_OK;
    if (main_sparse && atoi(main_sparse) != 0)
        s = nonzero_column_runs(ctx, n_chips, sh->main.data(), sh->log_n.data(), widths.data(), sh->main_pitch.data(), &main_runs, nullptr);
    if (s == LURKHIP_OK)
        s = commit_impl(ctx, n_chips, sh->main.data(), false, sh->log_n.data(), widths.data(), log_blowup, LURKHIP_REPR_MONTY, 0, &sh->main_commit,
                        sh->root_m, nullptr, false, /*padded_groups=*/true, sh->main_pitch.data(), main_runs.empty() ? nullptr : &main_runs);
    span_end(ctx, "commit_main");
    if (s != LURKHIP_OK) {
        delete sh;
        return s;
    }
    if (root)
        for (int i = 0; i < 8; i++) root[i] = bb::from_monty(sh->root_m[i]);
    *out = sh;
    return LURKHIP_OK;
}

// ------------------------------------------------------------------ one shard over several ranks (include/lurkhip.h; split.hip)
int32_t lurkhip_setup_split(lurkhip_ctx* ctx, const lurkhip_split_comm* comm, int32_t split_min_log_n, int32_t n_prep,
                            const uint32_t* const* prep_traces_dev, const uint32_t* log_heights, const uint32_t* widths, int32_t log_blowup,
                            lurkhip_pk** out, uint32_t* root) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out && n_prep >= 0 && (n_prep == 0 || (prep_traces_dev && log_heights && widths)), "bad setup arguments");
    SplitEnv env;
    LH_TRY(split_env_init(ctx, comm, split_min_log_n, &env));
    auto* pk = new lurkhip_pk();
    pk->log_blowup = log_blowup;
    pk->split = env;
    std::vector<SplitMat> sm;
    for (int i = 0; i < n_prep; i++) {
        if (i && log_heights[i] > log_heights[i - 1]) {
            delete pk;
            return set_error(ctx, LURKHIP_ERR_INVALID_ARG, "pass the preprocessed traces sorted by height, tallest first");
        }
        pk->traces.push_back(prep_traces_dev[i]);
        pk->log_heights.push_back(log_heights[i]);
        pk->widths.push_back(widths[i]);
        sm.push_back(SplitMat{prep_traces_dev[i], log_heights[i], widths[i], widths[i], 0u, split::K_FULL, 0u, 0u, 0u, 0u});
    }
    if (n_prep > 0) {
        const int32_t s = split_commit(ctx, env, n_prep, sm.data(), log_blowup, &pk->commit, pk->root_m);
        if (s != LURKHIP_OK) {
            delete pk;
            return s;
        }
    }
    if (root)
        for (int i = 0; i < 8; i++) root[i] = bb::from_monty(pk->root_m[i]);
    *out = pk;
    return LURKHIP_OK;
}

int32_t lurkhip_shard_commit_split(lurkhip_ctx* ctx, const lurkhip_split_comm* comm, int32_t split_min_log_n, int32_t n_chips,
                                   lurkhip_air* const* airs, const uint32_t* log_heights, const uint32_t* const* main_traces_dev,
                                   const uint32_t* main_pitches, const int32_t* prep_indices, int32_t log_blowup, int32_t main_row_blocks,
                                   lurkhip_shard** out, uint32_t* root) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, n_chips > 0 && airs && log_heights && main_traces_dev && out, "bad shard arguments");
    SplitEnv env;
    LH_TRY(split_env_init(ctx, comm, split_min_log_n, &env));
    for (int i = 0; i < n_chips; i++) {
        LH_ARG(ctx, airs[i] && (!main_pitches || main_pitches[i] >= air_of(airs[i]).width), "chip %d: row pitch below its width", i);
        LH_ARG(ctx, !air_reads_prep_next(airs[i]), "chip %d reads a preprocessed column on the next row: not supported over several ranks", i);
    }
    auto* sh = new lurkhip_shard();
    sh->log_blowup = log_blowup;
    sh->split = env;
    sh->main_row_blocks = main_row_blocks != 0;
    std::vector<int> order(n_chips);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return log_heights[a] > log_heights[b]; });
    std::vector<SplitMat> sm;
    for (int i : order) {
        const lair::ChipAir& air = air_of(airs[i]);
        sh->airs.push_back(airs[i]);
        sh->machine_index.push_back(i);
        sh->log_n.push_back(log_heights[i]);
        sh->main.push_back(main_traces_dev[i]);
        sh->prep_index.push_back(prep_indices ? prep_indices[i] : -1);
        sh->main_pitch.push_back(main_pitches ? main_pitches[i] : air.width);
        // the columns a chip's constraints read on the next row travel to the rank that owns the row (QuotientArgs::next_off)
        const int kind = sh->main_row_blocks && (int)log_heights[i] >= env.min_log_n ? split::K_BLOCK : split::K_FULL;
        sm.push_back(SplitMat{main_traces_dev[i], log_heights[i], air.width, sh->main_pitch.back(), 0u, kind, 0u, 0u, air_next_columns(airs[i]), air.log_quotient_degree()});
    }
    span_begin(ctx, "commit_main");
    const int32_t s = split_commit(ctx, env, n_chips, sm.data(), log_blowup, &sh->main_commit, sh->root_m);
    span_end(ctx, "commit_main");
    if (s != LURKHIP_OK) {
        delete sh;
        return s;
    }
    if (root)
        for (int i = 0; i < 8; i++) root[i] = bb::from_monty(sh->root_m[i]);
    *out = sh;
    return LURKHIP_OK;
}

int32_t lurkhip_shard_free(lurkhip_ctx* ctx, lurkhip_shard* sh) {
    LH_CHECK_CTX_NOLOCK(ctx);
    if (!sh) return LURKHIP_OK;
    if (sh->main_commit) free_commitment(ctx, sh->main_commit);
    delete sh;
    return LURKHIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ p3 TwoAdicFriPcs::open
// Opens committed matrices at their points and proves the openings with FRI: barycentric evaluation on the low coset,
// alpha-batched reduced openings per LDE height, FRI commit phase, proof of work, query openings.  `rounds[r].points[m]` lists
// the (one or two) entries of `pts` matrix m of commitment r is opened at.  Device buffers go to `pooled`, layer commitments to
// `to_free`: the caller releases both (also on failure).
namespace {
struct OpenOut {
    std::vector<std::vector<std::vector<std::vector<ef>>>> opened;  // [round][matrix][point][column], Montgomery
    std::vector<uint32_t> layer_roots_m;
    ef final_poly;
    uint32_t pow_witness = 0;
    std::vector<uint32_t> indices;
    std::vector<uint32_t> round_record_words, layer_record_words;
    std::vector<size_t> round_off, layer_off;  // word offsets into rec_host
    const uint32_t* rec_host = nullptr;        // page-locked staging of the context: read it before the next call on the context
    std::vector<std::vector<uint32_t>> round_records;  // split commitments: round r's records assembled on the host (else empty: rec_host + round_off[r])
    std::vector<std::vector<uint32_t>> layer_records;  // ... and the FRI layers that were folded on the ranks' row blocks
    size_t rec_words = 0;
    int log_max = 0;
    size_t n_layers = 0;
};
}  // namespace

static int32_t pcs_open_impl(lurkhip_ctx* ctx, const lurkhip_protocol_profile& prof, const std::vector<Round>& rounds,
                             const std::vector<ef>& pts, int log_blowup, Challenger& ch, uint32_t num_queries, uint32_t pow_bits,
                             std::vector<lurkhip_commitment*>& to_free, std::vector<void*>& pooled, OpenOut& out, const SplitEnv* sp = nullptr) {
    // One shard over several ranks (sp): a commitment holds this rank's storage rows of every matrix taller than G rows
    // (lurkhip_commitment::is_local).  A polynomial of degree < N is as well interpolated from its 2N values on the whole LDE coset
    // as from the N on the low one, so every rank sums its rows against the weights of the 2N-point domain and the partial sums are
    // all-reduced; the reduced openings are row-local and all-gathered; FRI then runs on every rank.
    auto bary_log = [&](const lurkhip_commitment* c, int m) { return c->is_local(m) ? c->log_h[m] : c->log_h[m] - log_blowup; };
    auto dot_rows = [&](const lurkhip_commitment* c, int m) { return (size_t)1 << (c->is_local(m) ? c->rows_log(m) : c->log_h[m] - log_blowup); };
    auto palloc = [&](size_t bytes, uint32_t** p) -> int32_t {
        void* v = nullptr;
        int32_t s = pool_alloc(ctx, bytes, &v);
        if (s == LURKHIP_OK) {
            pooled.push_back(v);
            *p = (uint32_t*)v;
        }
        return s;
    };
#define PTRY(expr) LH_TRY(expr)
#define PHIP(expr) LH_HIP(ctx, expr)
    // ---- p3 TwoAdicFriPcs::open
    span_begin(ctx, "open");
    uint32_t max_w = 1;
    int log_global_max = 0;
    for (const Round& r : rounds)
        for (int m = 0; m < r.c->n_mats; m++) {
            max_w = std::max(max_w, r.c->width[m]);
            log_global_max = std::max(log_global_max, r.c->log_h[m]);
        }
    // (one shard over several ranks: a rank's block of a matrix has 2^-log_g of its rows, and is short or tall by THAT -- a rank's share of
    // a tall proof is a short proof's work)
    const int lane_shift = log_blowup + (sp ? sp->log_g : 0);
    const int side_lanes_wanted = log_global_max - lane_shift >= (int)SIDE_LANES_SHORT_PROOF_LOG_N ? 1 : (int)lurkhip_ctx::N_SIDE;
    // caches keyed by (log size, point index)
    std::map<std::pair<int, int>, uint32_t*> bary, denoms;
    auto get_weights = [&](std::map<std::pair<int, int>, uint32_t*>& cache, int mode, int log_m, int pt, uint32_t** outp) -> int32_t {
        auto key = std::make_pair(log_m, pt);
        auto it = cache.find(key);
        if (it == cache.end()) {
            uint32_t* buf = nullptr;
            LH_TRY(palloc(((size_t)16) << log_m, &buf));
            LH_TRY(point_weights(ctx, mode, log_m, pts[pt], buf));
            it = cache.emplace(key, buf).first;
        }
        *outp = it->second;
        return LURKHIP_OK;
    };
    uint32_t* ro[32] = {};
    uint64_t num_reduced[32] = {};
    const uint32_t g_m = bb::to_monty(bb::GEN);
    // phase 1: barycentric sums of every matrix at its points, one read-back for all of them
    std::vector<size_t> dot_off;  // word offset of each matrix's [2][w][4] block
    size_t dot_words = 0;
    for (const Round& r : rounds)
        for (int m = 0; m < r.c->n_mats; m++) {
            dot_off.push_back(dot_words);
            dot_words += (size_t)2 * r.c->width[m] * 4;
        }
    uint32_t* dot_out = nullptr;
    PTRY(palloc(dot_words * 4, &dot_out));
    {
        // per-block partial sums of every matrix in one buffer, summed by one launch at the end
        size_t partial_words = 0;
        for (const Round& r : rounds)
            for (int m = 0; m < r.c->n_mats; m++) partial_words += column_dot_partial_words(r.c->width[m], dot_rows(r.c, m));
        uint32_t* partials = nullptr;
        PTRY(palloc(std::max<size_t>(partial_words, 4) * 4, &partials));
        std::vector<DotJob> jobs;
        // every weight table of the opening first, in one launch: the barycentric weights of each (height, point) and the inverse
        // denominators of the reduced openings (they do not depend on alpha_fri: queued here, ahead of the host wait)
        {
            std::vector<WeightJob> wjobs;
            auto want = [&](std::map<std::pair<int, int>, uint32_t*>& cache, int mode, int log_m, int pt) -> int32_t {
                const auto key = std::make_pair(log_m, pt);
                if (cache.count(key)) return LURKHIP_OK;
                uint32_t* buf = nullptr;
                LH_TRY(palloc(((size_t)16) << log_m, &buf));
                cache.emplace(key, buf);
                wjobs.push_back(WeightJob{mode, log_m, pts[pt], buf});
                return LURKHIP_OK;
            };
            for (const Round& r : rounds)
                for (int m = 0; m < r.c->n_mats; m++)
                    for (int pt : r.points[m]) {
                        PTRY(want(bary, 0, bary_log(r.c, m), pt));
                        PTRY(want(denoms, 1, r.c->log_h[m], pt));
                    }
            PTRY(point_weights_batch(ctx, wjobs));
        }
        size_t k = 0, at = 0;
        std::vector<NarrowDot> narrow;  // the slab kernel's matrices: one launch for all of them
        SideLane lane(ctx);  // short wide matrices on the side lane
        lane.want = side_lanes_wanted;
        PTRY(lane.open());
        for (const Round& r : rounds)
            for (int m = 0; m < r.c->n_mats; m++, k++) {
                const int log_n = r.c->log_h[m] - log_blowup;
                const std::vector<int>& mp = r.points[m];
                uint32_t *u0 = nullptr, *u1 = nullptr;
                PTRY(get_weights(bary, 0, bary_log(r.c, m), mp[0], &u0));
                if (mp.size() > 1) PTRY(get_weights(bary, 0, bary_log(r.c, m), mp[1], &u1));
                u0 += r.c->row_base(m) * 4;  // (this rank's rows of the table; 0 unless the commitment is split)
                if (u1) u1 += r.c->row_base(m) * 4;
                const size_t n_rows = dot_rows(r.c, m);
                if (column_dot_is_narrow(r.c->width[m])) {
                    narrow.push_back(NarrowDot{r.c->lde[m], r.c->width[m], n_rows, u0, u1, partials + at, r.c->pitch[m]});
                } else {
                    const auto on_side = lane.on_side(log_n + log_blowup - lane_shift < SIDE_LANE_MAX_LOG_N, (uint32_t)k);
                    PTRY(column_dot_partial(ctx, r.c->lde[m], r.c->width[m], r.c->pitch[m], n_rows, u0, u1, partials + at));
                }
                jobs.push_back(DotJob{partials + at, r.c->width[m], n_rows, (uint32_t)dot_off[k], u1 != nullptr});
                at += column_dot_partial_words(r.c->width[m], n_rows);
            }
        PTRY(column_dot_partial_batch(ctx, narrow));
        PTRY(lane.close());
        PTRY(column_dot_finish(ctx, jobs, dot_out));
    }
    uint32_t* dot_host = nullptr;  // page-locked (host_staging is not asked for again before the opened values are formed below)
    PTRY(host_staging(ctx, dot_words * 4, (void**)&dot_host));
    PHIP(hipMemcpyAsync(dot_host, dot_out, dot_words * 4, hipMemcpyDeviceToHost, ctx->stream));
    PHIP(stream_wait(ctx));
    if (sp) {
        // the ranks' partial sums, as 64-bit lanes of Montgomery words (< p each: G of them cannot overflow), reduced mod p here;
        // a matrix of at most G rows is whole on every rank: only rank 0's sum of it counts
        std::vector<uint64_t> lanes(dot_words);
        size_t k = 0;
        for (const Round& r : rounds)
            for (int m = 0; m < r.c->n_mats; m++, k++) {
                const size_t n = (size_t)2 * r.c->width[m] * 4;
                const bool mine = r.c->is_local(m) || sp->rank == 0;
                for (size_t j = 0; j < n; j++) lanes[dot_off[k] + j] = mine ? dot_host[dot_off[k] + j] : 0u;
            }
        PTRY(split_allreduce_u64_host(ctx, *sp, lanes.data(), lanes.size()));
        for (size_t j = 0; j < dot_words; j++) dot_host[j] = (uint32_t)(lanes[j] % bb::P);
    }
    // phase 2: opened values on the host: y = (z^N - g^N) / (N g^(N-1)) * sum
    // per round, per matrix, per point: ys[c] (Montgomery)
    auto& opened = out.opened;
    opened.assign(rounds.size(), {});
    {
        size_t k = 0;
        // the factor depends on the height and the point only (a ladder, an inversion and an extension ladder: 3 us of this
        // thread per matrix, 66 matrices, while the device waits): once per (height, point)
        std::map<std::pair<int, int>, ef> factors;
        auto factor_of = [&](int log_n, int pt) -> const ef& {
            const auto key = std::make_pair(log_n, pt);
            auto it = factors.find(key);
            if (it == factors.end()) {
                const size_t n = (size_t)1 << log_n;
                const uint32_t gn = pow_host(g_m, n), gn1 = pow_host(g_m, n - 1);
                const uint32_t denom_inv = pow_host(bb::mul(bb::to_monty((uint32_t)(n % bb::P)), gn1), bb::P - 2);
                ef zn = ef_pow_host(pts[pt], n);
                zn.c[0] = bb::sub(zn.c[0], gn);
                it = factors.emplace(key, bb::ef_scale(zn, denom_inv)).first;
            }
            return it->second;
        };
        for (size_t ri = 0; ri < rounds.size(); ri++) {
            const Round& r = rounds[ri];
            opened[ri].resize(r.c->n_mats);
            for (int m = 0; m < r.c->n_mats; m++, k++) {
                const uint32_t w = r.c->width[m];
                const std::vector<int>& mp = r.points[m];
                opened[ri][m].resize(mp.size());
                for (size_t p = 0; p < mp.size(); p++) {
                    const ef factor = factor_of(bary_log(r.c, m), mp[p]);
                    std::vector<ef>& ys = opened[ri][m][p];
                    ys.resize(w);
                    for (uint32_t c = 0; c < w; c++) {
                        const uint32_t* sp = &dot_host[dot_off[k] + ((size_t)p * w + c) * 4];
                        ys[c] = bb::ef_mul(factor, ef{{sp[0], sp[1], sp[2], sp[3]}});
                    }
                }
            }
        }
    }
    // the batching challenge: right after zeta at the pinned revision; the later upstream fix observes every opened value first
    if (prof.observe_openings)
        for (size_t ri = 0; ri < rounds.size(); ri++)
            for (auto& mat : opened[ri])
                for (auto& ys : mat)
                    for (const ef& y : ys) ch.observe_ef_m(y);
    const ef alpha_fri = ch.sample_ef_m();
    uint32_t* alpha_pows = nullptr;  // alpha_fri^c for c < max_w
    PTRY(palloc((size_t)max_w * 16, &alpha_pows));
    PTRY(ef_powers(ctx, alpha_fri.c, alpha_pows, max_w));
    uint32_t* alpha_pows_c = nullptr;  // the same powers, centred, 8 words each (lazy accumulators)
    PTRY(palloc((size_t)max_w * 32, &alpha_pows_c));
    PTRY(ef_powers(ctx, alpha_fri.c, alpha_pows_c, max_w, true));
    // (host copies of the powers, extended as the matrices need them: the widest matrix is not the first to be launched, and
    // the device waits for this thread here)
    std::vector<ef> alpha_pows_host{bb::ef_one()};
    alpha_pows_host.reserve(max_w);
    auto alpha_pows_upto = [&](uint32_t w) {
        while (alpha_pows_host.size() < w) alpha_pows_host.push_back(bb::ef_mul(alpha_pows_host.back(), alpha_fri));
    };
    // phase 3: the reduced openings of every matrix
    // narrow matrices wait per (height, first point) and go out together (fri.hip: k_reduce_openings_narrow)
    std::map<std::pair<int, int>, NarrowArgs> narrow;
    auto flush_narrow = [&](NarrowArgs& g) -> int32_t {
        const int32_t st = reduce_openings_narrow(ctx, g);
        g.n_mats = 0;
        return st;
    };
    // Matrices that are column ranges of a padded group buffer (lurkhip_commitment::pitch != width: the prover's own commitments)
    // wait per (round, group) and go out as ONE launch of the slice kernel: the group buffer is a wide matrix whose row pitch is
    // its width, each matrix a run of column slices with its own alpha offsets and reduced opened values (k_reduce_openings_wide
    // was written for one wide matrix cut into slices; nothing in it asks that the slices belong to the same matrix).
    static const uint32_t group_slice_w = getenv("LURKHIP_REDUCE_SLICE_W") ? (uint32_t)std::max(8, std::min(128, atoi(getenv("LURKHIP_REDUCE_SLICE_W")))) : 64u;
    struct GroupWide {
        WideArgs wa{};
        int log_h = 0;
    };
    std::map<std::pair<size_t, int>, GroupWide> group_wide;
    auto flush_group = [&](GroupWide& g) -> int32_t {
        if (g.wa.n_slices == 0) return LURKHIP_OK;
        const int32_t st = reduce_openings_wide(ctx, g.wa);
        g.wa.n_slices = 0;
        return st;
    };
    // every matrix of four columns and more that is not narrow waits per (height, first point) for the rows kernel (fri.hip:
    // k_reduce_openings_rows); LURKHIP_REDUCE_ROWS=0: the round-3 kernels (quad / stream / slices)
    static const bool rows_on = getenv("LURKHIP_REDUCE_ROWS") == nullptr || atoi(getenv("LURKHIP_REDUCE_ROWS")) != 0;
    std::map<std::pair<int, int>, RowsArgs> rows_groups;
    auto flush_rows = [&](RowsArgs& g) -> int32_t {
        const int32_t st = reduce_openings_rows(ctx, g);
        g.n_mats = 0;
        return st;
    };
    size_t mat_k = 0;
    SideLane ro_lane(ctx);  // the accumulators are per height: a height is one lane
    ro_lane.want = side_lanes_wanted;
    PTRY(ro_lane.open());
    for (size_t ri = 0; ri < rounds.size(); ri++) {
        const Round& r = rounds[ri];
        for (int m = 0; m < r.c->n_mats; m++, mat_k++) {
            const int log_h = r.c->log_h[m];
            const auto on_side = ro_lane.on_side(log_h - lane_shift < SIDE_LANE_MAX_LOG_N, (uint32_t)log_h);  // one accumulator per height
            const uint32_t w = r.c->width[m];
            const std::vector<int>& mp = r.points[m];
            uint32_t *d0 = nullptr, *d1 = nullptr;
            ef reduced_ys[2] = {bb::ef_zero(), bb::ef_zero()};
            alpha_pows_upto(w);
            for (size_t p = 0; p < mp.size(); p++) {
                const std::vector<ef>& ys = opened[ri][m][p];
                for (uint32_t c = 0; c < w; c++) reduced_ys[p] = bb::ef_add(reduced_ys[p], bb::ef_mul(alpha_pows_host[c], ys[c]));
            }
            const uint32_t ro_rows = 1u << r.c->rows_log(m);  // (this rank's rows of the height; all of them unless the commitment is split)
            if (!ro[log_h]) {
                PTRY(palloc((size_t)16 * ro_rows, &ro[log_h]));
                PHIP(hipMemsetAsync(ro[log_h], 0, (size_t)16 * ro_rows, ctx->stream));
            }
            PTRY(get_weights(denoms, 1, log_h, mp[0], &d0));
            if (mp.size() > 1) PTRY(get_weights(denoms, 1, log_h, mp[1], &d1));
            d0 += r.c->row_base(m) * 4;
            if (d1) d1 += r.c->row_base(m) * 4;
            // p3 keeps one alpha-power offset per LDE height (num_reduced[log_height]); fri_alpha_global: one for all heights
            uint64_t& offset = num_reduced[prof.fri_alpha_global ? 0 : log_h];
            const ef apow0 = ef_pow_host(alpha_fri, offset);
            const ef apow1 = ef_pow_host(alpha_fri, offset + w);
            if (rows_on && alpha_pows_c && w > NARROW_MAX_W && w <= ROWS_MAX_W) {
                RowsArgs& g = rows_groups[{log_h, mp[0]}];
                if (g.n_mats && g.d1 && d1 && g.d1 != d1) PTRY(flush_rows(g));  // a different second point: its own launch
                if (g.n_mats == 0) {
                    g = RowsArgs{};
                    g.m_rows = ro_rows;
                    g.alpha_pows = alpha_pows_c;
                    g.d0 = d0;
                    g.ro = ro[log_h];
                }
                if (d1) g.d1 = d1;
                g.m[g.n_mats++] = RowsMat{r.c->lde[m], r.c->pitch[m], w, d1 ? 1u : 0u, reduced_ys[0], reduced_ys[1], apow0, apow1};
                if (g.n_mats == ROWS_MAX_MATS) PTRY(flush_rows(g));
            } else if (r.c->pitch[m] != w && alpha_pows_c && !(rows_on && w <= NARROW_MAX_W)) {
                GroupWide& g = group_wide[{ri, r.c->group[m]}];
                if (g.wa.n_slices && g.wa.d1 != d1) PTRY(flush_group(g));  // a different second point: its own launch
                for (uint32_t c0 = 0; c0 < w; c0 += group_slice_w) {
                    if (g.wa.n_slices == WIDE_MAX_SLICES) PTRY(flush_group(g));  // (the accumulators are additive: a matrix may span launches)
                    if (g.wa.n_slices == 0) {
                        g.wa = WideArgs{};
                        g.wa.mat = r.c->group_base[r.c->group[m]];
                        g.wa.w = r.c->pitch[m];
                        g.wa.m_rows = ro_rows;
                        g.wa.alpha_pows = alpha_pows_c;
                        g.wa.d0 = d0;
                        g.wa.d1 = d1;
                        g.wa.ro = ro[log_h];
                        g.log_h = log_h;
                    }
                    const uint32_t sl = g.wa.n_slices++, n = std::min(group_slice_w, w - c0);
                    g.wa.c0[sl] = r.c->col_start[m] + c0;
                    g.wa.sw[sl] = n;
                    const ef shift = ef_pow_host(alpha_fri, c0);
                    g.wa.apow0[sl] = bb::ef_mul(apow0, shift);
                    g.wa.apow1[sl] = bb::ef_mul(apow1, shift);
                    for (size_t p = 0; p < mp.size(); p++) {
                        ef y = bb::ef_zero();
                        for (uint32_t j = 0; j < n; j++) y = bb::ef_add(y, bb::ef_mul(alpha_pows_host[j], opened[ri][m][p][c0 + j]));
                        (p == 0 ? g.wa.ys0 : g.wa.ys1)[sl] = y;
                    }
                }
            } else if (w <= NARROW_MAX_W && alpha_pows_c) {
                NarrowArgs& g = narrow[{log_h, mp[0]}];
                if (g.n_mats && g.d1 && d1 && g.d1 != d1) PTRY(flush_narrow(g));  // a different second point: its own launch
                if (g.n_mats == 0) g = NarrowArgs{{}, 0, ro_rows, alpha_pows_c, d0, nullptr, ro[log_h]};
                if (d1) g.d1 = d1;  // one second point per height (the first point's successor on that domain)
                g.m[g.n_mats++] = NarrowMat{r.c->lde[m], w, d1 ? 1u : 0u, reduced_ys[0], reduced_ys[1], apow0, apow1, r.c->pitch[m]};
                if (g.n_mats == NARROW_MAX_MATS) PTRY(flush_narrow(g));
            } else if (w > 128 && w <= 128 * WIDE_MAX_SLICES && alpha_pows_c) {
                // column slices of equal width (the last one takes the remainder), each with its own alpha offset and
                // reduced opened values
                WideArgs wa{};
                wa.mat = r.c->lde[m];
                wa.w = w;
                wa.m_rows = ro_rows;
                wa.alpha_pows = alpha_pows_c;
                wa.d0 = d0;
                wa.d1 = d1;
                wa.ro = ro[log_h];
                wa.n_slices = (w + 127) / 128;
                const uint32_t sw = (w + wa.n_slices - 1) / wa.n_slices;
                for (uint32_t sl = 0; sl < wa.n_slices; sl++) {
                    const uint32_t c0 = sl * sw, n = std::min(sw, w - c0);
                    wa.c0[sl] = c0;
                    wa.sw[sl] = n;
                    const ef shift = ef_pow_host(alpha_fri, c0);
                    wa.apow0[sl] = bb::ef_mul(apow0, shift);
                    wa.apow1[sl] = bb::ef_mul(apow1, shift);
                    for (size_t p = 0; p < mp.size(); p++) {
                        ef y = bb::ef_zero();
                        for (uint32_t j = 0; j < n; j++) y = bb::ef_add(y, bb::ef_mul(alpha_pows_host[j], opened[ri][m][p][c0 + j]));
                        (p == 0 ? wa.ys0 : wa.ys1)[sl] = y;
                    }
                }
                PTRY(reduce_openings_wide(ctx, wa));
            } else {
                PTRY(reduce_openings(ctx, r.c->lde[m], w, ro_rows, alpha_pows, alpha_pows_c, d0, d1, reduced_ys[0], reduced_ys[1], apow0, apow1, ro[log_h]));
            }
            offset += (uint64_t)mp.size() * w;
        }
    }
    for (auto& kv : narrow) {
        const auto on_side = ro_lane.on_side(kv.first.first - lane_shift < SIDE_LANE_MAX_LOG_N, (uint32_t)kv.first.first);
        PTRY(flush_narrow(kv.second));
    }
    for (auto& kv : rows_groups) {
        const auto on_side = ro_lane.on_side(kv.first.first - lane_shift < SIDE_LANE_MAX_LOG_N, (uint32_t)kv.first.first);
        PTRY(flush_rows(kv.second));
    }
    for (auto& kv : group_wide) {
        const auto on_side = ro_lane.on_side(kv.second.log_h - lane_shift < SIDE_LANE_MAX_LOG_N, (uint32_t)kv.second.log_h);
        PTRY(flush_group(kv.second));
    }
    PTRY(ro_lane.close());
    // One shard over several ranks: the FIRST FRI layers stay on the ranks' row blocks -- a fold pairs adjacent storage rows, so a
    // block folds into a block; a layer's tree is the ranks' subtrees under the top log2 G levels, its root reaches the (device-side)
    // transcript through one all-gather of 32 bytes per rank and log2 G tiny compression launches, no host round trip -- while a
    // rank's block has at least 2^fri_min_local pairs; then the folded vector is all-gathered and the remaining layers run on every
    // rank.  (With every layer on every rank FRI was the largest part of a proof that did not shrink with the number of ranks.)
    int k_dist = 0;
    if (sp) {
        static const int fri_min_local = getenv("LURKHIP_SPLIT_FRI_MIN_LOG") ? std::max(1, atoi(getenv("LURKHIP_SPLIT_FRI_MIN_LOG"))) : 10;
        k_dist = std::max(0, std::min(log_global_max - log_blowup, log_global_max - sp->log_g - fri_min_local));
        // the reduced openings the layers on every rank add in: the whole vectors; the distributed layers keep their rows of theirs
        for (int lh = sp->log_g + 1; lh < 32; lh++) {
            if (!ro[lh] || (k_dist > 0 && lh >= log_global_max - k_dist)) continue;
            uint32_t* whole = nullptr;
            PTRY(palloc((size_t)16 << lh, &whole));
            PTRY(split_allgather_dev(ctx, *sp, ro[lh], whole, (uint64_t)4 << (lh - sp->log_g)));
            ro[lh] = whole;
        }
    }
    span_end(ctx, "open");

    // ---- FRI commit phase
    span_begin(ctx, "fri_commit");
    const int log_max = log_global_max;
    std::vector<lurkhip_commitment*> layers;
    std::vector<uint32_t>& layer_roots_m = out.layer_roots_m;
    uint32_t* current = ro[log_max];
    // The transcript moves to the device for this phase (fri.hip: k_fri_challenge): per layer "commit, observe the root,
    // sample beta, fold" is a chain of launches with no host round trip; the state comes back with the final polynomial.
    const int n_layers = log_max > log_blowup ? log_max - log_blowup : 0;
    DevChallenger hc{};
    memcpy(hc.state, ch.state, sizeof hc.state);
    hc.n_in = (uint32_t)ch.input.size();
    hc.n_out = (uint32_t)ch.output.size();
    hc.out_head = 0;
    hc.squeeze = (uint32_t)ch.squeeze;
    hc.pop_front = ch.pop_front ? 1u : 0u;
    for (size_t i = 0; i < ch.input.size(); i++) hc.input[i] = ch.input[i];
    for (size_t i = 0; i < ch.output.size(); i++) hc.output[i] = ch.output[i];
    DevChallenger* ch_dev = nullptr;
    uint32_t* betas_dev = nullptr;
    PTRY(palloc(sizeof(DevChallenger), (uint32_t**)&ch_dev));
    PTRY(palloc((size_t)std::max(n_layers, 1) * 16, &betas_dev));
    uint32_t* roots_dev = nullptr;  // the layer roots, copied by k_fri_challenge as it observes them
    PTRY(palloc((size_t)std::max(n_layers, 1) * 32, &roots_dev));
    static_assert(sizeof(DevChallenger) % 4 == 0, "uploaded as words");
    PTRY(upload_words(ctx, (uint32_t*)ch_dev, (const uint32_t*)&hc, sizeof hc / 4));  // as launch arguments: no host wait
    struct DistLayer {
        lurkhip_commitment* local;  // the tree over this rank's pairs
        uint32_t* top_dev;          // the top of the tree: [G][8] subtree roots, [G / 2][8], .. , [1][8]
        int log_pairs;              // log2 of the layer's pairs (all ranks)
    };
    std::vector<DistLayer> dist_layers;
    for (int log_folded = log_max - 1, li = 0; log_folded >= log_blowup; log_folded--, li++) {
        lurkhip_commitment* lc = nullptr;
        if (li < k_dist) {
            const int log_local = log_folded - sp->log_g, G = sp->world();
            PTRY(commit_raw(ctx, {current}, {log_local}, {8u}, &lc));
            to_free.push_back(lc);
            uint32_t* top = nullptr;
            PTRY(palloc((size_t)(2 * G) * 32, &top));
            PTRY(split_allgather_dev(ctx, *sp, lc->digests + lc->level_off[(size_t)lc->log_max] * 8, top, 8));
            const P16Params* mp = nullptr;
            PTRY(get_merkle_params(ctx, &mp));
            uint32_t* level = top;
            for (int nodes = G >> 1; nodes >= 1; nodes >>= 1) {
                PTRY(merkle_level_digests(ctx, mp, level, (size_t)nodes, nullptr, level + (size_t)nodes * 16));
                level += (size_t)nodes * 16;
            }
            PTRY(fri_challenge(ctx, ch_dev, level, betas_dev + 4 * li, roots_dev + 8 * li));
            uint32_t* next = nullptr;
            PTRY(palloc((size_t)16 << log_local, &next));
            // (the layer's reduced openings: this rank's rows of them, or none at that height)
            PTRY(fri_fold(ctx, current, log_folded + 1, betas_dev + 4 * li, ro[log_folded], next, (uint32_t)sp->rank << log_local, 1u << log_local));
            current = next;
            dist_layers.push_back(DistLayer{lc, top, log_folded});
            layers.push_back(lc);
            if (li + 1 == k_dist) {  // from here on every rank holds the whole vector
                uint32_t* whole = nullptr;
                PTRY(palloc((size_t)16 << log_folded, &whole));
                PTRY(split_allgather_dev(ctx, *sp, current, whole, (uint64_t)4 << log_local));
                current = whole;
            }
            continue;
        }
        PTRY(commit_raw(ctx, {current}, {log_folded}, {8u}, &lc));
        to_free.push_back(lc);
        layers.push_back(lc);
        const uint32_t* root_dev = lc->digests + lc->level_off[lc->log_max] * 8;
        PTRY(fri_challenge(ctx, ch_dev, root_dev, betas_dev + 4 * li, roots_dev + 8 * li));
        uint32_t* next = nullptr;
        PTRY(palloc(((size_t)16) << log_folded, &next));
        PTRY(fri_fold(ctx, current, log_folded + 1, betas_dev + 4 * li, ro[log_folded], next));
        current = next;
    }
    std::vector<uint32_t> fin((size_t)4 << log_blowup);
    layer_roots_m.resize((size_t)layers.size() * 8);
    {
        // one page-locked block: [roots | transcript state | final polynomial]
        const size_t b_roots = layer_roots_m.size() * 4, o_hc = (b_roots + 15) & ~(size_t)15, o_fin = o_hc + ((sizeof hc + 15) & ~(size_t)15);
        uint8_t* st = nullptr;
        PTRY(host_staging(ctx, o_fin + fin.size() * 4, (void**)&st));
        if (b_roots) PHIP(hipMemcpyAsync(st, roots_dev, b_roots, hipMemcpyDeviceToHost, ctx->stream));
        PHIP(hipMemcpyAsync(st + o_hc, ch_dev, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
        PHIP(hipMemcpyAsync(st + o_fin, current, fin.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHIP(stream_wait(ctx));
        memcpy(layer_roots_m.data(), st, b_roots);
        memcpy(&hc, st + o_hc, sizeof hc);
        memcpy(fin.data(), st + o_fin, fin.size() * 4);
    }
    memcpy(ch.state, hc.state, sizeof hc.state);
    ch.input.assign(hc.input, hc.input + hc.n_in);
    ch.output.assign(hc.output + hc.out_head, hc.output + hc.out_head + hc.n_out);
    for (size_t i = 4; i < fin.size(); i++)
        if (fin[i] != fin[i & 3]) {
            return set_error(ctx, LURKHIP_ERR_EXEC, "internal error: the FRI commit phase did not end on a constant");
        }
    const ef final_poly{{fin[0], fin[1], fin[2], fin[3]}};
    out.final_poly = final_poly;
    ch.observe_ef_m(final_poly);
    span_end(ctx, "fri_commit");

    // ---- proof of work, query indices
    span_begin(ctx, "fri_query");
    uint32_t& pow_witness = out.pow_witness;
    pow_witness = 0;
    if (pow_bits > 0) {
        uint32_t st[16];
        memcpy(st, ch.state, sizeof st);
        for (size_t i = 0; i < ch.input.size(); i++) st[i] = ch.input[i];
        PTRY(pow_grind(ctx, st, (int)ch.input.size(), (int)pow_bits, ch.first_sample_lane(), &pow_witness));
    }
    if (!ch.check_witness((int)pow_bits, pow_witness)) {
        return set_error(ctx, LURKHIP_ERR_EXEC, "proof-of-work witness rejected by the host transcript");
    }
    std::vector<uint32_t>& indices = out.indices;
    indices.assign(num_queries, 0);
    for (auto& ix : indices) ix = ch.sample_bits(log_max);
    uint32_t* indices_dev = nullptr;
    PTRY(palloc((size_t)num_queries * 4, &indices_dev));
    PTRY(upload_words(ctx, indices_dev, indices.data(), (size_t)num_queries));
    // every record of every round and layer goes to one device buffer and comes back in one copy
    auto& round_record_words = out.round_record_words;
    auto& layer_record_words = out.layer_record_words;
    round_record_words.assign(rounds.size(), 0);
    layer_record_words.assign(layers.size(), 0);
    std::vector<std::vector<OpenMat>> round_mats(rounds.size());
    size_t rec_words = 0;
    // A split commitment answers a query from the rank that owns its storage row: rows of the local matrices and the path inside
    // the rank's subtree; the rows of the matrices of at most G rows and the top log2 G siblings are on every rank's host.
    struct SplitRound {
        std::vector<std::vector<uint32_t>> owned;  // per rank: the queries it answers
        uint32_t local_words = 0;                  // words of a local record: local rows | local path
        uint32_t local_rows_words = 0;
        uint32_t* indices_dev = nullptr;
    };
    std::vector<SplitRound> split_rounds(rounds.size());
    for (size_t ri = 0; ri < rounds.size(); ri++) {
        const lurkhip_commitment* c = rounds[ri].c;
        if (c->split_log_g) {
            SplitRound& sr = split_rounds[ri];
            const uint32_t log_local = (uint32_t)(c->log_max - c->split_log_g);
            uint32_t all_w = 0;
            for (int m = 0; m < c->n_mats; m++) {
                all_w += c->width[m];
                if (c->is_local(m)) round_mats[ri].push_back(OpenMat{c->lde[m], c->width[m], (uint32_t)c->rows_log(m), c->pitch[m]});
            }
            PTRY(gather_openings(ctx, round_mats[ri], c->digests, c->level_off, log_local, nullptr, num_queries, 0, nullptr, &sr.local_words));
            sr.local_rows_words = sr.local_words - 8 * log_local;
            round_record_words[ri] = all_w + 8 * (uint32_t)c->log_max;
            sr.owned.assign((size_t)1 << c->split_log_g, {});
            std::vector<uint32_t> mine;
            for (uint32_t q = 0; q < num_queries; q++) {
                const uint32_t ix = indices[q] >> (log_max - c->log_max);
                sr.owned[ix >> log_local].push_back(q);
                if ((int)(ix >> log_local) == c->split_rank) mine.push_back(ix & ((1u << log_local) - 1u));
            }
            if (!mine.empty()) {
                PTRY(palloc(mine.size() * 4, &sr.indices_dev));
                PTRY(upload_words(ctx, sr.indices_dev, mine.data(), mine.size()));
            }
            rec_words += mine.size() * sr.local_words;
            continue;
        }
        for (int m = 0; m < c->n_mats; m++) round_mats[ri].push_back(OpenMat{c->lde[m], c->width[m], (uint32_t)c->log_h[m], c->pitch[m]});
        PTRY(gather_openings(ctx, round_mats[ri], c->digests, c->level_off, (uint32_t)c->log_max, nullptr, num_queries, 0, nullptr, &round_record_words[ri]));
        rec_words += (size_t)num_queries * round_record_words[ri];
    }
    std::vector<SplitRound> split_layers(dist_layers.size());  // (same bookkeeping as a split commitment's round: who answers which query)
    for (size_t li = 0; li < layers.size(); li++) {
        const lurkhip_commitment* c = layers[li];
        if (li < dist_layers.size()) {
            SplitRound& sl = split_layers[li];
            const uint32_t log_local = (uint32_t)c->log_h[0];
            PTRY(gather_openings(ctx, {OpenMat{c->lde[0], 8, log_local}}, c->digests, c->level_off, log_local, nullptr, num_queries, 0, nullptr, &sl.local_words));
            sl.local_rows_words = 8;
            layer_record_words[li] = 8 + 8 * (uint32_t)dist_layers[li].log_pairs;
            sl.owned.assign((size_t)sp->world(), {});
            std::vector<uint32_t> mine;
            for (uint32_t q = 0; q < num_queries; q++) {
                const uint32_t pair = indices[q] >> (li + 1);
                sl.owned[pair >> log_local].push_back(q);
                if ((int)(pair >> log_local) == sp->rank) mine.push_back(pair & ((1u << log_local) - 1u));
            }
            if (!mine.empty()) {
                PTRY(palloc(mine.size() * 4, &sl.indices_dev));
                PTRY(upload_words(ctx, sl.indices_dev, mine.data(), mine.size()));
            }
            rec_words += mine.size() * sl.local_words;
            continue;
        }
        PTRY(gather_openings(ctx, {OpenMat{c->lde[0], 8, (uint32_t)c->log_h[0]}}, c->digests, c->level_off, (uint32_t)c->log_max, nullptr, num_queries, 0,
                             nullptr, &layer_record_words[li]));
        rec_words += (size_t)num_queries * layer_record_words[li];
    }
    uint32_t* rec_dev = nullptr;
    PTRY(palloc(std::max<size_t>(rec_words, 4) * 4, &rec_dev));
    auto& round_off = out.round_off;
    auto& layer_off = out.layer_off;
    round_off.assign(rounds.size(), 0);
    layer_off.assign(layers.size(), 0);
    size_t rec_at = 0;
    for (size_t ri = 0; ri < rounds.size(); ri++) {
        const lurkhip_commitment* c = rounds[ri].c;
        uint32_t rw = 0;
        round_off[ri] = rec_at;
        if (c->split_log_g) {
            const SplitRound& sr = split_rounds[ri];
            const uint32_t n_mine = (uint32_t)sr.owned[(size_t)c->split_rank].size();
            if (n_mine) PTRY(gather_openings(ctx, round_mats[ri], c->digests, c->level_off, (uint32_t)(c->log_max - c->split_log_g), sr.indices_dev, n_mine, 0, rec_dev + rec_at, &rw));
            rec_at += (size_t)n_mine * sr.local_words;
            continue;
        }
        PTRY(gather_openings(ctx, round_mats[ri], c->digests, c->level_off, (uint32_t)c->log_max, indices_dev, num_queries, (uint32_t)(log_max - c->log_max),
                             rec_dev + rec_at, &rw));
        rec_at += (size_t)num_queries * rw;
    }
    for (size_t li = 0; li < layers.size(); li++) {
        const lurkhip_commitment* c = layers[li];
        uint32_t rw = 0;
        layer_off[li] = rec_at;
        if (li < dist_layers.size()) {
            const SplitRound& sl = split_layers[li];
            const uint32_t n_mine = (uint32_t)sl.owned[(size_t)sp->rank].size();
            if (n_mine)
                PTRY(gather_openings(ctx, {OpenMat{c->lde[0], 8, (uint32_t)c->log_h[0]}}, c->digests, c->level_off, (uint32_t)c->log_max, sl.indices_dev, n_mine, 0,
                                     rec_dev + rec_at, &rw));
            rec_at += (size_t)n_mine * sl.local_words;
            continue;
        }
        // index_i = index >> li, pair = index_i >> 1
        PTRY(gather_openings(ctx, {OpenMat{c->lde[0], 8, (uint32_t)c->log_h[0]}}, c->digests, c->level_off, (uint32_t)c->log_max, indices_dev, num_queries,
                             (uint32_t)li + 1, rec_dev + rec_at, &rw));
        rec_at += (size_t)num_queries * rw;
    }
    const uint32_t* rec_host = nullptr;
    PTRY(host_staging(ctx, std::max<size_t>(rec_words, 4) * 4, (void**)&rec_host));
    if (rec_words) {
        hipLaunchKernelGGL(k_records_canonical, dim3((unsigned)std::min<size_t>((rec_words + 255) / 256, 2048)), dim3(256), 0, ctx->stream, rec_dev, rec_words);
        PHIP(hipMemcpyAsync((void*)rec_host, rec_dev, rec_words * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    std::vector<std::vector<uint32_t>> tops_m(dist_layers.size());  // the distributed layers' top levels (Montgomery), for the paths
    for (size_t li = 0; li < dist_layers.size(); li++) {
        tops_m[li].resize((size_t)(2 * sp->world() - 1) * 8);
        PHIP(hipMemcpyAsync(tops_m[li].data(), dist_layers[li].top_dev, tops_m[li].size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    PHIP(stream_wait(ctx));
    out.layer_records.assign(layers.size(), {});
    // The owners' answers of EVERY distributed layer and split round in ONE all-gather (a dozen 50 us collectives before): every rank
    // knows who answers what, so the segments -- padded to the largest count of any rank -- lie at the same offsets for all.
    std::vector<size_t> layer_seg(dist_layers.size(), 0), layer_at(dist_layers.size(), 0), round_seg(rounds.size(), 0), round_at(rounds.size(), 0);
    size_t answers_words = 0;
    std::vector<uint32_t> answers_mine, answers_all;
    if (sp) {
        auto seg_of = [](const SplitRound& r) {
            size_t most = 0;
            for (const auto& o : r.owned) most = std::max(most, o.size());
            return most * r.local_words;
        };
        for (size_t li = 0; li < dist_layers.size(); li++) {
            layer_seg[li] = seg_of(split_layers[li]);
            layer_at[li] = answers_words;
            answers_words += layer_seg[li];
        }
        for (size_t ri = 0; ri < rounds.size(); ri++) {
            if (!rounds[ri].c->split_log_g) continue;
            round_seg[ri] = seg_of(split_rounds[ri]);
            round_at[ri] = answers_words;
            answers_words += round_seg[ri];
        }
        answers_words = std::max<size_t>(answers_words, 1);
        answers_mine.assign(answers_words, 0u);
        answers_all.resize(answers_words * (size_t)sp->world());
        for (size_t li = 0; li < dist_layers.size(); li++) {
            const size_t n_mine = split_layers[li].owned[(size_t)sp->rank].size();
            if (n_mine) memcpy(&answers_mine[layer_at[li]], rec_host + layer_off[li], n_mine * split_layers[li].local_words * 4);
        }
        for (size_t ri = 0; ri < rounds.size(); ri++) {
            if (!rounds[ri].c->split_log_g) continue;
            const size_t n_mine = split_rounds[ri].owned[(size_t)sp->rank].size();
            if (n_mine) memcpy(&answers_mine[round_at[ri]], rec_host + round_off[ri], n_mine * split_rounds[ri].local_words * 4);
        }
        PTRY(split_allgather_host(ctx, *sp, answers_mine.data(), answers_all.data(), answers_words * 4));
    }
    for (size_t li = 0; li < dist_layers.size(); li++) {
        const SplitRound& sl = split_layers[li];
        const int G = sp->world();
        const uint32_t log_local = (uint32_t)layers[li]->log_h[0];
        std::vector<uint32_t>& recs = out.layer_records[li];
        recs.resize((size_t)num_queries * layer_record_words[li]);
        for (int r = 0; r < G; r++)
            for (size_t k = 0; k < sl.owned[(size_t)r].size(); k++) {
                const uint32_t q = sl.owned[(size_t)r][k];
                const uint32_t pair = indices[q] >> (li + 1);
                uint32_t* o = &recs[(size_t)q * layer_record_words[li]];
                memcpy(o, &answers_all[(size_t)r * answers_words + layer_at[li] + k * sl.local_words], (size_t)sl.local_words * 4);  // the pair | the path inside the owner's subtree
                o += sl.local_words;
                size_t level = 0;  // word offset of top level t: G, G / 2, .. nodes of 8 words
                for (int t = 0, nodes = G; t < sp->log_g; t++, nodes >>= 1) {
                    const uint32_t* sib = &tops_m[li][level + (size_t)(((pair >> log_local) >> t) ^ 1u) * 8];
                    for (int j = 0; j < 8; j++) *o++ = bb::from_monty(sib[j]);
                    level += (size_t)nodes * 8;
                }
            }
    }
    // the split rounds' records: every rank's answers gathered (padded to the largest count), then assembled query by query
    out.round_records.assign(rounds.size(), {});
    for (size_t ri = 0; ri < rounds.size() && sp; ri++) {
        const lurkhip_commitment* c = rounds[ri].c;
        if (!c->split_log_g) continue;
        const SplitRound& sr = split_rounds[ri];
        const int G = 1 << c->split_log_g;
        const uint32_t log_local = (uint32_t)(c->log_max - c->split_log_g);
        std::vector<uint32_t>& recs = out.round_records[ri];
        recs.resize((size_t)num_queries * round_record_words[ri]);
        for (int r = 0; r < G; r++)
            for (size_t k = 0; k < sr.owned[(size_t)r].size(); k++) {
                const uint32_t q = sr.owned[(size_t)r][k];
                const uint32_t ix = indices[q] >> (log_max - c->log_max);
                const uint32_t* loc = &answers_all[(size_t)r * answers_words + round_at[ri] + k * sr.local_words];
                uint32_t* o = &recs[(size_t)q * round_record_words[ri]];
                memcpy(o, loc, (size_t)sr.local_rows_words * 4);
                o += sr.local_rows_words;
                for (int m = 0; m < c->n_mats; m++) {  // (after every local matrix in the committed order: the tree's heights descend)
                    if (c->is_local(m)) continue;
                    const uint32_t* row = c->tiny_rows_m[(size_t)m].data() + (size_t)(ix >> (c->log_max - c->log_h[m])) * c->width[m];
                    for (uint32_t j = 0; j < c->width[m]; j++) *o++ = bb::from_monty(row[j]);
                }
                memcpy(o, loc + sr.local_rows_words, (size_t)8 * log_local * 4);
                o += 8 * log_local;
                for (int t = 0; t < c->split_log_g; t++) {
                    const uint32_t* sib = &c->top_levels_m[(size_t)t][(size_t)(((ix >> log_local) >> t) ^ 1u) * 8];
                    for (int j = 0; j < 8; j++) *o++ = bb::from_monty(sib[j]);
                }
            }
    }
    span_end(ctx, "fri_query");
    out.rec_host = rec_host;
    out.rec_words = rec_words;
    out.log_max = log_max;
    out.n_layers = layers.size();
    return LURKHIP_OK;

#undef PTRY
#undef PHIP
}

extern "C" {

// ------------------------------------------------------------------ prove_shard
static int32_t shard_prove_impl(lurkhip_ctx* ctx, const lurkhip_pk* pk, lurkhip_shard* sh, lurkhip_challenger* chal,
                                const uint32_t* public_values, uint32_t n_public, uint32_t num_queries, uint32_t pow_bits,
                                lurkhip_proof** out);

int32_t lurkhip_shard_prove(lurkhip_ctx* ctx, const lurkhip_pk* pk, lurkhip_shard* sh, lurkhip_challenger* chal,
                            const uint32_t* public_values, uint32_t n_public, uint32_t num_queries, uint32_t pow_bits,
                            lurkhip_proof** out) {
    LH_CHECK_CTX(ctx);
    try {  // nothing unwinds across the C boundary
        host_mark("shard_prove in");
        const int32_t st = shard_prove_impl(ctx, pk, sh, chal, public_values, n_public, num_queries, pow_bits, out);
        host_mark("shard_prove out");
        return st;
    } catch (const std::bad_alloc&) {
        return set_error(ctx, LURKHIP_ERR_OOM, "host allocation failed while proving");
    } catch (const std::exception& e) {
        return set_error(ctx, LURKHIP_ERR_EXEC, "internal error while proving: %s", e.what());
    }
}

int32_t lurkhip_prover_stats(lurkhip_ctx* ctx, uint64_t* out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out != nullptr, "null argument");
    out[0] = ctx->perm_cells;
    out[1] = ctx->perm_cells_transformed;
    return LURKHIP_OK;
}

int32_t lurkhip_shard_prove_split(lurkhip_ctx* ctx, const lurkhip_pk* pk, lurkhip_shard* sh, lurkhip_challenger* chal,
                                  const uint32_t* public_values, uint32_t n_public, uint32_t num_queries, uint32_t pow_bits,
                                  lurkhip_proof** out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, sh && sh->split.on(), "not a shard of lurkhip_shard_commit_split");
    return lurkhip_shard_prove(ctx, pk, sh, chal, public_values, n_public, num_queries, pow_bits, out);
}

static int32_t shard_prove_impl(lurkhip_ctx* ctx, const lurkhip_pk* pk, lurkhip_shard* sh, lurkhip_challenger* chal,
                                const uint32_t* public_values, uint32_t n_public, uint32_t num_queries, uint32_t pow_bits,
                                lurkhip_proof** out) {
    LH_ARG(ctx, pk && sh && chal && out && (n_public == 0 || public_values), "null argument");
    LH_ARG(ctx, num_queries >= 1 && num_queries <= 1024 && pow_bits <= 30, "bad FRI parameters");
    LH_ARG(ctx, pk->log_blowup == sh->log_blowup, "the key was set up with blow-up 2^%d, the shard committed with 2^%d", pk->log_blowup, sh->log_blowup);
    for (size_t i = 0; i < sh->airs.size(); i++) {
        const lair::ChipAir& air = air_of(sh->airs[i]);
        LH_ARG(ctx, n_public >= air.num_public, "chip %s reads %u public values, %u given", air.name.c_str(), air.num_public, n_public);
        const int pi = sh->prep_index[i];
        LH_ARG(ctx, pi < 0 ? air.prep_width == 0 : (pk->commit != nullptr && (size_t)pi < pk->traces.size()),
               "chip %s: preprocessed trace index %d does not exist in the key", air.name.c_str(), pi);
        if (pi >= 0)
            LH_ARG(ctx, pk->log_heights[pi] == sh->log_n[i] && pk->widths[pi] == air.prep_width,
                   "chip %s: its preprocessed trace in the key has another shape", air.name.c_str());
    }
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const lurkhip_protocol_profile prof = profile_of(ctx);
    Challenger& ch = chal->ch;
    const int n_chips = (int)sh->airs.size();
    // One shard over several ranks (lurkhip_shard_commit_split): chips of at least 2^min_log_n rows are cut -- this rank computes
    // their permutation rows for its block of trace rows and their quotient values for its block of LDE storage rows --, the shorter
    // ones are proved whole by every rank; commitments are made by all ranks together (split.hip).  Every rank ends with the same words.
    const SplitEnv* sp = sh->split.on() ? &sh->split : nullptr;
    LH_ARG(ctx, pk->split.log_g == sh->split.log_g && pk->split.rank == sh->split.rank && pk->split.min_log_n == sh->split.min_log_n,
           "the key and the shard were not made by the same set of ranks");
    auto cut = [&](int i) { return sp != nullptr && (int)sh->log_n[i] >= sp->min_log_n; };
    const int log_blowup = sh->log_blowup;
    std::vector<lurkhip_commitment*> to_free;
    std::vector<void*> pooled;
    auto cleanup = [&]() {
        (void)stream_wait(ctx);
        for (auto* c : to_free) free_commitment(ctx, c);
        for (void* p : pooled) pool_release(ctx, p);
    };
    auto palloc = [&](size_t bytes, uint32_t** p) -> int32_t {
        void* v = nullptr;
        int32_t s = pool_alloc(ctx, bytes, &v);
        if (s == LURKHIP_OK) {
            pooled.push_back(v);
            *p = (uint32_t*)v;
        }
        return s;
    };
#define PTRY(expr)                  \
    do {                            \
        int32_t s__ = (expr);       \
        if (s__ != LURKHIP_OK) {    \
            cleanup();              \
            return s__;             \
        }                           \
    } while (0)
#define PHIP(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            cleanup();                                                                              \
            return set_error(ctx, LURKHIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
        }                                                                                           \
    } while (0)

    // ---- permutation traces
    if (prof.observe_chip_meta)  // hardened transcript: bind every chip's shape before any per-shard challenge is drawn
        for (int i = 0; i < n_chips; i++) {
            ch.observe(sh->log_n[i]);
            ch.observe(air_of(sh->airs[i]).width);
            ch.observe((uint32_t)(sh->prep_index[i] + 1));
        }
    const ef perm_alpha = ch.sample_ef_m(), perm_beta = ch.sample_ef_m();
    std::vector<uint32_t*> perm(n_chips, nullptr);
    std::vector<ef> cumsum(n_chips);
    std::vector<uint32_t> perm_widths(n_chips), lqds(n_chips);
    span_begin(ctx, "permutation");
    // one table of beta powers for the proof (every chip reads a prefix) and every chip's interaction start values, written by
    // its permutation trace and read again by its quotient: two launches per chip in the permutation stage and three in the
    // quotient stage fewer
    uint32_t* beta_pows = nullptr;
    std::vector<uint32_t*> chip_starts(n_chips, nullptr);
    {
        uint32_t n_bp = 1;
        size_t start_words = 0;
        for (int i = 0; i < n_chips; i++) {
            n_bp = std::max(n_bp, air_beta_pows(sh->airs[i]));
            start_words += (size_t)std::max(air_num_interactions(sh->airs[i]), 1u) * 4;
        }
        PTRY(palloc((size_t)n_bp * 32, &beta_pows));
        PTRY(ef_powers(ctx, perm_beta.c, beta_pows, n_bp, true));
        uint32_t* all_starts = nullptr;
        PTRY(palloc(start_words * 4, &all_starts));
        size_t at = 0;
        for (int i = 0; i < n_chips; i++) {
            chip_starts[i] = all_starts + at;
            at += (size_t)std::max(air_num_interactions(sh->airs[i]), 1u) * 4;
        }
    }
    // the short chips' launches (a few workgroups each: starts, interpreter rows, one-block scan) go to the side lane, under the
    // tall chips' kernels (short by the rows THIS rank works on: a cut chip's block)
    auto lane_log = [&](int i) { return (int)sh->log_n[i] - (cut(i) ? sp->log_g : 0); };
    SideLane lane(ctx);
    {
        int tallest = 0;
        for (int i = 0; i < n_chips; i++) tallest = std::max(tallest, lane_log(i));
        lane.want = tallest >= (int)SIDE_LANES_SHORT_PROOF_LOG_N ? 1 : lurkhip_ctx::N_SIDE;
    }
    // Round 5: which batch columns does anybody compute?  (stark_kernels.h: PermSink -- a compiled permutation kernel skips a batch
    // whose multiplicities are zero on all 64 rows of a wave; a column no wave marks is identically zero.)  One word per column,
    // zeroed before the lanes fork, read back with the cumulative sums: the LDE of the permutation traces then leaves the dead
    // columns out (commit_impl: live_runs) -- 42 % of the permutation cells of a real `(fib N)` shard.  LURKHIP_PERM_SPARSE_LDE=0: off.
    // (read per proof: a test switches it; LURKHIP_PERM_SPARSE_MIN_CELLS lowers the threshold below so that mid-sized test machines take the route)
    const char* sparse_env = getenv("LURKHIP_PERM_SPARSE_LDE");
    const bool sparse_lde = sparse_env == nullptr || atoi(sparse_env) != 0;  // (split: a column is dead when every rank says so -- the flags travel with the blocks' totals below)
    const char* min_cells_env = getenv("LURKHIP_PERM_SPARSE_MIN_CELLS");
    const uint64_t min_cells = min_cells_env ? (uint64_t)strtoull(min_cells_env, nullptr, 10) : ((uint64_t)1 << 22);
    uint32_t* live_dev = nullptr;
    std::vector<size_t> live_off(n_chips + 1, 0);
    for (int i = 0; i < n_chips; i++) live_off[i + 1] = live_off[i] + air_of(sh->airs[i]).permutation_width();
    // (an LDE of up to 2^10 rows is one kernel: nothing to leave out; and a small proof keeps its round trips few: the read-back
    // is only worth a wait when there are at least 2^22 permutation cells it could spare)
    uint64_t eligible_cells = 0;
    for (int i = 0; i < n_chips; i++)
        if (sh->log_n[i] > 10) eligible_cells += (uint64_t)air_of(sh->airs[i]).permutation_width() << sh->log_n[i];
    if (sparse_lde && eligible_cells >= min_cells && eligible_cells > 0) {
        PTRY(palloc(live_off[n_chips] * 4, &live_dev));
        PHIP(hipMemsetAsync(live_dev, 0, live_off[n_chips] * 4, ctx->stream));
    }
    PTRY(interaction_starts_batch(ctx, n_chips, sh->airs.data(), perm_alpha, beta_pows, chip_starts.data()));  // one launch, before the lanes fork
    PTRY(lane.open());
    for (int i = 0; i < n_chips; i++) {
        const lair::ChipAir& air = air_of(sh->airs[i]);
        perm_widths[i] = 4 * air.permutation_width();
        lqds[i] = air.log_quotient_degree();
        if ((int)lqds[i] > log_blowup) {
            cleanup();
            return set_error(ctx, LURKHIP_ERR_UNSUPPORTED, "chip %s needs a quotient domain larger than the LDE", air.name.c_str());
        }
    }
    // The permutation traces of one height are column ranges of one buffer with a 128-byte-aligned row pitch (plan_source_groups):
    // the first pass of their LDE then reads whole lines, like the main traces a caller lays out that way.
    std::vector<uint32_t> perm_pitch(n_chips), perm_col(n_chips);
    if (sp) {  // a buffer per chip: a cut chip's holds this rank's block of rows only
        for (int i = 0; i < n_chips; i++) {
            perm_pitch[i] = perm_widths[i];
            PTRY(palloc((((size_t)perm_pitch[i] << sh->log_n[i]) >> (cut(i) ? sp->log_g : 0)) * 4, &perm[i]));
        }
    } else {
        std::vector<int32_t> grp(n_chips);
        int32_t n_groups = 0;
        plan_source_groups(n_chips, sh->log_n.data(), perm_widths.data(), perm_pitch.data(), perm_col.data(), grp.data(), &n_groups);
        std::vector<uint32_t*> base(n_groups, nullptr);
        for (int i = 0; i < n_chips; i++) {
            if (!base[grp[i]]) PTRY(palloc(((size_t)perm_pitch[i] << sh->log_n[i]) * 4, &base[grp[i]]));
            perm[i] = base[grp[i]] + perm_col[i];
        }
    }
    for (int i = 0; i < n_chips; i++) {
        const auto on_side = lane.on_side(lane_log(i) < SIDE_LANE_MAX_LOG_N, (uint32_t)i);
        const size_t h = (size_t)1 << sh->log_n[i];
        const uint32_t* prep = sh->prep_index[i] >= 0 ? pk->traces[sh->prep_index[i]] : nullptr;
        if (cut(i)) {  // this rank's block of trace rows, its running sum from zero: the previous ranks' totals are added below
            const size_t rows = h >> sp->log_g, r0 = (size_t)sp->rank * rows;
            PTRY(permutation_trace_impl(ctx, sh->airs[i], (uint32_t)rows, sh->main[i] + (sh->main_row_blocks ? 0 : r0) * sh->main_pitch[i], prep ? prep + r0 * air_of(sh->airs[i]).prep_width : nullptr,
                                        perm_alpha, perm_beta, perm[i], nullptr, beta_pows, chip_starts[i], sh->main_pitch[i], perm_pitch[i], live_dev ? live_dev + live_off[i] : nullptr,
                                        /*starts_ready=*/true, /*defer_scan=*/true));
            PTRY(scan_ef_column(ctx, perm[i] + perm_widths[i] - 4, perm_pitch[i], rows));
            continue;
        }
        PTRY(permutation_trace_impl(ctx, sh->airs[i], (uint32_t)h, sh->main[i], prep, perm_alpha, perm_beta, perm[i], nullptr, beta_pows, chip_starts[i],
                                    sh->main_pitch[i], perm_pitch[i], live_dev ? live_dev + live_off[i] : nullptr, /*starts_ready=*/true,
                                    /*defer_scan=*/scan_is_one_chunk(h)));
    }
    PTRY(lane.close());
    {  // the short chips' running sums in one launch (each was a launch of one workgroup on its lane)
        std::vector<uint32_t*> cols;
        std::vector<uint32_t> strides, ns;
        for (int i = 0; i < n_chips; i++)
            if (!cut(i) && scan_is_one_chunk((size_t)1 << sh->log_n[i])) {
                cols.push_back(perm[i] + perm_widths[i] - 4);
                strides.push_back(perm_pitch[i]);
                ns.push_back(1u << sh->log_n[i]);
            }
        if (!cols.empty()) PTRY(scan_ef_columns_one_chunk(ctx, (int)cols.size(), cols.data(), strides.data(), ns.data()));
    }
    // cumulative sums: the last element of each trace, gathered by one launch into one buffer and copied once -- and not waited
    // for: nothing needs them before the permutation commitment's root is read back, whose wait covers this copy too
    // (22 16-byte copies and a host round trip before: 0.3 ms of the stage)
    uint32_t* cs_host = nullptr;  // page-locked (host_staging is not used again before the sums are read below)
    uint32_t* cs_dev = nullptr;
    const size_t stage_words = (size_t)n_chips * 4 + live_off[n_chips] + 12;  // sums | live flags | permutation root (8) | alpha (4)
    {
        PTRY(palloc((size_t)n_chips * 16, &cs_dev));
        PTRY(host_staging(ctx, stage_words * 4, (void**)&cs_host));
        for (int at = 0; at < n_chips; at += GATHER_EF_MAX) {
            GatherEfArgs a{};
            a.n = (uint32_t)std::min(GATHER_EF_MAX, n_chips - at);
            for (uint32_t k = 0; k < a.n; k++) {
                const int i = at + (int)k;
                a.src[k] = perm[i] + ((((size_t)1 << sh->log_n[i]) >> (cut(i) ? sp->log_g : 0)) - 1) * perm_pitch[i] + perm_widths[i] - 4;  // (a cut chip: the block's total)
            }
            hipLaunchKernelGGL(k_gather_ef, dim3(1), dim3(GATHER_EF_MAX * 4), 0, ctx->stream, a, cs_dev + (size_t)at * 4);
        }
        PHIP(hipGetLastError());
        PHIP(hipMemcpyAsync(cs_host, cs_dev, (size_t)n_chips * 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    // the live batch columns as runs of base columns (the running-sum column is always live); this is the one place the proof
    // waits for the device between the main root and the permutation root
    std::vector<ColumnRuns> live_runs;
    std::vector<uint32_t> gathered;  // (split) every rank's sums | live flags
    if (live_dev) {
        uint32_t* live_host = cs_host + (size_t)n_chips * 4;
        PHIP(hipMemcpyAsync((void*)live_host, live_dev, live_off[n_chips] * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHIP(stream_wait(ctx));
        if (sp) {
            // a cut chip's flags say what THIS rank's rows need: a column is left out of the exchanges and the LDE when no rank needs
            // it, and the quotient kernels (whose storage rows are not the trace rows the flags were made from) read the agreed flags
            const size_t per_rank = (size_t)n_chips * 4 + live_off[n_chips];
            gathered.resize(per_rank * (size_t)sp->world());
            PTRY(split_allgather_host(ctx, *sp, cs_host, gathered.data(), (uint64_t)per_rank * 4));
            for (int r = 0; r < sp->world(); r++)
                for (size_t c = 0; c < live_off[n_chips]; c++) live_host[c] |= gathered[(size_t)r * per_rank + (size_t)n_chips * 4 + c];
            PTRY(upload_words(ctx, live_dev, live_host, live_off[n_chips]));
        }
        live_runs.resize(n_chips);
        for (int i = 0; i < n_chips; i++) {
            const uint32_t pw = perm_widths[i] / 4;
            ColumnRuns& runs = live_runs[i];
            for (uint32_t c = 0; c < pw; c++) {
                if (c + 1 < pw && !live_host[live_off[i] + c]) continue;
                if (!runs.empty() && runs.back().first + runs.back().second == 4 * c) runs.back().second += 4;
                else runs.push_back({4 * c, 4u});
            }
        }
    }
    if (sp) {
        // the blocks' totals of every rank: a cut chip's running sum on this rank starts at the sum of the previous ranks' totals, its
        // cumulative sum is the sum of all of them; the other chips' sums are every rank's own (equal) values
        const size_t per_rank = gathered.empty() ? (size_t)n_chips * 4 : gathered.size() / (size_t)sp->world();
        if (gathered.empty()) {
            PHIP(stream_wait(ctx));
            gathered.resize(per_rank * (size_t)sp->world());
            PTRY(split_allgather_host(ctx, *sp, cs_host, gathered.data(), (uint64_t)per_rank * 4));
        }
        const std::vector<uint32_t>& all = gathered;
        for (int i = 0; i < n_chips; i++) {
            if (!cut(i)) {
                cumsum[i] = ef{{cs_host[4 * i], cs_host[4 * i + 1], cs_host[4 * i + 2], cs_host[4 * i + 3]}};
                continue;
            }
            ef before = bb::ef_zero(), total = bb::ef_zero();
            for (int r = 0; r < sp->world(); r++) {
                const uint32_t* t = &all[(size_t)r * per_rank + (size_t)i * 4];
                if (r == sp->rank) before = total;
                total = bb::ef_add(total, ef{{t[0], t[1], t[2], t[3]}});
            }
            cumsum[i] = total;
            if (sp->rank) PTRY(add_ef_to_column(ctx, perm[i] + perm_widths[i] - 4, perm_pitch[i], ((size_t)1 << sh->log_n[i]) >> sp->log_g, before.c));
        }
    }
    {
        uint64_t all = 0, live = 0;
        for (int i = 0; i < n_chips; i++) {
            const uint64_t rows = (uint64_t)1 << sh->log_n[i];
            all += rows * perm_widths[i];
            uint64_t w = perm_widths[i];
            if (!live_runs.empty() && (sp ? cut(i) : sh->log_n[i] > 10)) {
                w = 0;
                for (const auto& r : live_runs[i]) w += r.second;
            }
            live += rows * w;
        }
        ctx->perm_cells = all;
        ctx->perm_cells_transformed = live;
    }
    span_end(ctx, "permutation");
    // Round 5: the constraint-folding challenge on the device.  "Observe the permutation root, sample alpha" is what k_fri_challenge
    // does for a FRI layer; drawn there -- the host's transcript state uploaded as launch arguments --, alpha's powers are built
    // from device memory (ef_powers_dev) and the quotient kernels read the chips' cumulative sums where the permutation stage left
    // them: the host goes on queueing the quotient stage and its commitment while the device is still inside the permutation
    // commitment's chain of tree levels, and catches its own transcript up at the quotient root's read-back (one wait instead of
    // two; alpha is compared).  Not for the profiles that observe the sums or fold with ascending powers.
    // Measured and OFF by default (LURKHIP_DEV_ALPHA=1 turns it on; read per proof): a 2^12-row proof takes 5.48 / 5.50 ms with it and
    // 5.30 / 5.43 without, the 2^20-row step 47.7 against 47.6 -- the round trip it removes was never on the critical path: while the
    // host waits for the permutation root the device is inside that commitment's chain of tree levels, and the quotient stage's
    // kernels (0.55 ms of kernel time on four lanes) take as long to run as to queue.  What a small proof waits for is the device.
    const char* dev_alpha_env = getenv("LURKHIP_DEV_ALPHA");
    const bool dev_alpha = dev_alpha_env != nullptr && atoi(dev_alpha_env) != 0 && !prof.observe_chip_meta && !prof.constraint_alpha_ascending && !sp;
    lurkhip_commitment* perm_commit = nullptr;
    uint32_t perm_root_m[8];
    span_begin(ctx, "commit_perm");
    if (sp) {
        std::vector<SplitMat> sm(n_chips);
        for (int i = 0; i < n_chips; i++)
            sm[i] = SplitMat{perm[i], sh->log_n[i], perm_widths[i], perm_pitch[i], 0u, cut(i) ? split::K_BLOCK : split::K_FULL, 0u, 0u, 0u, 0u, live_dev && cut(i) ? &live_runs[i] : nullptr};
        PTRY(split_commit(ctx, *sp, n_chips, sm.data(), log_blowup, &perm_commit, perm_root_m));
    } else
    PTRY(commit_impl(ctx, n_chips, perm.data(), false, sh->log_n.data(), perm_widths.data(), log_blowup, LURKHIP_REPR_MONTY, 0, &perm_commit,
                     dev_alpha ? nullptr : perm_root_m, nullptr, false, /*padded_groups=*/true, perm_pitch.data(), live_dev ? &live_runs : nullptr));
    span_end(ctx, "commit_perm");
    to_free.push_back(perm_commit);
    uint32_t* alpha_dev = nullptr;
    uint32_t* const root_alpha_host = cs_host + stage_words - 12;
    if (dev_alpha) {
        DevChallenger hc{};
        memcpy(hc.state, ch.state, sizeof hc.state);
        hc.n_in = (uint32_t)ch.input.size();
        hc.n_out = (uint32_t)ch.output.size();
        hc.out_head = 0;
        hc.squeeze = (uint32_t)ch.squeeze;
        hc.pop_front = ch.pop_front ? 1u : 0u;
        for (size_t i = 0; i < ch.input.size(); i++) hc.input[i] = ch.input[i];
        for (size_t i = 0; i < ch.output.size(); i++) hc.output[i] = ch.output[i];
        DevChallenger* ch_dev = nullptr;
        uint32_t* root_copy_dev = nullptr;
        PTRY(palloc(sizeof(DevChallenger), (uint32_t**)&ch_dev));
        PTRY(palloc(48, &root_copy_dev));
        alpha_dev = root_copy_dev + 8;
        PTRY(upload_words(ctx, (uint32_t*)ch_dev, (const uint32_t*)&hc, sizeof hc / 4));
        const uint32_t* root_dev = perm_commit->digests + perm_commit->level_off[perm_commit->log_max] * 8;
        PTRY(fri_challenge(ctx, ch_dev, root_dev, alpha_dev, root_copy_dev));
        PHIP(hipMemcpyAsync(root_alpha_host, root_copy_dev, 48, hipMemcpyDeviceToHost, ctx->stream));  // read with the quotient root
    } else {
        if (!sp)
            for (int i = 0; i < n_chips; i++) cumsum[i] = ef{{cs_host[4 * i], cs_host[4 * i + 1], cs_host[4 * i + 2], cs_host[4 * i + 3]}};
        ch.observe_digest_m(perm_root_m);
        if (prof.observe_chip_meta)  // ... and the cumulative sums before the constraint-folding challenge
            for (int i = 0; i < n_chips; i++) ch.observe_ef_m(cumsum[i]);
    }

    // ---- quotient
    const ef alpha = dev_alpha ? bb::ef_zero() : ch.sample_ef_m();
    std::vector<uint32_t*> qmats;
    std::vector<uint32_t> q_logn, q_widths, q_shifts;
    std::vector<int> q_chip;  // chip of each quotient chunk
    std::vector<SplitMat> q_split;  // (sp) the chunks as split_commit takes them
    span_begin(ctx, "quotient_all");
    // tables a chip's quotient finds in the context's caches are written at first use on the stream that asks: ask on the main
    // stream, before the lanes fork (two chips of one height on two side lanes raced for the first proof of a context)
    for (int i = 0; i < n_chips; i++) (void)selector_table_of(ctx, sh->log_n[i], lqds[i]);
    // one table of alpha powers and one copy of the public values for all chips (natural order only: the reversed table of the
    // constraint_alpha_ascending profiles depends on the chip's own constraint count)
    uint32_t* alpha_pows_all = nullptr;
    uint32_t* public_m_dev = nullptr;
    if (!prof.constraint_alpha_ascending) {
        uint32_t k_max = 1;
        for (int i = 0; i < n_chips; i++) k_max = std::max(k_max, air_total_constraints(sh->airs[i]));
        PTRY(palloc((size_t)k_max * 32, &alpha_pows_all));
        if (dev_alpha) PTRY(ef_powers_dev(ctx, alpha_dev, alpha_pows_all, k_max, true));
        else PTRY(ef_powers(ctx, alpha.c, alpha_pows_all, k_max, true));
    }
    if (n_public) {
        std::vector<uint32_t> pubm(n_public);
        for (uint32_t i = 0; i < n_public; i++) pubm[i] = bb::to_monty(public_values[i] % bb::P);
        PTRY(palloc((size_t)n_public * 4, &public_m_dev));
        PTRY(upload_words(ctx, public_m_dev, pubm.data(), n_public));
    }
    PTRY(lane.open());
    for (int i = 0; i < n_chips; i++) {
        const auto on_side = lane.on_side(lane_log(i) < SIDE_LANE_MAX_LOG_N, (uint32_t)i);
        const size_t h = (size_t)1 << sh->log_n[i];
        const uint32_t qd = 1u << lqds[i];
        uint32_t* chunks = nullptr;
        const int pi = sh->prep_index[i];
        const uint32_t pitches[3] = {sh->main_commit->pitch[i], pi >= 0 ? pk->commit->pitch[pi] : 0u, perm_commit->pitch[i]};
        const uint32_t wq = two_adic_generator_monty((int)(sh->log_n[i] + lqds[i]));
        const uint32_t wq_inv = pow_host(wq, bb::P - 2);
        if (cut(i)) {
            // this rank's storage rows of the LDEs are its rows of the quotient domain (the first N << lqd storage rows of the 2N)
            const uint32_t log_rows = sh->log_n[i] + (uint32_t)log_blowup - (uint32_t)sp->log_g, rows = 1u << log_rows;
            const uint64_t s_base = (uint64_t)sp->rank << log_rows, q_rows = (uint64_t)h << lqds[i];
            const QuotientSplit qs{(uint32_t)s_base, s_base >= q_rows ? 0u : (uint32_t)std::min<uint64_t>(rows, q_rows - s_base), sh->main_commit->next_off[i], log_rows};
            PTRY(palloc((size_t)rows * 16, &chunks));
            PTRY(quotient_impl(ctx, sh->airs[i], sh->log_n[i], sh->main_commit->lde[i], pi >= 0 ? pk->commit->lde[pi] : nullptr, perm_commit->lde[i], perm_alpha, perm_beta,
                               alpha, cumsum[i], public_values, chunks, beta_pows, chip_starts[i], pitches, /*honest_running_sum=*/true, alpha_pows_all, public_m_dev, nullptr,
                               &qs, live_dev ? live_dev + live_off[i] : nullptr));
            for (uint32_t c = 0; c < qd; c++) {
                q_split.push_back(SplitMat{chunks, sh->log_n[i], 4u, 4u, bb::from_monty(pow_host(wq_inv, c)), split::K_QUOTIENT, lqds[i], c, 0u, 0u});
                q_chip.push_back(i);
            }
            continue;
        }
        PTRY(palloc(h * qd * 16, &chunks));
        // (split, a chip that is not cut: every rank has its whole LDEs)
        const uint32_t* main_lde = sp ? sh->main_commit->full_lde[i] : sh->main_commit->lde[i];
        const uint32_t* perm_lde = sp ? perm_commit->full_lde[i] : perm_commit->lde[i];
        const uint32_t* prep_lde = pi >= 0 ? (sp ? pk->commit->full_lde[pi] : pk->commit->lde[pi]) : nullptr;
        PTRY(quotient_impl(ctx, sh->airs[i], sh->log_n[i], main_lde, prep_lde, perm_lde, perm_alpha, perm_beta, alpha,
                           cumsum[i], public_values, chunks, beta_pows, chip_starts[i], pitches, /*honest_running_sum=*/true, alpha_pows_all, public_m_dev,
                           dev_alpha ? cs_dev + 4 * (size_t)i : nullptr, nullptr, live_dev ? live_dev + live_off[i] : nullptr));
        for (uint32_t c = 0; c < qd; c++) {
            qmats.push_back(chunks + (size_t)c * h * 4);
            q_logn.push_back(sh->log_n[i]);
            q_widths.push_back(4);
            q_shifts.push_back(bb::from_monty(pow_host(wq_inv, c)));  // generator / (generator * w_Q^c)
            q_chip.push_back(i);
            if (sp) q_split.push_back(SplitMat{chunks + (size_t)c * h * 4, sh->log_n[i], 4u, 4u, q_shifts.back(), split::K_FULL, 0u, 0u, 0u, 0u});
        }
    }
    PTRY(lane.close());
    span_end(ctx, "quotient_all");
    lurkhip_commitment* quot_commit = nullptr;
    uint32_t quot_root_m[8];
    span_begin(ctx, "commit_quotient");
    if (sp) PTRY(split_commit(ctx, *sp, (int)q_split.size(), q_split.data(), log_blowup, &quot_commit, quot_root_m));
    else
    PTRY(commit_impl(ctx, (int32_t)qmats.size(), qmats.data(), false, q_logn.data(), q_widths.data(), log_blowup, LURKHIP_REPR_MONTY, 0,
                     &quot_commit, quot_root_m, q_shifts.data(), false, /*padded_groups=*/true));
    span_end(ctx, "commit_quotient");
    to_free.push_back(quot_commit);
    if (dev_alpha) {  // the host's transcript catches up (the quotient root's read-back waited for everything queued before it)
        memcpy(perm_root_m, root_alpha_host, sizeof perm_root_m);
        for (int i = 0; i < n_chips; i++) cumsum[i] = ef{{cs_host[4 * i], cs_host[4 * i + 1], cs_host[4 * i + 2], cs_host[4 * i + 3]}};
        ch.observe_digest_m(perm_root_m);
        const ef alpha_host = ch.sample_ef_m();
        if (memcmp(alpha_host.c, root_alpha_host + 8, 16) != 0) {
            cleanup();
            return set_error(ctx, LURKHIP_ERR_EXEC, "internal error: the device's and the host's transcript drew different constraint-folding challenges");
        }
    }
    ch.observe_digest_m(quot_root_m);

    // ---- opening points
    const ef zeta = ch.sample_ef_m();
    // point table: per trace height, (zeta, zeta * w_N); quotient chunks only use zeta
    std::vector<ef> pts;
    std::map<uint32_t, std::pair<int, int>> pts_of_logn;
    auto points_for = [&](uint32_t log_n) {
        auto it = pts_of_logn.find(log_n);
        if (it != pts_of_logn.end()) return it->second;
        // the zeta entry is shared by every height: index 0
        if (pts.empty()) pts.push_back(zeta);
        pts.push_back(bb::ef_scale(zeta, two_adic_generator_monty((int)log_n)));
        auto pr = std::make_pair(0, (int)pts.size() - 1);
        pts_of_logn.emplace(log_n, pr);
        return pr;
    };
    std::vector<Round> rounds;
    if (pk->commit) {
        Round r{pk->commit, {}};
        for (size_t m = 0; m < pk->traces.size(); m++) {
            auto pr = points_for(pk->log_heights[m]);
            r.points.push_back({pr.first, pr.second});
        }
        rounds.push_back(r);
    }
    for (lurkhip_commitment* c : {sh->main_commit, perm_commit}) {
        Round r{c, {}};
        for (int i = 0; i < n_chips; i++) {
            auto pr = points_for(sh->log_n[i]);
            r.points.push_back({pr.first, pr.second});
        }
        rounds.push_back(r);
    }
    {
        if (pts.empty()) pts.push_back(zeta);
        Round r{quot_commit, {}};
        for (size_t m = 0; m < q_chip.size(); m++) r.points.push_back({0});
        rounds.push_back(r);
    }

    // ---- p3 TwoAdicFriPcs::open (values at the points, reduced openings, FRI)
    OpenOut oo;
    PTRY(pcs_open_impl(ctx, prof, rounds, pts, log_blowup, ch, num_queries, pow_bits, to_free, pooled, oo, sp));
    const auto& opened = oo.opened;
    const auto& layer_roots_m = oo.layer_roots_m;
    const ef final_poly = oo.final_poly;
    const uint32_t pow_witness = oo.pow_witness;
    const auto& indices = oo.indices;
    const auto& round_record_words = oo.round_record_words;
    const auto& layer_record_words = oo.layer_record_words;
    const auto& round_off = oo.round_off;
    const auto& layer_off = oo.layer_off;
    const uint32_t* rec_host = oo.rec_host;
    const size_t rec_words = oo.rec_words;
    const int log_max = oo.log_max;
    const size_t n_fri_layers = oo.n_layers;

    // ---- serialise (canonical values); layout documented in lurk_amd/prover.py
    auto* proof = new lurkhip_proof();
    std::vector<uint32_t>& o = proof->words;
    o.insert(o.end(), {PROOF_MAGIC, (uint32_t)n_chips, (uint32_t)log_blowup, num_queries, pow_bits, n_public, (uint32_t)n_fri_layers,
                       (uint32_t)log_max, (uint32_t)(pk->commit ? pk->traces.size() : 0), (uint32_t)q_chip.size()});
    for (int i = 0; i < n_chips; i++) {
        const lair::ChipAir& air = air_of(sh->airs[i]);
        o.insert(o.end(), {(uint32_t)sh->machine_index[i], sh->log_n[i], air.width, air.prep_width, perm_widths[i], 1u << lqds[i],
                           (uint32_t)(sh->prep_index[i] + 1)});
        push_ef(o, cumsum[i]);
    }
    for (uint32_t i = 0; i < n_public; i++) o.push_back(public_values[i] % bb::P);
    for (int i = 0; i < 8; i++) o.push_back(bb::from_monty(sh->root_m[i]));
    for (int i = 0; i < 8; i++) o.push_back(bb::from_monty(perm_root_m[i]));
    for (int i = 0; i < 8; i++) o.push_back(bb::from_monty(quot_root_m[i]));
    // opened values: round by round, matrix by matrix, point by point, column by column
    for (size_t ri = 0; ri < rounds.size(); ri++)
        for (auto& mat : opened[ri])
            for (auto& ys : mat)
                for (const ef& y : ys) push_ef(o, y);
    for (uint32_t v : layer_roots_m) o.push_back(bb::from_monty(v));
    push_ef(o, final_poly);
    o.push_back(pow_witness);
    for (uint32_t ix : indices) o.push_back(ix);
    o.reserve(o.size() + rec_words + rounds.size() + n_fri_layers);
    for (size_t ri = 0; ri < rounds.size(); ri++) {
        o.push_back(round_record_words[ri]);
        const size_t n = (size_t)num_queries * round_record_words[ri];
        if (!oo.round_records[ri].empty()) o.insert(o.end(), oo.round_records[ri].begin(), oo.round_records[ri].end());  // (a split commitment's, assembled on the host)
        else o.insert(o.end(), rec_host + round_off[ri], rec_host + round_off[ri] + n);  // (canonical already: k_records_canonical)
    }
    for (size_t li = 0; li < n_fri_layers; li++) {
        o.push_back(layer_record_words[li]);
        const size_t n = (size_t)num_queries * layer_record_words[li];
        if (!oo.layer_records[li].empty()) o.insert(o.end(), oo.layer_records[li].begin(), oo.layer_records[li].end());
        else o.insert(o.end(), rec_host + layer_off[li], rec_host + layer_off[li] + n);
    }
    cleanup();
#undef PTRY
#undef PHIP
    *out = proof;
    return LURKHIP_OK;
}

// ------------------------------------------------------------------ standalone Pcs::open
int32_t lurkhip_open(lurkhip_ctx* ctx, int32_t n_rounds, lurkhip_commitment* const* commitments, const uint32_t* n_points,
                     const uint32_t* points, lurkhip_challenger* chal, uint32_t num_queries, uint32_t pow_bits, lurkhip_proof** out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, n_rounds >= 1 && n_rounds <= 64 && commitments && n_points && points && chal && out, "bad open arguments");
    LH_ARG(ctx, num_queries >= 1 && num_queries <= 1024 && pow_bits <= 30, "bad FRI parameters");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<Round> rounds;
    std::vector<ef> pts;
    std::map<std::array<uint32_t, 4>, int> pt_index;
    const int log_blowup = commitments[0] ? commitments[0]->log_blowup : 0;
    size_t k = 0, pk_ = 0;
    for (int r = 0; r < n_rounds; r++) {
        lurkhip_commitment* c = commitments[r];
        LH_ARG(ctx, c && c->n_mats > 0, "round %d: null or empty commitment", r);
        LH_ARG(ctx, c->log_blowup == log_blowup && log_blowup >= 1, "round %d: every commitment must use the same blow-up (>= 2)", r);
        Round rd{c, {}};
        for (int m = 0; m < c->n_mats; m++, k++) {
            LH_ARG(ctx, n_points[k] == 1 || n_points[k] == 2, "round %d matrix %d: one or two opening points per matrix", r, m);
            std::vector<int> mp;
            for (uint32_t q = 0; q < n_points[k]; q++, pk_++) {
                std::array<uint32_t, 4> key;
                for (int i = 0; i < 4; i++) {
                    LH_ARG(ctx, points[4 * pk_ + i] < bb::P, "opening point is not canonical");
                    key[i] = points[4 * pk_ + i];
                }
                auto it = pt_index.find(key);
                if (it == pt_index.end()) {
                    it = pt_index.emplace(key, (int)pts.size()).first;
                    pts.push_back(ef{{bb::to_monty(key[0]), bb::to_monty(key[1]), bb::to_monty(key[2]), bb::to_monty(key[3])}});
                }
                mp.push_back(it->second);
            }
            LH_ARG(ctx, mp.size() == 1 || mp[0] != mp[1], "round %d matrix %d: the two points coincide", r, m);
            rd.points.push_back(mp);
        }
        rounds.push_back(rd);
    }
    const lurkhip_protocol_profile prof = profile_of(ctx);
    std::vector<lurkhip_commitment*> to_free;
    std::vector<void*> pooled;
    OpenOut oo;
    int32_t st;
    try {  // nothing unwinds across the C boundary
        st = pcs_open_impl(ctx, prof, rounds, pts, log_blowup, chal->ch, num_queries, pow_bits, to_free, pooled, oo);
    } catch (const std::bad_alloc&) {
        st = set_error(ctx, LURKHIP_ERR_OOM, "host allocation failed while opening");
    } catch (const std::exception& e) {
        st = set_error(ctx, LURKHIP_ERR_EXEC, "internal error while opening: %s", e.what());
    }
    lurkhip_proof* proof = nullptr;
    if (st == LURKHIP_OK) {
        // layout documented in lurk_amd/commit.py (parse_opening)
        proof = new lurkhip_proof();
        std::vector<uint32_t>& o = proof->words;
        o.insert(o.end(), {OPENING_MAGIC, (uint32_t)n_rounds, (uint32_t)log_blowup, num_queries, pow_bits, (uint32_t)oo.n_layers, (uint32_t)oo.log_max});
        for (const Round& r : rounds) {
            o.push_back((uint32_t)r.c->n_mats);
            for (int m = 0; m < r.c->n_mats; m++)
                o.insert(o.end(), {(uint32_t)(r.c->log_h[m] - log_blowup), r.c->width[m], (uint32_t)r.points[m].size()});
        }
        for (size_t ri = 0; ri < rounds.size(); ri++)
            for (auto& mat : oo.opened[ri])
                for (auto& ys : mat)
                    for (const ef& y : ys) push_ef(o, y);
        for (uint32_t v : oo.layer_roots_m) o.push_back(bb::from_monty(v));
        push_ef(o, oo.final_poly);
        o.push_back(oo.pow_witness);
        for (uint32_t ix : oo.indices) o.push_back(ix);
        for (size_t ri = 0; ri < rounds.size(); ri++) {
            o.push_back(oo.round_record_words[ri]);
            const size_t n = (size_t)num_queries * oo.round_record_words[ri];
            o.insert(o.end(), oo.rec_host + oo.round_off[ri], oo.rec_host + oo.round_off[ri] + n);  // (canonical already: k_records_canonical)
        }
        for (size_t li = 0; li < oo.n_layers; li++) {
            o.push_back(oo.layer_record_words[li]);
            const size_t n = (size_t)num_queries * oo.layer_record_words[li];
            o.insert(o.end(), oo.rec_host + oo.layer_off[li], oo.rec_host + oo.layer_off[li] + n);
        }
    }
    (void)stream_wait(ctx);
    for (auto* c : to_free) free_commitment(ctx, c);
    for (void* p : pooled) pool_release(ctx, p);
    if (st != LURKHIP_OK) return st;
    *out = proof;
    return LURKHIP_OK;
}

int64_t lurkhip_proof_words(const lurkhip_proof* p) { return p ? (int64_t)p->words.size() : -1; }
int32_t lurkhip_proof_read(const lurkhip_proof* p, uint32_t* out, uint64_t capacity_words) {
    if (!p || !out || capacity_words < p->words.size()) return LURKHIP_ERR_INVALID_ARG;
    memcpy(out, p->words.data(), p->words.size() * 4);
    return LURKHIP_OK;
}
int32_t lurkhip_proof_free(lurkhip_proof* p) {
    delete p;
    return LURKHIP_OK;
}

}  // extern "C"
