// C-ABI of the commit stage: coset LDE of every trace matrix + one mixed-height Merkle tree.
//
// Replaces (S1 in SURVEY.md 8a; third-party): p3 TwoAdicFriPcs::commit(Vec<(domain, RowMajorMatrix)>)
// as called by sphinx's prover for the main, permutation and quotient traces [UPSTREAM-RECALL].
#include <stdlib.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <numeric>

#include "babybear.h"
#include "commit.h"
#include "lde.h"

namespace lurkhip {

int32_t get_ntt_plan(lurkhip_ctx* ctx, int log_n, const NttPlan** out) {
    LH_ARG(ctx, log_n >= 0 && log_n <= bb::TWO_ADICITY, "log_n %d outside [0,27]", log_n);
    if (!ctx->ntt_plans[log_n]) {
        NttPlan* p = new NttPlan();
        int32_t s = p->init(ctx, log_n);
        if (s != LURKHIP_OK) {
            delete p;
            return s;
        }
        ctx->ntt_plans[log_n] = p;
        ctx->cleanups.push_back([p]() {
            p->destroy();
            delete p;
        });
    }
    *out = (const NttPlan*)ctx->ntt_plans[log_n];
    return LURKHIP_OK;
}

namespace {

uint32_t hpow(uint32_t a_m, uint64_t e) {
    uint32_t r = bb::R1;
    while (e) {
        if (e & 1) r = bb::mul(r, a_m);
        a_m = bb::mul(a_m, a_m);
        e >>= 1;
    }
    return r;
}

uint32_t brev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

// evaluations (natural order over H) -> coefficients (natural order), Montgomery, unscaled by 1/N.
// `scratch` and `coef` are N x w device buffers.
int32_t interpolate(lurkhip_ctx* ctx, int log_n, int w, const uint32_t* evals, bool canonical, uint32_t* scratch,
                    uint32_t* coef) {
    const NttPlan* plan = nullptr;
    LH_TRY(get_ntt_plan(ctx, log_n, &plan));
    return ntt_dif(ctx, *plan, /*inverse=*/true, evals, coef, scratch, w, nullptr, canonical, false, /*bitrev_store=*/true);
}

// s^i / N, i < N, for the coset shift s: immutable, so cached per (log_n, s) while the cache stays under 1 GiB (nullptr
// beyond that: the caller builds the table in its own scratch)
int32_t cached_scale_table(lurkhip_ctx* ctx, int log_n, uint32_t s_m, const uint32_t** out) {
    const size_t n = (size_t)1 << log_n;
    *out = nullptr;
    auto key = std::make_pair(log_n, s_m);
    auto it = ctx->lde_scale_tables.find(key);
    if (it != ctx->lde_scale_tables.end()) {
        *out = it->second;
        return LURKHIP_OK;
    }
    if (ctx->lde_scale_bytes + n * 4 > ((size_t)1 << 30)) return LURKHIP_OK;
    const uint32_t n_inv = hpow(bb::to_monty((uint32_t)(n % bb::P)), bb::P - 2);
    uint32_t* tbl = nullptr;
    LH_HIP(ctx, hipMalloc(&tbl, n * 4));
    LH_TRY(fill_powers(ctx, tbl, s_m, n_inv, n));
    ctx->lde_scale_tables[key] = tbl;
    ctx->lde_scale_bytes += n * 4;
    *out = tbl;
    return LURKHIP_OK;
}

// coefficients -> LDE on the coset g * <w_{N << b}>, rows in bit-reversed order, Montgomery
int32_t extend(lurkhip_ctx* ctx, int log_n, int w, int log_blowup, const uint32_t* coef, uint32_t* lde,
               uint32_t* row_scale /* N words scratch */, bool out_canonical, uint32_t shift_m) {
    const NttPlan* plan = nullptr;
    LH_TRY(get_ntt_plan(ctx, log_n, &plan));
    const size_t n = (size_t)1 << log_n;
    const uint32_t n_inv = hpow(bb::to_monty((uint32_t)(n % bb::P)), bb::P - 2);
    const uint32_t w_big = two_adic_generator_monty(log_n + log_blowup);
    for (uint32_t q = 0; q < (1u << log_blowup); q++) {
        const uint32_t s_q = bb::mul(shift_m, hpow(w_big, brev(q, log_blowup)));
        const uint32_t* scale = nullptr;
        LH_TRY(cached_scale_table(ctx, log_n, s_q, &scale));
        if (!scale) {
            LH_TRY(fill_powers(ctx, row_scale, s_q, n_inv, n));
            scale = row_scale;
        }
        LH_TRY(ntt_dif(ctx, *plan, /*inverse=*/false, coef, lde + q * n * w, nullptr, w, scale, false, out_canonical,
                       /*bitrev_store=*/false));
    }
    return LURKHIP_OK;
}

// interpolate + extend for up to NTT_MAX_BATCH device-resident matrices of one shape (blow-up >= 1: the LDE buffer is the
// interpolation scratch): one launch per pass for all of them.  *done = false (nothing launched for the extension) when a
// coset's scale table is not cacheable; the caller then extends the matrices one by one.
// `coef_tiled_words`: capacity of every coefs[m] in words when it is at least ntt_tiled_words(log_n, w) (else 0): the
// coefficients then live chunk-tiled between the inverse and the forward transforms, the inverse's intermediate goes tiled into
// the LDE buffer (2 N w words: always large enough) and the forward transforms' into a pooled temporary -- of the twelve matrix
// transfers of an LDE only the first read (the caller's row-major trace) and the last two writes (the committed row-major
// LDE) keep unaligned row segments.
int32_t lde_batch(lurkhip_ctx* ctx, int log_n, int w, int log_blowup, int n_batch, const uint32_t* const* evals, bool canonical,
                  uint32_t* const* coefs, uint32_t* const* ldes, const uint32_t* shifts_m, bool* done, size_t coef_tiled_words,
                  bool coefs_wanted) {
    const NttPlan* plan = nullptr;
    LH_TRY(get_ntt_plan(ctx, log_n, &plan));
    const size_t n = (size_t)1 << log_n;
    *done = false;
    const size_t tw = ntt_tiled_words(log_n, w);
    const bool tiled = tw != 0 && coef_tiled_words >= tw && tw <= ((size_t)w << (log_n + log_blowup));
    const uint32_t w_big = two_adic_generator_monty(log_n + log_blowup);
    std::vector<NttBatch> cosets((size_t)1 << log_blowup);
    for (uint32_t q = 0; q < (1u << log_blowup); q++) {
        NttBatch& e = cosets[q];
        e = NttBatch{};
        e.n = n_batch;
        for (int m = 0; m < n_batch; m++) {
            const uint32_t s_q = bb::mul(shifts_m[m], hpow(w_big, brev(q, log_blowup)));
            LH_TRY(cached_scale_table(ctx, log_n, s_q, &e.row_scale[m]));
            if (!e.row_scale[m]) return LURKHIP_OK;
            e.src[m] = coefs[m];
            e.dst[m] = ldes[m] + q * n * w;
        }
    }
    NttBatch b{};
    b.n = n_batch;
    for (int m = 0; m < n_batch; m++) {
        b.src[m] = evals[m];
        b.dst[m] = coefs[m];
        b.scratch[m] = ldes[m];
    }
    // Fused route (round 3, ntt.hip: k_ntt_fused): the inverse's last pass, the coset scalings and both forward first passes in
    // one launch; coefs[m] is only the inverse's intermediate buffer then (never holds coefficients).
    if (!tiled && !coefs_wanted && log_blowup == 1 && ntt_lde_fused_eligible(log_n, w)) {
        NttBatch inv = b;
        inv.reversed_schedule = true;
        inv.skip_last_pass = true;
        for (int m = 0; m < n_batch; m++) inv.scratch[m] = nullptr;
        LH_TRY(ntt_dif_batch(ctx, *plan, /*inverse=*/true, inv, w, canonical, false, /*bitrev_store=*/false));
        uint32_t* outs[NTT_MAX_BATCH][2];
        const uint32_t* scales[NTT_MAX_BATCH][2];
        for (int m = 0; m < n_batch; m++)
            for (int q = 0; q < 2; q++) {
                outs[m][q] = cosets[q].dst[m];
                scales[m][q] = cosets[q].row_scale[m];
            }
        bool fused = false;
        LH_TRY(ntt_lde_fused(ctx, *plan, n_batch, coefs, outs, scales, w, &fused));
        if (!fused) return set_error(ctx, LURKHIP_ERR_HIP, "fused LDE pass refused a shape it declared eligible");
        for (NttBatch e : cosets) {
            for (int m = 0; m < n_batch; m++) {
                e.src[m] = e.dst[m];
                e.row_scale[m] = nullptr;
            }
            e.skip_first_pass = true;
            LH_TRY(ntt_dif_batch(ctx, *plan, /*inverse=*/false, e, w, false, false, /*bitrev_store=*/false));
        }
        *done = true;
        return LURKHIP_OK;
    }
    b.dst_tiled = b.scratch_tiled = tiled;
    std::vector<void*> temps;
    if (tiled)
        for (int m = 0; m < n_batch; m++) {
            void* t = nullptr;
            const int32_t st = pool_alloc(ctx, tw * 4, &t);
            if (st != LURKHIP_OK) {
                for (void* p : temps) pool_release(ctx, p);
                return st;
            }
            temps.push_back(t);
            for (NttBatch& e : cosets) e.scratch[m] = (uint32_t*)t;  // the cosets run one after the other on the stream
        }
    for (NttBatch& e : cosets) e.src_tiled = e.scratch_tiled = tiled;
    int32_t st = ntt_dif_batch(ctx, *plan, /*inverse=*/true, b, w, canonical, false, /*bitrev_store=*/true);
    for (const NttBatch& e : cosets)
        if (st == LURKHIP_OK) st = ntt_dif_batch(ctx, *plan, /*inverse=*/false, e, w, false, false, /*bitrev_store=*/false);
    for (void* p : temps) pool_release(ctx, p);  // stream-ordered (deferred while a side lane is open)
    LH_TRY(st);
    *done = true;
    return LURKHIP_OK;
}

}  // namespace

void free_commitment(lurkhip_ctx* ctx, lurkhip_commitment* c) {
    if (!c) return;
    if (c->aux) free_commitment(ctx, c->aux);
    c->aux = nullptr;
    for (void* p : c->early_scratch) pool_release(ctx, p);
    c->early_scratch.clear();
    if (c->early_level > 0) pool_release(ctx, c->early_digests);  // (an early sponge whose tree was never built)
    if (c->owns_lde)
        for (size_t i = 0; i < c->lde.size(); i++)
            if (i >= c->lde_is_view.size() || !c->lde_is_view[i]) pool_release(ctx, c->lde[i]);
    for (auto p : c->coeffs) pool_release(ctx, p);
    for (auto p : c->owned) pool_release(ctx, p);
    pool_release(ctx, c->digests);
    delete c;
}

namespace {

// early_sponge: a group of at most this many LDE rows (the row sponge kernel hashes such rows sixteen lanes to the row) and at least
// this many columns (128 permutations one after the other: 0.4 ms)
constexpr size_t EARLY_SPONGE_MAX_ROWS = 4096;
constexpr uint32_t EARLY_SPONGE_MIN_WIDTH = 1024;

// The uniform column table of the matrices in `idx` (one LeafCol per column of the concatenated row), written on the device by
// a kernel that takes the matrices' (base, width) pairs as launch arguments: no host staging, no copy, nothing to keep alive.
// The table is scratch of the tree being built (pooled, released with the tree's other scratch).  Round 1 cached the tables per
// context keyed by their matrix pointers and uploaded a miss from host memory; the pool hands a proof's matrices back in a
// different permutation every proof, so in a proving loop every lookup missed -- a hipMalloc and a pageable copy per group --
// and the cache's bound (1024 tables, every 17th fib-mix proof) cost 1024 hipFree behind a stream wait: one 80-125 ms step.
constexpr int COLFILL_MATS = 24;
struct ColFill {
    const uint32_t* base[COLFILL_MATS];
    uint32_t width[COLFILL_MATS];      // row pitch in words (LeafCol::width is what the hashing kernels multiply a row index by)
    uint32_t start[COLFILL_MATS + 1];  // first column of matrix i in this launch's part of the table
    uint32_t n;
};
__global__ __launch_bounds__(256) void k_fill_cols(LeafCol* __restrict__ out, ColFill a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.start[a.n]) return;
    // (selects over compile-time indices: a run-time index would move the argument block to scratch)
    const uint32_t* base = a.base[0];
    uint32_t width = a.width[0], first = 0;
#pragma unroll
    for (int k = 1; k < COLFILL_MATS; k++)
        if ((uint32_t)k < a.n && i >= a.start[k]) {
            base = a.base[k];
            width = a.width[k];
            first = a.start[k];
        }
    out[i] = LeafCol{base, width, i - first};
}

int32_t make_cols(lurkhip_ctx* ctx, lurkhip_commitment* c, const std::vector<int>& idx, LeafCol** out_dev,
                  uint32_t* total_w, std::vector<void*>& scratch) {
    uint32_t n_cols = 0;
    for (int m : idx) n_cols += c->width[m];
    *total_w = n_cols;
    void* dev = nullptr;
    LH_TRY(pool_alloc(ctx, std::max<size_t>(n_cols, 1) * sizeof(LeafCol), &dev));
    scratch.push_back(dev);
    uint32_t done = 0;
    for (size_t i0 = 0; i0 < idx.size(); i0 += COLFILL_MATS) {
        ColFill a{};
        a.n = (uint32_t)std::min<size_t>(COLFILL_MATS, idx.size() - i0);
        for (uint32_t k = 0; k < a.n; k++) {
            const int m = idx[i0 + k];
            a.base[k] = c->lde[m];
            a.width[k] = c->pitch[m];
            a.start[k + 1] = a.start[k] + c->width[m];
        }
        const uint32_t part = a.start[a.n];
        if (part) hipLaunchKernelGGL(k_fill_cols, dim3((part + 255) / 256), dim3(256), 0, ctx->stream, (LeafCol*)dev + done, a);
        done += part;
    }
    LH_HIP(ctx, hipGetLastError());
    *out_dev = (LeafCol*)dev;
    return LURKHIP_OK;
}

// The last TOP_NODES nodes of a tree collapse inside one workgroup (merkle_top).  A level costs about one cooperative
// permutation (~6 us) either way; measured on the bench shard 64 is 0.2 ms per proof better than 512 (the wider levels gain
// from being spread over the CUs) and no different from 16 or 2.
constexpr size_t TOP_NODES = 64;

// whether build_tree hashes every injected group's rows ahead of the levels (one launch of row sponges, the levels only compress)
bool tree_prehashes_rows(size_t n_leaves) {
    const char* fused_env = getenv("LURKHIP_MERKLE_FUSED");
    const char* group_env = getenv("LURKHIP_MERKLE_COOP_GROUP");
    return (fused_env == nullptr || atoi(fused_env) != 0) && n_leaves > MERKLE_COOP_MAX_PARENTS && (group_env == nullptr || atoi(group_env) > 1);
}

// The row sponge of the height group of LDE height 2^lde_log_h, launched on the context's hash stream behind `ctx->hash_ready`
// (recorded by the caller right after that group's LDE passes, on the stream they were queued on): it runs under the LDE passes
// of the other groups.  Meant for the group whose rows are the longest CHAIN -- a fib machine's hash chips, 2125 columns at 2^9
// rows: 266 sixteen-lane permutations one after the other, 0.83 ms whatever else the device does, which used to start only when
// every LDE of the commitment was done (0.6 ms later in a 2^12-row proof).  build_tree then leaves that group out of its sponge
// launch and waits for `hash_done` before the first level.  Must be called while the caller's side lane is open (pool releases
// deferred: nothing this hands to the hash stream is a block that work queued behind the event has just released).
int32_t early_sponge(lurkhip_ctx* ctx, lurkhip_commitment* c, int lde_log_h) {
    const P16Params* params = nullptr;
    LH_TRY(get_merkle_params(ctx, &params));
    c->log_max = *std::max_element(c->log_h.begin(), c->log_h.end());
    const size_t n_leaves = (size_t)1 << c->log_max;
    if (!tree_prehashes_rows(n_leaves)) return LURKHIP_OK;
    const int level = c->log_max - lde_log_h;
    std::vector<int> group;
    for (int m = 0; m < c->n_mats; m++)
        if (c->log_h[m] == lde_log_h) group.push_back(m);  // (ascending index = the stable height order of build_tree)
    if (group.empty()) return LURKHIP_OK;
    const size_t n_rows = (size_t)1 << lde_log_h;
    uint32_t* out = nullptr;
    if (level == 0) {
        c->level_off.assign(c->log_max + 1, 0);
        size_t total = 0;
        for (int l = 0; l <= c->log_max; l++) {
            c->level_off[l] = total;
            total += n_leaves >> l;
        }
        LH_TRY(pool_alloc(ctx, total * 8 * sizeof(uint32_t), (void**)&c->digests));
        out = c->digests;
    } else {
        LH_TRY(pool_alloc(ctx, n_rows * 8 * sizeof(uint32_t), (void**)&c->early_digests));
        out = c->early_digests;
    }
    c->early_level = level;
    hipStream_t main_stream = ctx->stream;
    LH_HIP(ctx, hipStreamWaitEvent(ctx->hash_stream, ctx->hash_ready, 0));
    ctx->stream = ctx->hash_stream;
    LeafCol* cols = nullptr;
    uint32_t tw = 0;
    int32_t st = make_cols(ctx, c, group, &cols, &tw, c->early_scratch);
    if (st == LURKHIP_OK) {
        SpongeGroups g{};
        g.cols[0] = cols, g.total_w[0] = tw, g.n_rows[0] = n_rows, g.out[0] = out, g.n = 1;
        st = merkle_row_sponges(ctx, params, g);
    }
    hipError_t e = hipEventRecord(ctx->hash_done, ctx->hash_stream);
    ctx->stream = main_stream;
    if (st == LURKHIP_OK && e != hipSuccess) st = set_error(ctx, LURKHIP_ERR_HIP, "hash stream: %s", hipGetErrorString(e));
    return st;
}

int32_t build_tree(lurkhip_ctx* ctx, lurkhip_commitment* c) {
    const P16Params* params = nullptr;
    LH_TRY(get_merkle_params(ctx, &params));
    c->log_max = *std::max_element(c->log_h.begin(), c->log_h.end());
    // stable order by height, tallest first (p3 sorts matrices by height descending)
    std::vector<int> order(c->n_mats);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return c->log_h[a] > c->log_h[b]; });
    const size_t n_leaves = (size_t)1 << c->log_max;
    c->level_off.assign(c->log_max + 1, 0);
    size_t total = 0;
    for (int l = 0; l <= c->log_max; l++) {
        c->level_off[l] = total;
        total += n_leaves >> l;
    }
    const int early_level = c->early_level;  // (-1: no group was hashed ahead)
    c->early_level = -1;
    if (early_level != 0) LH_TRY(pool_alloc(ctx, total * 8 * sizeof(uint32_t), (void**)&c->digests));
    // leaves
    std::vector<int> tallest;
    for (int m : order)
        if (c->log_h[m] == c->log_max) tallest.push_back(m);
    // the hashing spans of a small tree (the FRI layers below 2^16 leaves: a latency chain of a few launches each) are detail
    const int span_level = c->log_max >= 16 ? 1 : 2;
    // Row sponges ahead of the levels: the leaves and every group of rows injected at a level of more than COOP_MAX_PARENTS
    // parents (the levels below that hash their few rows lane-cooperatively, merkle.hip) share ONE launch, longest rows first
    // (merkle_row_sponges); the levels then only compress.  LURKHIP_MERKLE_FUSED=0: the round-1 schedule, every level hashing its own rows.
    const char* fused_env = getenv("LURKHIP_MERKLE_FUSED");
    const bool fused = (fused_env == nullptr || atoi(fused_env) != 0) && n_leaves > MERKLE_COOP_MAX_PARENTS;
    std::vector<uint32_t*> inj_digests(c->log_max + 1, nullptr);  // by level
    std::vector<void*> scratch;
    scratch.swap(c->early_scratch);  // (the early sponge's column table and digests: released with the rest, behind the join below)
    if (early_level > 0) scratch.push_back(c->early_digests);
    auto drop_scratch = [&]() {
        for (void* p : scratch) pool_release(ctx, p);  // stream-ordered: reusable by later work only
    };
    const char* group_env = getenv("LURKHIP_MERKLE_COOP_GROUP");  // cooperative levels per launch (1 = one launch per level, round 1)
    const int coop_group = group_env ? std::max(1, std::min(5, atoi(group_env))) : 5;
    // With grouped cooperative levels the rows injected at the cooperative levels and in the one-workgroup top are hashed ahead
    // too: hashed where they are injected, the 1963 columns of a fib machine's hash chips (512 LDE rows) were 246 cooperative
    // permutations in a row -- one 0.9 ms launch of 512 workgroups -- and in the sponge launch they are 8 waves among thousands.
    const bool prehash_all = fused && coop_group > 1;
    if (fused)
        for (int l = 1, groups = 1; l <= c->log_max && groups < SPONGE_MAX_GROUPS; l++) {
            const size_t n_parents = n_leaves >> l;
            if (n_parents <= MERKLE_COOP_MAX_PARENTS && !prehash_all) break;
            bool any = false;
            for (int m : order) any = any || c->log_h[m] == c->log_max - l;
            if (!any) continue;
            if (l == early_level) {  // hashed ahead on the hash stream (early_sponge)
                inj_digests[l] = c->early_digests;
                continue;
            }
            void* d = nullptr;
            const int32_t st = pool_alloc(ctx, n_parents * 8 * sizeof(uint32_t), &d);
            if (st != LURKHIP_OK) {
                drop_scratch();
                return st;
            }
            scratch.push_back(d);
            inj_digests[l] = (uint32_t*)d;
            groups++;
        }
    LeafCol* cols = nullptr;
    uint32_t tw = 0;
    if (early_level != 0 && !fused) {
        const int32_t st = make_cols(ctx, c, tallest, &cols, &tw, scratch);
        if (st != LURKHIP_OK) {
            drop_scratch();
            return st;
        }
    }
    span_begin(ctx, "merkle_leaves", span_level);
    if (fused) {
        // ONE column table for every group of the sponge launch (round 4: one k_fill_cols launch per tree instead of one per height
        // group -- a dozen 4 us kernels in a row on the stream of every commitment): the groups' matrices in group order, a group's
        // table is a run of it
        SpongeGroups g{};
        std::vector<int> flat;
        std::vector<uint32_t> group_w;
        auto add_group = [&](const std::vector<int>& mats, uint64_t n_rows, uint32_t* out) {
            uint32_t w = 0;
            for (int m : mats) {
                flat.push_back(m);
                w += c->width[m];
            }
            group_w.push_back(w);
            g.total_w[g.n] = w;
            g.n_rows[g.n] = n_rows;
            g.out[g.n] = out;
            g.n++;
        };
        if (early_level != 0) add_group(tallest, n_leaves, c->digests);
        for (int l = 1; l <= c->log_max; l++) {
            if (!inj_digests[l] || l == early_level) continue;
            std::vector<int> inject;
            for (int m : order)
                if (c->log_h[m] == c->log_max - l) inject.push_back(m);
            add_group(inject, n_leaves >> l, inj_digests[l]);
        }
        int32_t st = LURKHIP_OK;
        if (g.n) {
            LeafCol* all_cols = nullptr;
            uint32_t all_w = 0;
            st = make_cols(ctx, c, flat, &all_cols, &all_w, scratch);
            uint32_t at = 0;
            for (int k = 0; k < g.n; k++) {
                g.cols[k] = all_cols + at;
                at += group_w[k];
            }
        }
        if (st == LURKHIP_OK && g.n) st = merkle_row_sponges(ctx, params, g);
        if (st == LURKHIP_OK && early_level >= 0 && hipStreamWaitEvent(ctx->stream, ctx->hash_done, 0) != hipSuccess)
            st = set_error(ctx, LURKHIP_ERR_HIP, "joining the hash stream failed");
        if (st != LURKHIP_OK) {
            drop_scratch();
            return st;
        }
    } else {
        const int32_t st = merkle_leaves(ctx, params, cols, tw, n_leaves, c->digests);
        if (st != LURKHIP_OK) {
            drop_scratch();
            return st;
        }
    }
    const char* stage = "merkle_leaves";  // the span that is open
    // inner levels
    int32_t status = LURKHIP_OK;
    for (int l = 1; l <= c->log_max && status == LURKHIP_OK; l++) {
        const size_t n_parents = n_leaves >> l;
        const int lh = c->log_max - l;
        std::vector<int> inject;
        for (int m : order)
            if (c->log_h[m] == lh) inject.push_back(m);
        uint32_t* children = c->digests + c->level_off[l - 1] * 8;
        uint32_t* parents = c->digests + c->level_off[l] * 8;
        if ((n_parents << 1) <= TOP_NODES) {
            // finish the tree in one workgroup (wider levels are faster spread over the CUs)
            TopInject ti{};
            for (int t = 0; l + t <= c->log_max && status == LURKHIP_OK; t++) {
                if (inj_digests[l + t]) {  // hashed ahead
                    ti.dig[t] = inj_digests[l + t];
                    continue;
                }
                std::vector<int> inj;
                for (int m : order)
                    if (c->log_h[m] == lh - t) inj.push_back(m);
                if (inj.empty()) continue;
                LeafCol* tc = nullptr;
                status = make_cols(ctx, c, inj, &tc, &ti.w[t], scratch);
                ti.cols[t] = tc;
            }
            if (status != LURKHIP_OK) break;
            span_switch(ctx, stage, "merkle_top", span_level);
            stage = "merkle_top";
            status = merkle_top(ctx, params, children, n_parents << 1, ti);
            break;
        }
        if (l == 1) {
            span_switch(ctx, stage, "merkle_levels", span_level);
            stage = "merkle_levels";
        }
        if (fused && n_parents > MERKLE_COOP_MAX_PARENTS && (inject.empty() || inj_digests[l])) {
            status = merkle_level_digests(ctx, params, children, n_parents, inj_digests[l], parents);
            continue;
        }
        if (n_parents <= MERKLE_COOP_MAX_PARENTS && coop_group > 1) {
            // the cooperative levels from here to the one-workgroup top, up to five per launch (merkle.hip: k_levels_coop)
            int group = 0;
            while (group < coop_group && ((n_parents >> group) << 1) > TOP_NODES) group++;
            TopInject ti{};
            for (int t = 0; t < group && status == LURKHIP_OK; t++) {
                if (inj_digests[l + t]) {  // hashed ahead
                    ti.dig[t] = inj_digests[l + t];
                    continue;
                }
                std::vector<int> inj;
                for (int m : order)
                    if (c->log_h[m] == lh - t) inj.push_back(m);
                if (inj.empty()) continue;
                LeafCol* tc = nullptr;
                status = make_cols(ctx, c, inj, &tc, &ti.w[t], scratch);
                ti.cols[t] = tc;
            }
            if (status == LURKHIP_OK) status = merkle_levels_coop(ctx, params, children, n_parents << 1, group, ti);
            l += group - 1;
            continue;
        }
        LeafCol* icols = nullptr;
        uint32_t iw = 0;
        if (!inject.empty()) status = make_cols(ctx, c, inject, &icols, &iw, scratch);
        if (status == LURKHIP_OK) status = merkle_level(ctx, params, children, n_parents, icols, iw, parents);
    }
    drop_scratch();
    if (status != LURKHIP_OK) return status;
    span_end(ctx, stage, span_level);
    return LURKHIP_OK;
}

}  // namespace

// Tree over matrices that are already on the device and are committed as they are (no LDE): FRI layers.
// The commitment does not own the matrices.
int32_t commit_raw(lurkhip_ctx* ctx, const std::vector<uint32_t*>& mats, const std::vector<int>& log_heights,
                   const std::vector<uint32_t>& widths, lurkhip_commitment** out) {
    lurkhip_commitment* c = new lurkhip_commitment();
    c->n_mats = (int)mats.size();
    c->lde = mats;
    c->owns_lde = false;
    c->coeffs.assign(mats.size(), nullptr);
    c->log_h = log_heights;
    c->width = widths;
    c->pitch = widths;
    int32_t s = build_tree(ctx, c);
    if (s != LURKHIP_OK) {
        (void)stream_wait(ctx);
        free_commitment(ctx, c);
        return s;
    }
    *out = c;
    return LURKHIP_OK;
}

int32_t commitment_build_tree(lurkhip_ctx* ctx, lurkhip_commitment* c) { return build_tree(ctx, c); }

int32_t commitment_root_m(lurkhip_ctx* ctx, const lurkhip_commitment* c, uint32_t* root_m) {
    void* pin = nullptr;
    LH_TRY(pinned_small(ctx, &pin));
    LH_HIP(ctx, hipMemcpyAsync(pin, c->digests + c->level_off[c->log_max] * 8, 32, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, stream_wait(ctx));
    memcpy(root_m, pin, 32);
    return LURKHIP_OK;
}

// the matrix as given, Montgomery form (lurkhip_mmcs_commit: no LDE)
__global__ void k_copy_words(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, bool to_monty) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = to_monty ? bb::to_monty(in[i]) : in[i];
}

// flags[c] = 1 when column c of a row-major device matrix holds a non-zero word (lurkhip_commit_dev_sparse): lanes along the row,
// a workgroup per block of rows, one plain store per non-zero column and workgroup
__global__ void k_columns_nonzero(const uint32_t* __restrict__ mat, uint32_t width, uint32_t pitch, size_t rows, uint32_t* __restrict__ flags) {
    const size_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (uint32_t c = threadIdx.x; c < width; c += blockDim.x) {
        uint32_t any = 0;
        for (size_t r = r0; r < r1; r++) any |= mat[r * pitch + c];
        if (any) flags[c] = 1u;
    }
}

// (commit.h) Layout of the prover's own traces ahead of their commitment: see the declaration.  Measured at the end of round 5 and
// OFF by default: with aligned sources k_lde_in<10,5> takes 1.650 ms per fib-mix proof against 1.603 dense (the line re-fetches of
// dense 78- / 148-word rows are served by L2: adjacent column chunks of a row run on neighbouring workgroups), while the writers
// lose -- a 64-row tile of a trace kernel is one contiguous run in a dense matrix and 64 runs ending mid-line in a pitched one
// (jit_trace_staged 0.94 -> 1.11 ms, jit_perm_rows 2.23 -> 2.28 ms).  LURKHIP_SRC_PADDED=1 turns the aligned layout on (at most
// half more memory per group), =2 pads every group whatever it costs; read on every call so that a test can switch it.
void plan_source_groups(int n, const uint32_t* log_heights, const uint32_t* widths, uint32_t* pitch, uint32_t* col_start, int32_t* group,
                        int32_t* n_groups) {
    const char* env = getenv("LURKHIP_SRC_PADDED");
    const int mode = env ? atoi(env) : 0;
    int32_t ng = 0;
    std::map<uint32_t, std::vector<int>> by_height;
    for (int i = 0; i < n; i++) {
        pitch[i] = widths[i];
        col_start[i] = 0;
        group[i] = -1;
        if (mode && lde_group_takes((int)log_heights[i])) by_height[log_heights[i]].push_back(i);
    }
    for (auto it = by_height.rbegin(); it != by_height.rend(); ++it) {
        uint32_t W = 0;
        for (int i : it->second) W += widths[i];
        const uint32_t Wp = (W + 31u) & ~31u;
        // (a source buffer is scratch of one proof and its padding is never read: up to half more memory is accepted here, where
        // the LDE buffers, which stay resident until the proof is out, stop at an eighth -- the 2^20 x 78 eval trace gets pitch 96)
        if ((Wp == W && it->second.size() == 1) || (mode == 1 && (Wp - W) * 2 > W)) continue;
        uint32_t at = 0;
        for (int i : it->second) {
            pitch[i] = Wp;
            col_start[i] = at;
            group[i] = ng;
            at += widths[i];
        }
        ng++;
    }
    for (int i = 0; i < n; i++)
        if (group[i] < 0) group[i] = ng++;
    *n_groups = ng;
}

int32_t commit_impl(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats, bool mats_on_host,
                    const uint32_t* log_heights, const uint32_t* widths, int32_t log_blowup, int32_t repr,
                    int32_t keep_coeffs, lurkhip_commitment** out, uint32_t* root, const uint32_t* shifts, bool raw, bool padded_groups,
                    const uint32_t* src_pitches, const std::vector<ColumnRuns>* live_runs, bool lde_only) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, n_mats > 0 && mats && log_heights && widths && out, "bad commit arguments");
    LH_ARG(ctx, !src_pitches || !mats_on_host, "row pitches are for device-resident matrices");
    LH_ARG(ctx, !live_runs || (!mats_on_host && (int)live_runs->size() == n_mats), "live column runs: one list per device-resident matrix");
    LH_ARG(ctx, log_blowup >= 0 && log_blowup <= 4, "log_blowup %d outside [0,4]", log_blowup);
    LH_ARG(ctx, repr == LURKHIP_REPR_CANONICAL || repr == LURKHIP_REPR_MONTY, "bad repr %d", repr);
    for (int i = 0; i < n_mats; i++) {
        LH_ARG(ctx, mats[i] != nullptr && widths[i] > 0, "matrix %d is empty", i);
        LH_ARG(ctx, (int)log_heights[i] + log_blowup <= bb::TWO_ADICITY, "matrix %d too tall", i);
        LH_ARG(ctx, !src_pitches || src_pitches[i] >= widths[i], "matrix %d: row pitch below its width", i);
    }
    LH_HIP(ctx, hipSetDevice(ctx->device));
    lurkhip_commitment* c = new lurkhip_commitment();
    c->n_mats = n_mats;
    c->log_blowup = log_blowup;
    c->lde.assign(n_mats, nullptr);
    c->coeffs.assign(n_mats, nullptr);
    c->log_h.resize(n_mats);
    c->width.assign(widths, widths + n_mats);
    c->pitch.assign(widths, widths + n_mats);
    c->lde_is_view.assign(n_mats, 0);
    c->group.assign(n_mats, -1);
    c->col_start.assign(n_mats, 0);
    std::vector<void*> uploads;  // host inputs staged in pooled device buffers (released, stream-ordered, once the LDEs are queued)
    auto fail = [&](int32_t s) {
        (void)stream_wait(ctx);
        if (ctx->hash_stream) (void)hipStreamSynchronize(ctx->hash_stream);  // (an early sponge may still be writing)
        for (void* u : uploads) pool_release(ctx, u);
        free_commitment(ctx, c);
        return s;
    };
#define TRY_C(expr)                          \
    do {                                     \
        int32_t s__ = (expr);                \
        if (s__ != LURKHIP_OK) return fail(s__); \
    } while (0)
#define HIP_C(expr)                                                                                     \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(set_error(ctx, e__ == hipErrorOutOfMemory ? LURKHIP_ERR_OOM : LURKHIP_ERR_HIP,  \
                                  "%s failed: %s", #expr, hipGetErrorString(e__)));                     \
    } while (0)

    span_begin(ctx, "lde");  // one span for the matrices of the commitment (with host inputs it includes their uploads)
    // Host inputs of an extending commitment are uploaded whole and then take the device route (round 4): one code path for the
    // LDE, and the caller's matrices of one height share tiles like the prover's do.
    std::vector<const uint32_t*> uploaded_mats;
    if (mats_on_host && !raw && log_blowup == 1 && !keep_coeffs && lde_group_enabled()) {
        uploaded_mats.resize(n_mats);
        for (int i = 0; i < n_mats; i++) {
            const size_t bytes = ((size_t)widths[i] << log_heights[i]) * sizeof(uint32_t);
            void* d = nullptr;
            TRY_C(pool_alloc(ctx, bytes, &d));
            uploads.push_back(d);
            HIP_C(hipMemcpyAsync(d, mats[i], bytes, hipMemcpyHostToDevice, ctx->stream));
            uploaded_mats[i] = (const uint32_t*)d;
        }
        mats = uploaded_mats.data();
        mats_on_host = false;
    }
    std::vector<char> extended(n_mats, 0);
    std::vector<size_t> coef_words(n_mats, 0);
    // Round 4: device-resident matrices of one height go through the grouped LDE (lde.hip) as ONE virtual row -- blow-up 2, nobody
    // keeping coefficients, 2^5 .. 2^20 rows.  A group holds up to LDE_MAX_MATS matrices of up to LDE_MAX_CLASSES coset shifts
    // (the quotient chunks of a height: shift w_Q^-c for chunk c), filled in shift order; tallest heights first.
    // Round 5: an entry of a group is a RUN of columns of a matrix (c0, w) -- the whole matrix unless the caller says which columns
    // are not identically zero (live_runs: the permutation traces; the other columns of those LDEs are zero-filled below).
    struct GroupPlan {
        int log_n;
        std::vector<int> idx;
        std::vector<uint32_t> c0, w, ostart;  // ostart: the run's first column in the group's output row (whole matrices side by side)
        uint32_t out_w = 0;
        std::vector<uint32_t> cls, shift_m;
        const uint32_t* scale[2][LDE_MAX_CLASSES];
    };
    std::vector<char> sparse(n_mats, 0);  // the matrix is extended run by run: its dead columns are zero-filled
    std::vector<GroupPlan> groups;
    std::vector<char> grouped(n_mats, 0);
    std::vector<std::pair<int, std::vector<int>>> height_members;  // per height (tallest first): its matrices in group order
    if (!mats_on_host && !raw && log_blowup == 1 && !keep_coeffs) {
        std::map<uint32_t, std::vector<int>> by_height;
        auto shift_of = [&](int i) { return bb::to_monty(shifts ? shifts[i] % bb::P : bb::GEN); };
        for (int i = 0; i < n_mats; i++) {
            if (!lde_group_takes((int)log_heights[i])) continue;
            by_height[log_heights[i]].push_back(i);
            if (!live_runs) continue;
            // run by run only when the matrix is sure to stay on this route: its two coset tables exist (they are cached: the groups
            // below find them again), and its runs are well-formed
            const uint32_t w_big = two_adic_generator_monty((int)log_heights[i] + 1);
            const uint32_t* t0 = nullptr;
            const uint32_t* t1 = nullptr;
            TRY_C(cached_scale_table(ctx, (int)log_heights[i], shift_of(i), &t0));
            TRY_C(cached_scale_table(ctx, (int)log_heights[i], bb::mul(shift_of(i), w_big), &t1));
            bool ok = t0 && t1;
            uint32_t at = 0;
            for (const auto& r : (*live_runs)[i]) {
                ok = ok && r.first >= at && r.second > 0 && r.first + r.second <= widths[i];
                at = r.first + r.second;
            }
            uint32_t covered = 0;
            for (const auto& r : (*live_runs)[i]) covered += r.second;
            sparse[i] = ok && covered < widths[i] && log_heights[i] > 10;  // (up to 2^10 rows an LDE is one kernel: nothing to leave out)
        }
        for (auto it = by_height.rbegin(); it != by_height.rend(); ++it) {
            std::vector<uint32_t> order;  // shifts in order of first appearance
            for (int i : it->second)
                if (std::find(order.begin(), order.end(), shift_of(i)) == order.end()) order.push_back(shift_of(i));
            std::vector<int> sorted;
            for (uint32_t sh : order)
                for (int i : it->second)
                    if (shift_of(i) == sh) sorted.push_back(i);
            height_members.push_back({(int)it->first, sorted});
            GroupPlan g{};
            g.log_n = (int)it->first;
            auto flush = [&]() {
                if (!g.idx.empty()) groups.push_back(g);
                g.idx.clear();
                g.c0.clear();
                g.w.clear();
                g.ostart.clear();
                g.out_w = 0;
                g.cls.clear();
                g.shift_m.clear();
            };
            for (int i : sorted) {
                const uint32_t sh = shift_of(i);
                // the matrix's entries: its live runs (the whole matrix unless it is extended run by run), preceded by an entry of
                // width 0 when its first columns are dead; never more than a launch holds -- the narrowest dead gaps are bridged
                // (their zeros are transformed like any other column) until the runs fit
                ColumnRuns runs{{0u, widths[i]}};
                if (sparse[i]) {
                    runs = (*live_runs)[i];
                    if (runs.empty() || runs[0].first != 0) runs.insert(runs.begin(), {0u, 0u});
                    while (runs.size() > (size_t)LDE_MAX_MATS) {
                        size_t best = 1;
                        for (size_t k = 1; k + 1 < runs.size(); k++)
                            if (runs[k + 1].first - (runs[k].first + runs[k].second) < runs[best + 1].first - (runs[best].first + runs[best].second)) best = k;
                        runs[best].second = runs[best + 1].first + runs[best + 1].second - runs[best].first;
                        runs.erase(runs.begin() + (long)best + 1);
                    }
                }
                size_t cl = std::find(g.shift_m.begin(), g.shift_m.end(), sh) - g.shift_m.begin();
                if (g.idx.size() + runs.size() > (size_t)LDE_MAX_MATS || (cl == g.shift_m.size() && cl == (size_t)LDE_MAX_CLASSES)) {
                    flush();
                    cl = 0;
                }
                if (cl == g.shift_m.size()) g.shift_m.push_back(sh);
                for (const auto& r : runs) {
                    g.idx.push_back(i);
                    g.c0.push_back(r.first);
                    g.w.push_back(r.second);
                    g.ostart.push_back(g.out_w + r.first);
                    g.cls.push_back((uint32_t)cl);
                }
                g.out_w += widths[i];
                if (sparse[i]) grouped[i] = 1;
            }
            flush();
        }
        // the groups' coset tables (cached per context); a group whose tables do not fit the cache keeps the old route
        std::vector<GroupPlan> kept;
        for (GroupPlan& g : groups) {
            const uint32_t w_big = two_adic_generator_monty(g.log_n + 1);
            bool ok = true;
            for (int q = 0; q < 2 && ok; q++)
                for (size_t cl = 0; cl < g.shift_m.size() && ok; cl++) {
                    const uint32_t s_q = q ? bb::mul(g.shift_m[cl], w_big) : g.shift_m[cl];
                    TRY_C(cached_scale_table(ctx, g.log_n, s_q, &g.scale[q][cl]));
                    ok = g.scale[q][cl] != nullptr;
                }
            if (!ok) {
                for (int i : g.idx)
                    if (sparse[i]) return fail(set_error(ctx, LURKHIP_ERR_EXEC, "a coset table vanished between two look-ups"));
                continue;
            }
            for (int i : g.idx) grouped[i] = 1;
            kept.push_back(g);
        }
        groups.swap(kept);
    }
    // Pitched sources (round 5: the prover's traces are column ranges of aligned group buffers, so that the LDE's first pass reads
    // whole lines): the grouped LDE takes the pitch; a matrix on any other route is first copied to a dense pooled buffer.
    std::vector<const uint32_t*> dense_mats;
    if (src_pitches) {
        bool any = false;
        for (int i = 0; i < n_mats; i++) any = any || (!grouped[i] && src_pitches[i] != widths[i]);
        if (any) {
            dense_mats.assign(mats, mats + n_mats);
            for (int i = 0; i < n_mats; i++) {
                if (grouped[i] || src_pitches[i] == widths[i]) continue;
                void* d = nullptr;
                TRY_C(pool_alloc(ctx, ((size_t)widths[i] << log_heights[i]) * sizeof(uint32_t), &d));
                uploads.push_back(d);
                HIP_C(hipMemcpy2DAsync(d, (size_t)widths[i] * 4, mats[i], (size_t)src_pitches[i] * 4, (size_t)widths[i] * 4, (size_t)1 << log_heights[i],
                                       hipMemcpyDeviceToDevice, ctx->stream));
                dense_mats[i] = (const uint32_t*)d;
            }
            mats = dense_mats.data();
        }
    }
    // Padded group buffers (the prover's own commitments): the LDEs of a group are column ranges of ONE buffer [2N][pitch], pitch =
    // the group's width rounded up to a 128-byte line, when that costs at most an eighth more memory -- the last LDE pass then
    // writes whole lines (a 32-column tile of a 92-word row straddles two lines on every row) and every reader takes the pitch.
    static const int pad_mode = getenv("LURKHIP_LDE_PADDED") ? atoi(getenv("LURKHIP_LDE_PADDED")) : 1;
    if (padded_groups && pad_mode)
        for (const auto& hm : height_members) {
            std::vector<int> members;
            for (int i : hm.second)
                if (grouped[i]) members.push_back(i);
            if (members.empty()) continue;
            uint32_t W = 0;
            for (int i : members) W += widths[i];
            const uint32_t Wp = (W + 31u) & ~31u;
            if (Wp == W && members.size() == 1) continue;
            // (lde_only: a rank's column tiles of a split commitment -- a ragged share of the group's columns, a G-th of its memory: always)
            if (pad_mode == 1 && !lde_only && (Wp - W) * 8 > W) continue;
            uint32_t* base = nullptr;
            TRY_C(pool_alloc(ctx, ((size_t)Wp << (hm.first + log_blowup)) * sizeof(uint32_t), (void**)&base));
            c->owned.push_back(base);
            const int gi = (int)c->group_base.size();
            c->group_base.push_back(base);
            uint32_t at = 0;
            for (int i : members) {
                c->lde[i] = base + at;
                c->lde_is_view[i] = 1;
                c->pitch[i] = Wp;
                c->group[i] = gi;
                c->col_start[i] = at;
                at += widths[i];
            }
        }
    for (int i = 0; i < n_mats; i++) {
        const size_t bytes = ((size_t)widths[i] << log_heights[i]) * sizeof(uint32_t);
        c->log_h[i] = (int)log_heights[i] + log_blowup;
        if (!c->lde_is_view[i]) TRY_C(pool_alloc(ctx, bytes << log_blowup, (void**)&c->lde[i]));
        if (grouped[i]) continue;  // no coefficient buffer: the grouped LDE never writes coefficients
        // coefficient buffer: room for the chunk-tiled form when nobody asked to keep (row-major) coefficients
        coef_words[i] = (!keep_coeffs && !mats_on_host && !raw && log_blowup >= 1) ? ntt_tiled_words((int)log_heights[i], (int)widths[i]) : 0;
        TRY_C(pool_alloc(ctx, std::max(bytes, coef_words[i] * 4), (void**)&c->coeffs[i]));
    }
    // Short matrices (below 2^13 rows: a few workgroups per NTT pass) go through their passes on the context's side lane, under
    // the tall matrices' passes; the tree needs all of them, so the lane is joined before it.  Only with device inputs and while
    // the coset-shift table cache has room (its fallback scratch, arena slot 2, is shared by the streams).
    SideLane lane(ctx);
    lane.want = 1;  // one side stream: the short matrices' passes share the coset-table scratch in order
    // LURKHIP_LDE_TALL_LANES=1 (A/B): the tall height groups alternate between the context's stream and a second side stream, so
    // that one group's first pass runs in the tail of another's last
    static const bool tall_lanes = getenv("LURKHIP_LDE_TALL_LANES") != nullptr && atoi(getenv("LURKHIP_LDE_TALL_LANES")) != 0;
    if (tall_lanes) lane.want = 2;
    constexpr uint32_t SIDE_MAX_LOG_N = 13;
    if (!mats_on_host && log_blowup >= 1 && ctx->lde_scale_bytes + ((size_t)64 << 20) < ((size_t)1 << 30)) TRY_C(lane.open());
    // The chain group: the height whose concatenated row is the longest run of permutations hashed sixteen lanes to the row (few
    // rows, thousands of columns: the hash chips).  Its LDE goes first and its row sponge starts at once on the hash stream
    // (early_sponge), under the LDE passes of every other group.  Opt-in (LURKHIP_EARLY_SPONGE=1): measured on a 2^12-row proof the
    // main commitment drops from 1.42 to 1.09 ms and the proof from 7.34 to 7.1-7.35 ms, but with two proofs in flight a step goes
    // from 5.0 to 5.8 ms (one more stream per context competing for the hardware queues), and the 2^20-row step does not move.
    static const int early_mode = getenv("LURKHIP_EARLY_SPONGE") ? atoi(getenv("LURKHIP_EARLY_SPONGE")) : 0;
    static const bool early_on = early_mode != 0;
    int chain_log_n = -1;
    if (early_on && !lde_only && lane.active && groups.size() >= 2) {
        std::map<int, uint32_t> width_of;  // per height, grouped matrices only
        std::map<int, bool> all_grouped;
        for (int i = 0; i < n_mats; i++) {
            width_of[(int)log_heights[i]] += widths[i];
            all_grouped[(int)log_heights[i]] = (all_grouped.count((int)log_heights[i]) ? all_grouped[(int)log_heights[i]] : true) && grouped[i];
        }
        uint32_t best = 0;
        for (const auto& kv : width_of)
            if (all_grouped[kv.first] && ((size_t)1 << (kv.first + log_blowup)) <= EARLY_SPONGE_MAX_ROWS && kv.second >= EARLY_SPONGE_MIN_WIDTH && kv.second > best) {
                best = kv.second;
                chain_log_n = kv.first;
            }
        uint32_t max_log_n = 0;
        for (int i = 0; i < n_mats; i++) max_log_n = std::max(max_log_n, log_heights[i]);
        // (LURKHIP_EARLY_SPONGE=2, A/B: the group with the most WORDS instead -- the tallest one of a big shard: its sponge, bound by
        // instruction issue, under the memory-bound first / last LDE passes of the other groups)
        if (early_mode == 2) {
            uint64_t most = 0;
            chain_log_n = -1;
            for (const auto& kv : width_of)
                if (all_grouped[kv.first] && ((uint64_t)kv.second << kv.first) > most) {
                    most = (uint64_t)kv.second << kv.first;
                    chain_log_n = kv.first;
                }
        }
        if (chain_log_n >= 0 && !tree_prehashes_rows((size_t)1 << (max_log_n + log_blowup))) chain_log_n = -1;
        if (chain_log_n >= 0) std::stable_partition(groups.begin(), groups.end(), [&](const GroupPlan& g) { return g.log_n == chain_log_n; });
    }
    for (size_t gi = 0; gi < groups.size(); gi++) {
        const GroupPlan& g = groups[gi];
        const bool short_group = (uint32_t)g.log_n < SIDE_MAX_LOG_N;
        const auto on_side = lane.on_side(short_group || (tall_lanes && lane.lanes >= 2 && (gi & 1)), short_group ? 0u : 1u);
        const uint32_t* ev[LDE_MAX_MATS];
        uint32_t* ld[LDE_MAX_MATS];
        uint32_t gw[LDE_MAX_MATS], gp[LDE_MAX_MATS], gs[LDE_MAX_MATS];
        uint32_t live_w = 0;
        for (uint32_t w : g.w) live_w += w;
        const bool dead_columns = live_w != g.out_w;
        for (size_t m = 0; m < g.idx.size(); m++) {
            ev[m] = mats[g.idx[m]] + g.c0[m];
            ld[m] = c->lde[g.idx[m]] + g.c0[m];
            gw[m] = g.w[m];
            gp[m] = c->pitch[g.idx[m]];
            gs[m] = src_pitches ? src_pitches[g.idx[m]] : widths[g.idx[m]];
        }
        TRY_C(lde_group(ctx, g.log_n, (int)g.idx.size(), ev, gw, ld, g.cls.data(), (int)g.shift_m.size(), g.scale, repr == LURKHIP_REPR_CANONICAL, false, gp, gs,
                        dead_columns ? g.ostart.data() : nullptr, g.out_w));
        for (int i : g.idx) extended[i] = 1;
        if (g.log_n == chain_log_n && (gi + 1 == groups.size() || groups[gi + 1].log_n != chain_log_n)) {  // the chain group's last plan is queued
            TRY_C(hash_stream_of(ctx));
            HIP_C(hipEventRecord(ctx->hash_ready, ctx->stream));  // (on the stream its passes were queued on)
            TRY_C(early_sponge(ctx, c, chain_log_n + log_blowup));
        }
    }
    if (!mats_on_host && log_blowup >= 1) {
        // device-resident matrices of one shape go through the passes together
        std::map<std::pair<uint32_t, uint32_t>, std::vector<int>> shapes;
        for (int i = 0; i < n_mats; i++)
            if (!grouped[i]) shapes[{log_heights[i], widths[i]}].push_back(i);
        // tallest first
        for (auto it = shapes.rbegin(); it != shapes.rend(); ++it) {
            const auto& kv = *it;
            const auto on_side = lane.on_side(kv.first.first < SIDE_MAX_LOG_N);
            const std::vector<int>& idx = kv.second;
            for (size_t at = 0; at < idx.size(); at += NTT_MAX_BATCH) {
                const int nb = (int)std::min<size_t>(NTT_MAX_BATCH, idx.size() - at);
                const uint32_t* ev[NTT_MAX_BATCH];
                uint32_t *co[NTT_MAX_BATCH], *ld[NTT_MAX_BATCH];
                uint32_t sh[NTT_MAX_BATCH];
                for (int m = 0; m < nb; m++) {
                    const int i = idx[at + m];
                    ev[m] = mats[i];
                    co[m] = c->coeffs[i];
                    ld[m] = c->lde[i];
                    sh[m] = bb::to_monty(shifts ? shifts[i] % bb::P : bb::GEN);
                }
                bool done = false;
                TRY_C(lde_batch(ctx, (int)kv.first.first, (int)kv.first.second, log_blowup, nb, ev, repr == LURKHIP_REPR_CANONICAL, co, ld, sh, &done,
                                coef_words[idx[at]], keep_coeffs != 0));
                if (done)
                    for (int m = 0; m < nb; m++) extended[idx[at + m]] = 1;
            }
        }
    }
    for (int i = 0; i < n_mats; i++) {
        const int log_n = (int)log_heights[i];
        const auto on_side = lane.on_side((uint32_t)log_n < SIDE_MAX_LOG_N);
        const int w = (int)widths[i];
        const size_t n = (size_t)1 << log_n;
        const size_t bytes = n * w * sizeof(uint32_t);
        uint32_t* coef = c->coeffs[i];
        if (raw) {  // FieldMerkleTreeMmcs::commit on the matrices as they are
            const uint32_t* src = mats[i];
            if (mats_on_host) {
                void* staged = nullptr;
                TRY_C(arena_get(ctx, 0, bytes, &staged));
                HIP_C(hipMemcpyAsync(staged, mats[i], bytes, hipMemcpyHostToDevice, ctx->stream));
                src = (const uint32_t*)staged;
            }
            const size_t words = n * (size_t)w;
            hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<size_t>((words + 255) / 256, 65535)), dim3(256), 0, ctx->stream, src, c->lde[i],
                               words, repr == LURKHIP_REPR_CANONICAL);
            HIP_C(hipGetLastError());
        } else if (!extended[i]) {
            const uint32_t* src = mats[i];
            void* staged = nullptr;
            if (mats_on_host) {
                TRY_C(arena_get(ctx, 0, bytes, &staged));
                HIP_C(hipMemcpyAsync(staged, mats[i], bytes, hipMemcpyHostToDevice, ctx->stream));
                src = (const uint32_t*)staged;
            }
            void* scratch = nullptr;
            void* row_scale = nullptr;
            // the LDE buffer doubles as interpolation scratch when it is big enough (blow-up >= 1)
            if (log_blowup >= 1) scratch = c->lde[i];
            else TRY_C(arena_get(ctx, 1, bytes, &scratch));
            TRY_C(arena_get(ctx, 2, n * sizeof(uint32_t), &row_scale));
            TRY_C(interpolate(ctx, log_n, w, src, repr == LURKHIP_REPR_CANONICAL, (uint32_t*)scratch, coef));
            TRY_C(extend(ctx, log_n, w, log_blowup, coef, c->lde[i], (uint32_t*)row_scale, false,
                         bb::to_monty(shifts ? shifts[i] % bb::P : bb::GEN)));
        }
        if (!keep_coeffs) {
            pool_release(ctx, coef);  // stream-ordered: only later work can reuse it
            c->coeffs[i] = nullptr;
        }
    }
    TRY_C(lane.close());
    for (void* u : uploads) pool_release(ctx, u);
    uploads.clear();
    span_end(ctx, "lde");
    if (lde_only) {
        *out = c;
        return LURKHIP_OK;
    }
    TRY_C(build_tree(ctx, c));
    if (root) {
        uint32_t r[8];
        const uint32_t* droot = c->digests + c->level_off[c->log_max] * 8;
        HIP_C(hipMemcpyAsync(r, droot, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
        HIP_C(stream_wait(ctx));
        for (int k = 0; k < 8; k++) root[k] = repr == LURKHIP_REPR_CANONICAL ? bb::from_monty(r[k]) : r[k];
    }
#undef TRY_C
#undef HIP_C
    *out = c;
    return LURKHIP_OK;
}

// the runs of columns of device matrices that hold a non-zero word (one pass over the matrices, one wait)
int32_t nonzero_column_runs(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev, const uint32_t* log_heights, const uint32_t* widths,
                            const uint32_t* pitches, std::vector<ColumnRuns>* runs, uint32_t* zero_columns) {
    LH_HIP(ctx, hipSetDevice(ctx->device));
    size_t total_w = 0;
    for (int i = 0; i < n_mats; i++) {
        LH_ARG(ctx, mats_dev[i] != nullptr && widths[i] > 0 && log_heights[i] < 31, "matrix %d is empty", i);
        total_w += widths[i];
    }
    void* flags_dev = nullptr;
    LH_TRY(pool_alloc(ctx, total_w * 4, &flags_dev));
    std::vector<uint32_t> flags(total_w, 0);
    hipError_t e = hipMemsetAsync(flags_dev, 0, total_w * 4, ctx->stream);
    size_t at = 0;
    for (int i = 0; i < n_mats && e == hipSuccess; i++) {
        const size_t rows = (size_t)1 << log_heights[i];
        const unsigned blocks = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, rows / 64));
        hipLaunchKernelGGL(k_columns_nonzero, dim3(blocks), dim3(256), 0, ctx->stream, mats_dev[i], widths[i], pitches ? pitches[i] : widths[i], rows,
                           (uint32_t*)flags_dev + at);
        e = hipGetLastError();
        at += widths[i];
    }
    if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), flags_dev, total_w * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = stream_wait(ctx);
    pool_release(ctx, flags_dev);
    if (e != hipSuccess) return set_error(ctx, LURKHIP_ERR_HIP, "column scan failed: %s", hipGetErrorString(e));
    runs->assign(n_mats, {});
    uint32_t n_zero = 0;
    at = 0;
    for (int i = 0; i < n_mats; i++) {
        for (uint32_t c = 0; c < widths[i]; c++) {
            if (!flags[at + c]) {
                n_zero++;
                continue;
            }
            ColumnRuns& r = (*runs)[i];
            if (!r.empty() && r.back().first + r.back().second == c) r.back().second++;
            else r.push_back({c, 1u});
        }
        at += widths[i];
    }
    if (zero_columns) *zero_columns = n_zero;
    return LURKHIP_OK;
}

}  // namespace lurkhip

using namespace lurkhip;

extern "C" {

int32_t lurkhip_coset_lde_dev(lurkhip_ctx* ctx, int32_t log_n, int32_t width, int32_t log_blowup, const uint32_t* in,
                              uint32_t* out, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, log_n >= 0 && log_blowup >= 0 && log_n + log_blowup <= bb::TWO_ADICITY && width > 0, "bad LDE shape");
    LH_ARG(ctx, in && out, "null buffer");
    LH_ARG(ctx, repr == LURKHIP_REPR_CANONICAL || repr == LURKHIP_REPR_MONTY, "bad repr %d", repr);
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_n;
    const size_t bytes = n * width * sizeof(uint32_t);
    if (log_blowup == 1 && lde_group_takes(log_n)) {  // the grouped route (lde.hip) on a group of one
        const uint32_t gen_m = bb::to_monty(bb::GEN);
        const uint32_t* scale[2][LDE_MAX_CLASSES] = {};
        LH_TRY(cached_scale_table(ctx, log_n, gen_m, &scale[0][0]));
        LH_TRY(cached_scale_table(ctx, log_n, bb::mul(gen_m, two_adic_generator_monty(log_n + 1)), &scale[1][0]));
        if (scale[0][0] && scale[1][0]) {
            const uint32_t w = (uint32_t)width, cls = 0;
            return lde_group(ctx, log_n, 1, &in, &w, &out, &cls, 1, scale, repr == LURKHIP_REPR_CANONICAL, repr == LURKHIP_REPR_CANONICAL);
        }
    }
    void *coef = nullptr, *scratch = nullptr, *row_scale = nullptr;
    LH_TRY(arena_get(ctx, 1, bytes, &coef));
    LH_TRY(arena_get(ctx, 3, bytes, &scratch));
    LH_TRY(arena_get(ctx, 2, n * sizeof(uint32_t), &row_scale));
    const bool canon = repr == LURKHIP_REPR_CANONICAL;
    LH_TRY(interpolate(ctx, log_n, width, in, canon, (uint32_t*)scratch, (uint32_t*)coef));
    LH_TRY(extend(ctx, log_n, width, log_blowup, (const uint32_t*)coef, out, (uint32_t*)row_scale, canon, bb::to_monty(bb::GEN)));
    return LURKHIP_OK;
}

int32_t lurkhip_coset_lde(lurkhip_ctx* ctx, int32_t log_n, int32_t width, int32_t log_blowup, const uint32_t* in,
                          uint32_t* out, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, log_n >= 0 && log_blowup >= 0 && log_n + log_blowup <= bb::TWO_ADICITY && width > 0, "bad LDE shape");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = ((size_t)1 << log_n) * width * sizeof(uint32_t);
    void *din = nullptr, *dout = nullptr;
    LH_HIP(ctx, hipMalloc(&din, bytes));
    hipError_t e = hipMalloc(&dout, bytes << log_blowup);
    if (e != hipSuccess) {
        (void)hipFree(din);
        return set_error(ctx, LURKHIP_ERR_OOM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    int32_t s = LURKHIP_OK;
    if (hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        s = set_error(ctx, LURKHIP_ERR_HIP, "H2D copy failed");
    if (s == LURKHIP_OK) s = lurkhip_coset_lde_dev(ctx, log_n, width, log_blowup, (const uint32_t*)din, (uint32_t*)dout, repr);
    if (s == LURKHIP_OK && hipMemcpyAsync(out, dout, bytes << log_blowup, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        s = set_error(ctx, LURKHIP_ERR_HIP, "D2H copy failed");
    (void)stream_wait(ctx);
    (void)hipFree(din);
    (void)hipFree(dout);
    return s;
}

int32_t lurkhip_commit_dev(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev, const uint32_t* log_heights,
                           const uint32_t* widths, int32_t log_blowup, int32_t repr, int32_t keep_coeffs,
                           lurkhip_commitment** out, uint32_t* root) {
    return commit_impl(ctx, n_mats, mats_dev, false, log_heights, widths, log_blowup, repr, keep_coeffs, out, root);
}

int32_t lurkhip_commit_dev_sparse(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev, const uint32_t* log_heights,
                                  const uint32_t* widths, int32_t log_blowup, int32_t repr, int32_t aligned_groups,
                                  lurkhip_commitment** out, uint32_t* root, uint32_t* zero_columns) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, n_mats > 0 && mats_dev && log_heights && widths && out, "bad commit arguments");
    std::vector<ColumnRuns> runs;
    LH_TRY(nonzero_column_runs(ctx, n_mats, mats_dev, log_heights, widths, nullptr, &runs, zero_columns));
    return commit_impl(ctx, n_mats, mats_dev, false, log_heights, widths, log_blowup, repr, 0, out, root, nullptr, false, aligned_groups != 0, nullptr, &runs);
}

int32_t lurkhip_mmcs_commit(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats, const uint32_t* log_heights, const uint32_t* widths,
                            int32_t repr, lurkhip_commitment** out, uint32_t* root) {
    return commit_impl(ctx, n_mats, mats, true, log_heights, widths, 0, repr, 0, out, root, nullptr, /*raw=*/true);
}

int32_t lurkhip_commit_cosets_dev(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev, const uint32_t* log_heights,
                                  const uint32_t* widths, const uint32_t* shifts, int32_t log_blowup, int32_t repr,
                                  lurkhip_commitment** out, uint32_t* root) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, shifts != nullptr, "null shifts");
    return commit_impl(ctx, n_mats, mats_dev, false, log_heights, widths, log_blowup, repr, 0, out, root, shifts);
}

int32_t lurkhip_commit(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats, const uint32_t* log_heights,
                       const uint32_t* widths, int32_t log_blowup, int32_t repr, int32_t keep_coeffs,
                       lurkhip_commitment** out, uint32_t* root) {
    return commit_impl(ctx, n_mats, mats, true, log_heights, widths, log_blowup, repr, keep_coeffs, out, root);
}

int32_t lurkhip_commitment_free(lurkhip_ctx* ctx, lurkhip_commitment* c) {
    LH_CHECK_CTX_NOLOCK(ctx);
    if (!c) return LURKHIP_OK;
    free_commitment(ctx, c);
    return LURKHIP_OK;
}

int32_t lurkhip_commitment_matrix_pitch(lurkhip_ctx* ctx, lurkhip_commitment* c, int32_t index, uint32_t* pitch_words) {
    LH_CHECK_CTX_NOLOCK(ctx);
    LH_ARG(ctx, c && pitch_words && index >= 0 && index < c->n_mats, "bad matrix index %d", index);
    *pitch_words = c->pitch[index];
    return LURKHIP_OK;
}

int32_t lurkhip_commitment_matrix_dev(lurkhip_ctx* ctx, lurkhip_commitment* c, int32_t index, const uint32_t** lde_dev,
                                      uint32_t* log_height, uint32_t* width) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, c && index >= 0 && index < c->n_mats, "bad matrix index %d", index);
    if (lde_dev) *lde_dev = c->lde[index];
    if (log_height) *log_height = (uint32_t)c->log_h[index];
    if (width) *width = c->width[index];
    return LURKHIP_OK;
}

int32_t lurkhip_commitment_root(lurkhip_ctx* ctx, lurkhip_commitment* c, uint32_t* root, int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, c && root, "null argument");
    uint32_t r[8];
    LH_HIP(ctx, hipMemcpyAsync(r, c->digests + c->level_off[c->log_max] * 8, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, stream_wait(ctx));
    for (int k = 0; k < 8; k++) root[k] = repr == LURKHIP_REPR_CANONICAL ? bb::from_monty(r[k]) : r[k];
    return LURKHIP_OK;
}

// Opens leaf `index` (a row index of the tallest LDE): writes the opened row of every matrix (in the
// caller's matrix order, matrix m at row index >> (log_max - log_h[m])) back to back into `rows`, and
// the log_max sibling digests (leaf level first) into `path`.
int32_t lurkhip_commitment_open(lurkhip_ctx* ctx, lurkhip_commitment* c, uint64_t index, uint32_t* rows, uint32_t* path,
                                int32_t repr) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, c && rows && path, "null argument");
    LH_ARG(ctx, index < ((uint64_t)1 << c->log_max), "leaf index out of range");
    size_t off = 0;
    for (int m = 0; m < c->n_mats; m++) {
        uint64_t r = index >> (c->log_max - c->log_h[m]);
        LH_HIP(ctx, hipMemcpyAsync(rows + off, c->lde[m] + r * c->pitch[m], c->width[m] * 4, hipMemcpyDeviceToHost, ctx->stream));
        off += c->width[m];
    }
    for (int l = 0; l < c->log_max; l++) {
        uint64_t sib = (index >> l) ^ 1;
        LH_HIP(ctx, hipMemcpyAsync(path + l * 8, c->digests + (c->level_off[l] + sib) * 8, 32, hipMemcpyDeviceToHost, ctx->stream));
    }
    LH_HIP(ctx, stream_wait(ctx));
    if (repr == LURKHIP_REPR_CANONICAL) {
        for (size_t i = 0; i < off; i++) rows[i] = bb::from_monty(rows[i]);
        for (int i = 0; i < c->log_max * 8; i++) path[i] = bb::from_monty(path[i]);
    }
    return LURKHIP_OK;
}

}  // extern "C"
