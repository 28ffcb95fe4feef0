// One shard over G = 2^log_g ranks: the commitment (second bullet; collective C2 of SURVEY.md 7).
//
// What the reference does in one process -- p3 `TwoAdicFriPcs::commit` of a shard's matrices inside sphinx's `prove_shard`
// (/root/reference/benches/fib.rs:124; /root/reference/src/lair/execute.rs:231-241 makes anything below 2^22 rows one shard) --
// done by all ranks together, producing the root one rank would:
//   1. (exchange A, only for matrices whose rows were computed by row blocks) rows -> column tiles;
//   2. coset LDE of this rank's column tile of every height group (commit_impl, lde_only: the kernels of lde.hip);
//   3. (main traces) next-row copies of the columns the AIRs read on the next row, made by the rank that holds the column;
//   4. exchange B: ONE all-to-all, column tiles -> blocks of contiguous storage rows;
//   5. leaf sponges and tree levels over the rank's rows (build_tree: a subtree of the global tree, because a 2^k-aligned block
//      of storage rows of every matrix is what hangs below one node);
//   6. all-gather of the G subtree roots; the top log2 G levels -- with the matrices of at most G rows injected -- on the host of
//      every rank.
// The plan (who sends which block where) is split_plan.h; the copies between matrices and the linear send / receive buffers are
// one kernel launch per list.
#include "split.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>

#include "babybear.h"
#include "challenger.h"
#include "ctx.h"

namespace lurkhip {

namespace {

struct CopyJobDev {
    const uint32_t* src;
    uint32_t* dst;
    uint32_t spitch, dpitch, width, rows;
    uint32_t first_block, pad;
};
constexpr uint32_t COPY_PER_BLOCK = 256 * 8;

// block b copies words [b - first_block) * COPY_PER_BLOCK .. of its job: dst[r * dpitch + c] = src[r * spitch + c]
__global__ __launch_bounds__(256) void k_copy_jobs(const CopyJobDev* __restrict__ jobs, uint32_t n_jobs) {
    uint32_t lo = 0, hi = n_jobs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs[mid].first_block <= blockIdx.x) lo = mid;
        else hi = mid;
    }
    const CopyJobDev j = jobs[lo];
    const uint64_t total = (uint64_t)j.rows * j.width;
    uint64_t idx = (uint64_t)(blockIdx.x - j.first_block) * COPY_PER_BLOCK + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; k++, idx += 256) {
        if (idx >= total) break;
        const uint32_t r = (uint32_t)(idx / j.width), c = (uint32_t)(idx - (uint64_t)r * j.width);
        j.dst[(size_t)r * j.dpitch + c] = j.src[(size_t)r * j.spitch + c];
    }
}

// out[s] = col[s_next * pitch], s_next = the storage row of the quotient domain's next row (stark_kernels.h: quotient_body);
// rows outside the quotient domain (s >= Q) are never read: zero
__global__ __launch_bounds__(256) void k_next_rows(const uint32_t* __restrict__ col, uint32_t pitch, uint32_t log_q, uint32_t qd, uint32_t n_rows,
                                                    uint32_t* __restrict__ out) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_rows) return;
    const uint32_t q = 1u << log_q;
    if (s >= q) {
        out[s] = 0u;
        return;
    }
    const uint32_t i = log_q ? (__brev(s) >> (32 - log_q)) : 0u;
    const uint32_t i_next = (i + qd) & (q - 1);
    const uint32_t s_next = log_q ? (__brev(i_next) >> (32 - log_q)) : 0u;
    out[s] = col[(size_t)s_next * pitch];
}

__global__ __launch_bounds__(256) void k_add_ef(uint32_t* __restrict__ data, size_t stride_words, size_t n, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint4* p = reinterpret_cast<uint4*>(data + r * stride_words);
    const uint4 v = *p;
    *p = make_uint4(bb::add(v.x, a), bb::add(v.y, b), bb::add(v.z, c), bb::add(v.w, d));
}

struct HostHasher {  // p3 PaddingFreeSponge<16, 8, 8> / TruncatedPermutation<16, 2, 8> on the host (the twin of verify.cpp's)
    const P16Params& p;
    void sponge(const std::vector<std::pair<const uint32_t*, uint32_t>>& rows, uint32_t out[8]) const {
        uint32_t s[16] = {};
        int pos = 0;
        for (const auto& r : rows)
            for (uint32_t k = 0; k < r.second; k++) {
                s[pos++] = r.first[k];
                if (pos == 8) {
                    host_perm16(p, s);
                    pos = 0;
                }
            }
        if (pos) host_perm16(p, s);
        memcpy(out, s, 32);
    }
    void compress(const uint32_t* l, const uint32_t* r, uint32_t out[8]) const {
        uint32_t s[16];
        memcpy(s, l, 32);
        memcpy(s + 8, r, 32);
        host_perm16(p, s);
        memcpy(out, s, 32);
    }
};

// resolves a plan's job list against the buffers and runs it as one launch
struct BufRef {
    uint32_t* base;
    uint32_t pitch;
};
int32_t run_jobs(lurkhip_ctx* ctx, const std::vector<split::Job>& jobs, const std::vector<BufRef>& bufs, uint32_t* lin, bool pack, std::vector<void*>& scratch) {
    std::vector<CopyJobDev> dev;
    uint32_t blocks = 0;
    for (const split::Job& j : jobs) {
        if (j.rows == 0 || j.width == 0) continue;
        const BufRef& b = bufs[(size_t)j.buf];
        if (!b.base) return set_error(ctx, LURKHIP_ERR_EXEC, "split: a copy job refers to a buffer this rank does not hold");
        uint32_t* mat = b.base + (size_t)j.row0 * b.pitch + j.col0;
        const uint32_t mpitch = b.pitch * j.row_stride;
        CopyJobDev d{};
        d.src = pack ? mat : lin + j.lin_off;
        d.dst = pack ? lin + j.lin_off : mat;
        d.spitch = pack ? mpitch : j.lin_pitch;
        d.dpitch = pack ? j.lin_pitch : mpitch;
        d.width = j.width;
        d.rows = j.rows;
        d.first_block = blocks;
        blocks += (uint32_t)(((uint64_t)j.rows * j.width + COPY_PER_BLOCK - 1) / COPY_PER_BLOCK);
        dev.push_back(d);
    }
    if (dev.empty()) return LURKHIP_OK;
    void* tbl = nullptr;
    LH_TRY(pool_alloc(ctx, dev.size() * sizeof(CopyJobDev), &tbl));
    scratch.push_back(tbl);
    static_assert(sizeof(CopyJobDev) % 4 == 0, "uploaded as words");
    LH_TRY(upload_words(ctx, (uint32_t*)tbl, (const uint32_t*)dev.data(), dev.size() * sizeof(CopyJobDev) / 4));
    hipLaunchKernelGGL(k_copy_jobs, dim3(blocks), dim3(256), 0, ctx->stream, (const CopyJobDev*)tbl, (uint32_t)dev.size());
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

}  // namespace

int32_t split_env_init(lurkhip_ctx* ctx, const lurkhip_split_comm* comm, int32_t min_log_n, SplitEnv* out) {
    LH_ARG(ctx, comm && comm->alltoallv_dev && comm->allgather_dev && comm->allgather_host && comm->allreduce_sum_u64_host, "split: incomplete communicator");
    int log_g = 0;
    while ((1 << log_g) < comm->world) log_g++;
    LH_ARG(ctx, comm->world >= 2 && comm->world <= 64 && (1 << log_g) == comm->world, "split: the number of ranks (%d) must be a power of two from 2 to 64", comm->world);
    LH_ARG(ctx, comm->rank >= 0 && comm->rank < comm->world, "split: rank %d of %d", comm->rank, comm->world);
    LH_ARG(ctx, min_log_n >= log_g && min_log_n <= 27, "split: chips are cut from 2^%d rows up, which is below the number of ranks (or absurd)", min_log_n);
    out->comm = *comm;
    out->log_g = log_g;
    out->rank = comm->rank;
    out->min_log_n = min_log_n;
    return LURKHIP_OK;
}

int32_t split_allgather_host(lurkhip_ctx* ctx, const SplitEnv& env, const void* send, void* recv, uint64_t bytes_per_rank) {
    const int32_t st = env.comm.allgather_host(env.comm.user, send, recv, bytes_per_rank);
    return st == 0 ? LURKHIP_OK : set_error(ctx, LURKHIP_ERR_EXEC, "split: the host all-gather failed on rank %d (status %d)", env.rank, st);
}
int32_t split_allreduce_u64_host(lurkhip_ctx* ctx, const SplitEnv& env, uint64_t* buf, uint64_t n) {
    const int32_t st = env.comm.allreduce_sum_u64_host(env.comm.user, buf, n);
    return st == 0 ? LURKHIP_OK : set_error(ctx, LURKHIP_ERR_EXEC, "split: the host all-reduce failed on rank %d (status %d)", env.rank, st);
}
int32_t split_allgather_dev(lurkhip_ctx* ctx, const SplitEnv& env, const uint32_t* send_dev, uint32_t* recv_dev, uint64_t words_per_rank) {
    const int32_t st = env.comm.allgather_dev(env.comm.user, send_dev, recv_dev, words_per_rank, (void*)ctx->stream);
    return st == 0 ? LURKHIP_OK : set_error(ctx, LURKHIP_ERR_EXEC, "split: the device all-gather failed on rank %d (status %d)", env.rank, st);
}
int32_t add_ef_to_column(lurkhip_ctx* ctx, uint32_t* data, size_t stride_words, size_t n, const uint32_t o[4]) {
    if (n == 0) return LURKHIP_OK;
    hipLaunchKernelGGL(k_add_ef, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, data, stride_words, n, o[0], o[1], o[2], o[3]);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t split_commit(lurkhip_ctx* ctx, const SplitEnv& env, int n, const SplitMat* mats, int log_blowup, lurkhip_commitment** out, uint32_t* root_m) {
    LH_ARG(ctx, env.on() && n > 0 && mats && out, "split: bad commit arguments");
    LH_ARG(ctx, log_blowup == 1, "split: blow-up 2 only");
    const int log_g = env.log_g, G = 1 << log_g, rank = env.rank;
    std::vector<split::MatDesc> descs((size_t)n);
    for (int i = 0; i < n; i++) {
        const SplitMat& m = mats[i];
        LH_ARG(ctx, m.width > 0 && m.pitch >= m.width && m.log_n + 1 <= (uint32_t)bb::TWO_ADICITY, "split: matrix %d has a bad shape", i);
        LH_ARG(ctx, (int)m.log_n >= env.min_log_n || m.kind == split::K_FULL, "split: matrix %d is below the cut and must be whole on every rank", i);
        descs[(size_t)i] = split::MatDesc{m.log_n, m.width, m.kind, m.lqd, m.chunk, (int)m.log_n >= env.min_log_n ? m.n_next : 0u, m.next_lqd, {}};
        if (m.runs && (int)m.log_n >= env.min_log_n) descs[(size_t)i].runs = *m.runs;
    }
    split::Plan plan;
    try {
        plan = split::make_plan(log_g, rank, env.min_log_n, descs);
    } catch (const std::exception& e) {
        return set_error(ctx, LURKHIP_ERR_INVALID_ARG, "%s", e.what());
    }
    const P16Params* params_dev = nullptr;
    LH_TRY(get_merkle_params(ctx, &params_dev));
    const HostHasher H{*(const P16Params*)ctx->merkle_params_host};

    std::vector<void*> scratch;              // released (stream-ordered) on every way out
    lurkhip_commitment* c = new lurkhip_commitment();
    auto done = [&](int32_t s) {
        if (s != LURKHIP_OK) (void)stream_wait(ctx);
        for (void* p : scratch) pool_release(ctx, p);
        scratch.clear();
        if (s != LURKHIP_OK && c) {
            free_commitment(ctx, c);
            c = nullptr;
        }
        return s;
    };
#define S_TRY(expr)                              \
    do {                                         \
        int32_t s__ = (expr);                    \
        if (s__ != LURKHIP_OK) return done(s__); \
    } while (0)
    auto salloc = [&](size_t words, uint32_t** p) -> int32_t {
        void* v = nullptr;
        const int32_t s = pool_alloc(ctx, std::max<size_t>(words, 4) * 4, &v);
        if (s == LURKHIP_OK) {
            scratch.push_back(v);
            *p = (uint32_t*)v;
        }
        return s;
    };
    span_begin(ctx, "split_exchange_a");
    // ---- 1. exchange A: rows -> column tiles
    std::vector<uint32_t*> slab(plan.groups.size(), nullptr);
    // (rows of whole 128-byte lines: a rank's share of a group's columns is ragged, and the LDE's first pass reads unaligned row
    // segments at half the rate -- lde.hip)
    auto slab_pitch = [&](size_t g) { return (plan.slab_w[g] + 31u) & ~31u; };
    if (plan.has_a) {
        uint32_t *send = nullptr, *recv = nullptr;
        S_TRY(salloc(plan.a_send_off.back(), &send));
        S_TRY(salloc(plan.a_recv_off.back(), &recv));
        std::vector<BufRef> src((size_t)n);
        for (int i = 0; i < n; i++) src[(size_t)i] = BufRef{const_cast<uint32_t*>(mats[i].src), mats[i].pitch};
        S_TRY(run_jobs(ctx, plan.a_pack, src, send, true, scratch));
        if (env.comm.alltoallv_dev(env.comm.user, send, plan.a_send_off.data(), recv, plan.a_recv_off.data(), (void*)ctx->stream) != 0)
            return done(set_error(ctx, LURKHIP_ERR_EXEC, "split: the all-to-all before the LDE failed on rank %d", rank));
        ctx->split_words_a += plan.a_send_off.back() - (plan.a_send_off[(size_t)rank + 1] - plan.a_send_off[(size_t)rank]);
        ctx->split_exchanges++;
        std::vector<BufRef> dst(plan.groups.size());
        for (size_t g = 0; g < plan.groups.size(); g++) {
            if (plan.slab_w[g]) S_TRY(salloc((size_t)slab_pitch(g) << plan.groups[g].log_n, &slab[g]));
            dst[g] = BufRef{slab[g], slab_pitch(g)};
        }
        S_TRY(run_jobs(ctx, plan.a_unpack, dst, recv, false, scratch));
    }
    span_end(ctx, "split_exchange_a");
    // ---- 2. the LDE of this rank's tiles, and of the matrices that are not cut (whole, on every rank), in ONE call: the short
    // matrices' passes -- latency-bound chains of a few workgroups -- go to the side lane under the tiles' (commit_impl)
    std::vector<BufRef> tile_out(plan.tiles.size() + plan.my_extras.size(), BufRef{nullptr, 0});
    std::vector<int> small;
    for (int i = 0; i < n; i++)
        if (plan.group_of[(size_t)i] < 0) small.push_back(i);
    const size_t nt = plan.tiles.size();
    if (nt + small.size() > 0) {
        std::vector<const uint32_t*> ptr(nt);
        std::vector<uint32_t> lh(nt), w(nt), sp(nt), sh(nt);
        for (int i : small) {
            ptr.push_back(mats[i].src);
            lh.push_back(mats[i].log_n);
            w.push_back(mats[i].width);
            sp.push_back(mats[i].pitch);
            sh.push_back(mats[i].shift ? mats[i].shift : bb::GEN);
        }
        for (size_t k = 0; k < nt; k++) {
            const split::Tile& t = plan.tiles[k];
            const SplitMat& m = mats[t.mat];
            if (m.kind == split::K_FULL) {
                ptr[k] = m.src + t.c0;
                sp[k] = m.pitch;
            } else {
                ptr[k] = slab[(size_t)t.group] + t.slab_col;
                sp[k] = slab_pitch((size_t)t.group);
            }
            lh[k] = m.log_n;
            w[k] = t.w;
            sh[k] = m.shift ? m.shift : bb::GEN;
        }
        // (the tiles' LDE buffers stay with the commitment -- c->aux owns them with the short matrices' -- until it is freed: 1 / G of
        // the LDE's memory more per commitment, against a second call whose short chains would run after the exchange instead of under it)
        const int32_t st = commit_impl(ctx, (int32_t)ptr.size(), ptr.data(), false, lh.data(), w.data(), log_blowup, LURKHIP_REPR_MONTY, 0, &c->aux, nullptr, sh.data(),
                                       false, /*padded_groups=*/true, sp.data(), nullptr, /*lde_only=*/true);
        if (st != LURKHIP_OK) return done(st);
        for (size_t k = 0; k < nt; k++) tile_out[k] = BufRef{c->aux->lde[k], c->aux->pitch[k]};
    }
    span_begin(ctx, "split_exchange_b");
    // ---- 3. next-row copies of the columns this rank holds
    for (size_t k = 0; k < plan.my_extras.size(); k++) {
        const split::Group& g = plan.groups[(size_t)(plan.my_extras[k] >> 16)];
        const split::Extra& e = g.extras[(size_t)(plan.my_extras[k] & 0xffff)];
        const SplitMat& m = mats[e.mat];
        size_t tk = 0;  // the tile that holds column e.col of the matrix
        for (; tk < plan.tiles.size(); tk++)
            if (plan.tiles[tk].mat == e.mat && plan.tiles[tk].c0 <= e.col && e.col < plan.tiles[tk].c0 + plan.tiles[tk].w) break;
        if (tk == plan.tiles.size()) return done(set_error(ctx, LURKHIP_ERR_EXEC, "split: internal error: a next-row column without its tile"));
        const uint32_t rows = 2u << m.log_n;
        uint32_t* col = nullptr;
        S_TRY(salloc(rows, &col));
        hipLaunchKernelGGL(k_next_rows, dim3((rows + 255) / 256), dim3(256), 0, ctx->stream, tile_out[tk].base + (e.col - plan.tiles[tk].c0), tile_out[tk].pitch,
                           m.log_n + m.next_lqd, 1u << m.next_lqd, rows, col);
        tile_out[plan.tiles.size() + k] = BufRef{col, 1};
    }
    if (hipGetLastError() != hipSuccess) return done(set_error(ctx, LURKHIP_ERR_HIP, "split: k_next_rows launch failed"));
    // ---- 4. exchange B: column tiles -> row blocks
    std::vector<uint32_t*> local(plan.groups.size(), nullptr);
    {
        uint32_t *send = nullptr, *recv = nullptr;
        S_TRY(salloc(plan.b_send_off.back(), &send));
        S_TRY(salloc(plan.b_recv_off.back(), &recv));
        S_TRY(run_jobs(ctx, plan.b_pack, tile_out, send, true, scratch));
        if (env.comm.alltoallv_dev(env.comm.user, send, plan.b_send_off.data(), recv, plan.b_recv_off.data(), (void*)ctx->stream) != 0)
            return done(set_error(ctx, LURKHIP_ERR_EXEC, "split: the all-to-all after the LDE failed on rank %d", rank));
        ctx->split_words_b += plan.b_send_off.back() - (plan.b_send_off[(size_t)rank + 1] - plan.b_send_off[(size_t)rank]);
        ctx->split_exchanges++;
        std::vector<BufRef> dst(plan.groups.size());
        for (size_t g = 0; g < plan.groups.size(); g++) {
            const size_t words = (size_t)plan.groups[g].local_pitch * ((size_t)(2u << plan.groups[g].log_n) >> log_g);
            void* v = nullptr;
            S_TRY(pool_alloc(ctx, words * 4, &v));
            c->owned.push_back(v);
            local[g] = (uint32_t*)v;
            // (columns no rank sends -- the dead ones every rank agreed on -- are the zeros of the LDE of a zero column)
            if (plan.groups[g].sparse && hipMemsetAsync(v, 0, words * 4, ctx->stream) != hipSuccess) return done(set_error(ctx, LURKHIP_ERR_HIP, "split: zero-fill of a row block failed"));
            dst[g] = BufRef{local[g], plan.groups[g].local_pitch};
        }
        S_TRY(run_jobs(ctx, plan.b_unpack, dst, recv, false, scratch));
    }
    span_end(ctx, "split_exchange_b");
    // the slabs and the exchange buffers are done with (stream-ordered releases)
    for (void* p : scratch) pool_release(ctx, p);
    scratch.clear();
    // ---- this rank's part of the commitment
    c->n_mats = n;
    c->log_blowup = log_blowup;
    c->split_log_g = log_g;
    c->split_rank = rank;
    c->lde.assign((size_t)n, nullptr);
    c->coeffs.assign((size_t)n, nullptr);
    c->log_h.resize((size_t)n);
    c->width.resize((size_t)n);
    c->pitch.resize((size_t)n);
    c->lde_is_view.assign((size_t)n, 1);
    c->group.assign((size_t)n, -1);
    c->col_start.assign((size_t)n, 0);
    c->full_lde.assign((size_t)n, nullptr);
    c->next_off.assign((size_t)n, 0);
    c->tiny_rows_m.assign((size_t)n, {});
    c->log_max = 0;
    for (int i = 0; i < n; i++) {
        c->log_h[(size_t)i] = (int)mats[i].log_n + log_blowup;
        c->width[(size_t)i] = mats[i].width;
        c->log_max = std::max(c->log_max, c->log_h[(size_t)i]);
    }
    if (c->log_max <= log_g + 1) return done(set_error(ctx, LURKHIP_ERR_INVALID_ARG, "split: the tallest matrix has 2^%d rows -- too short to cut over %d ranks", c->log_max - log_blowup, G));
    for (size_t g = 0; g < plan.groups.size(); g++) {
        const split::Group& gr = plan.groups[g];
        c->group_base.push_back(local[g]);
        for (size_t k = 0; k < gr.mats.size(); k++) {
            const int i = gr.mats[k];
            c->lde[(size_t)i] = local[g] + gr.col_start[k];
            c->pitch[(size_t)i] = gr.local_pitch;
            c->group[(size_t)i] = (int)g;
            c->col_start[(size_t)i] = gr.col_start[k];
        }
        for (size_t e = 0; e < gr.extras.size(); e++)
            if (gr.extras[e].col == 0) c->next_off[(size_t)gr.extras[e].mat] = gr.W_local + (uint32_t)e - c->col_start[(size_t)gr.extras[e].mat];
    }
    std::map<int, int> small_group;  // group of c->aux -> group of c
    for (size_t ks = 0; ks < small.size(); ks++) {
        const int i = small[ks];
        const size_t k = nt + ks;  // (the short matrices follow the tiles in c->aux)
        const lurkhip_commitment* a = c->aux;
        c->full_lde[(size_t)i] = a->lde[k];
        c->pitch[(size_t)i] = a->pitch[k];
        c->lde[(size_t)i] = a->lde[k] + c->row_base(i) * a->pitch[k];
        if (a->group[k] >= 0) {
            auto it = small_group.find(a->group[k]);
            if (it == small_group.end()) {
                it = small_group.emplace(a->group[k], (int)c->group_base.size()).first;
                c->group_base.push_back(a->group_base[(size_t)a->group[k]] + c->row_base(i) * a->pitch[k]);
            }
            c->group[(size_t)i] = it->second;
            c->col_start[(size_t)i] = a->col_start[k];
        }
        if (!c->is_local(i)) {  // at most G rows: read back for the top of the tree and for the query answers
            const size_t rows = (size_t)1 << c->log_h[(size_t)i];
            c->tiny_rows_m[(size_t)i].resize(rows * mats[i].width);
            if (hipMemcpy2DAsync(c->tiny_rows_m[(size_t)i].data(), (size_t)mats[i].width * 4, a->lde[k], (size_t)a->pitch[k] * 4, (size_t)mats[i].width * 4, rows,
                                 hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
                return done(set_error(ctx, LURKHIP_ERR_HIP, "split: read-back of a short matrix failed"));
        }
    }
    // ---- 5. the subtree over this rank's rows
    {
        lurkhip_commitment t;
        t.owns_lde = false;
        for (int i = 0; i < n; i++) {
            if (!c->is_local(i)) continue;
            t.lde.push_back(c->lde[(size_t)i]);
            t.log_h.push_back(c->log_h[(size_t)i] - log_g);
            t.width.push_back(c->width[(size_t)i]);
            t.pitch.push_back(c->pitch[(size_t)i]);
        }
        t.n_mats = (int)t.lde.size();
        t.coeffs.assign(t.lde.size(), nullptr);
        S_TRY(commitment_build_tree(ctx, &t));
        c->digests = t.digests;
        c->level_off = t.level_off;
        t.digests = nullptr;
    }
    // ---- 6. the subtree roots of all ranks, and the top of the tree on the host
    const int log_local = c->log_max - log_g;
    uint32_t my_root[8];
    {
        void* pin = nullptr;
        S_TRY(pinned_small(ctx, &pin));
        if (hipMemcpyAsync(pin, c->digests + c->level_off[(size_t)log_local] * 8, 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx) != hipSuccess)
            return done(set_error(ctx, LURKHIP_ERR_HIP, "split: read-back of the subtree root failed"));
        memcpy(my_root, pin, 32);
    }
    std::vector<uint32_t> roots((size_t)G * 8);
    S_TRY(split_allgather_host(ctx, env, my_root, roots.data(), 32));
    auto inject = [&](int log_rows, size_t row, uint32_t node[8]) {  // node <- compress(node, sponge(row `row` of the matrices of 2^log_rows rows)), if any
        std::vector<std::pair<const uint32_t*, uint32_t>> rows;
        for (int i = 0; i < n; i++)
            if (c->log_h[(size_t)i] == log_rows && !c->tiny_rows_m[(size_t)i].empty()) rows.push_back({c->tiny_rows_m[(size_t)i].data() + row * c->width[(size_t)i], c->width[(size_t)i]});
        if (rows.empty()) return;
        uint32_t h[8], d[8];
        H.sponge(rows, h);
        H.compress(node, h, d);
        memcpy(node, d, 32);
    };
    c->top_levels_m.assign((size_t)log_g + 1, {});
    c->top_levels_m[0] = roots;
    for (int r = 0; r < G; r++) inject(log_g, (size_t)r, &c->top_levels_m[0][(size_t)r * 8]);
    for (int t = 1; t <= log_g; t++) {
        const int nodes = G >> t;
        c->top_levels_m[(size_t)t].resize((size_t)nodes * 8);
        for (int j = 0; j < nodes; j++) {
            uint32_t* node = &c->top_levels_m[(size_t)t][(size_t)j * 8];
            H.compress(&c->top_levels_m[(size_t)t - 1][(size_t)(2 * j) * 8], &c->top_levels_m[(size_t)t - 1][(size_t)(2 * j + 1) * 8], node);
            inject(log_g - t, (size_t)j, node);
        }
    }
    if (root_m) memcpy(root_m, c->top_levels_m[(size_t)log_g].data(), 32);
#undef S_TRY
    *out = c;
    c = nullptr;
    return done(LURKHIP_OK);
}

}  // namespace lurkhip

using namespace lurkhip;

extern "C" int32_t lurkhip_split_stats(lurkhip_ctx* ctx, uint64_t* out, int32_t reset) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out != nullptr, "null argument");
    out[0] = ctx->split_words_a * 4;
    out[1] = ctx->split_words_b * 4;
    out[2] = ctx->split_exchanges;
    if (reset) ctx->split_words_a = ctx->split_words_b = ctx->split_exchanges = 0;
    return LURKHIP_OK;
}

extern "C" int64_t lurkhip_split_plan(int32_t world, int32_t rank, int32_t split_min_log_n, int32_t n_mats, const uint32_t* log_heights,
                                      const uint32_t* widths, const int32_t* kinds, const uint32_t* lqds, const uint32_t* chunks, const uint32_t* n_next,
                                      const uint32_t* run_counts, const uint32_t* runs, uint64_t* out, uint64_t capacity) {
    if (n_mats <= 0 || !log_heights || !widths || !kinds) return LURKHIP_ERR_INVALID_ARG;
    int log_g = 0;
    while ((1 << log_g) < world) log_g++;
    if (world < 2 || (1 << log_g) != world) return LURKHIP_ERR_INVALID_ARG;
    std::vector<split::MatDesc> descs((size_t)n_mats);
    for (int i = 0; i < n_mats; i++)
        descs[(size_t)i] = split::MatDesc{log_heights[i], widths[i], kinds[i], lqds ? lqds[i] : 0u, chunks ? chunks[i] : 0u, n_next ? n_next[i] : 0u, 1u, {}};
    if (run_counts && runs)
        for (int i = 0, at = 0; i < n_mats; i++)
            for (uint32_t k = 0; k < run_counts[i]; k++, at++) descs[(size_t)i].runs.push_back({runs[2 * at], runs[2 * at + 1]});
    split::Plan p;
    try {
        p = split::make_plan(log_g, rank, split_min_log_n, descs);
    } catch (const std::exception& e) {
        set_error(nullptr, LURKHIP_ERR_INVALID_ARG, "%s", e.what());
        return LURKHIP_ERR_INVALID_ARG;
    }
    std::vector<uint64_t> o;
    auto put_jobs = [&](const std::vector<split::Job>& jobs) {
        o.push_back(jobs.size());
        for (const split::Job& j : jobs) o.insert(o.end(), {(uint64_t)j.buf, j.row0, j.col0, j.row_stride, j.lin_off, j.lin_pitch, j.width, j.rows});
    };
    auto put_vec = [&](const std::vector<uint64_t>& v) {
        o.push_back(v.size());
        o.insert(o.end(), v.begin(), v.end());
    };
    o.push_back((uint64_t)world);
    o.push_back(p.groups.size());
    for (size_t g = 0; g < p.groups.size(); g++) {
        const split::Group& gr = p.groups[g];
        o.insert(o.end(), {gr.log_n, gr.W, gr.local_pitch, (uint64_t)p.slab_w[g], gr.mats.size(), gr.extras.size(), gr.W_local, gr.sparse ? 1u : 0u});
        for (size_t k = 0; k < gr.mats.size(); k++) o.insert(o.end(), {(uint64_t)gr.mats[k], gr.col_start[k]});
        for (uint32_t b : gr.bounds) o.push_back(b);
        for (const split::Extra& e : gr.extras) o.insert(o.end(), {(uint64_t)e.mat, e.col, e.vcol, (uint64_t)e.owner});
    }
    o.push_back(p.tiles.size());
    for (const split::Tile& t : p.tiles) o.insert(o.end(), {(uint64_t)t.group, (uint64_t)t.mat, t.c0, t.w, t.slab_col});
    o.push_back(p.my_extras.size());
    for (int e : p.my_extras) o.push_back((uint64_t)e);
    o.push_back(p.has_a ? 1 : 0);
    put_vec(p.a_send_off);
    put_vec(p.a_recv_off);
    put_jobs(p.a_pack);
    put_jobs(p.a_unpack);
    put_vec(p.b_send_off);
    put_vec(p.b_recv_off);
    put_jobs(p.b_pack);
    put_jobs(p.b_unpack);
    if (out && capacity >= o.size()) memcpy(out, o.data(), o.size() * 8);
    return (int64_t)o.size();
}
