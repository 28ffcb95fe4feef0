// The two collectives of a multi-GPU proof behind the C ABI (round 4): RCCL over xGMI, enqueued on the context's stream.
//
// Replaces what the reference does inside ONE process: `Shard::shard` cuts an execution into shards
// (/root/reference/src/lair/execute.rs:186-241); sphinx's prover commits every shard's main trace, observes EVERY commitment
// into the shared challenger before any per-shard challenge is drawn, proves the shards, and the verifier checks that the
// chips' cumulative sums of all shards add up to zero (/root/reference/src/lair/lair_chip.rs:104-139 decides which chips a
// shard holds).  With one process per GPU the only data that cross ranks are (shard index, 8-word root) records --
// an all-gather -- and the extension-field sums -- an all-reduce; RCCL has no modular reduction, so sums travel as 4 x int64
// of canonical addends and are reduced mod p locally.
//
// librccl is loaded at run time (dlopen): the library has no link-time dependency on it, and a single-GPU user never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "babybear.h"
#include "ctx.h"

struct lurkhip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    lurkhip_ctx* split_ctx = nullptr;  // the context lurkhip_comm_split_vtable was asked for (pool and page-locked staging of the host collectives)
};

namespace lurkhip {
namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    // one shard over several ranks (lurkhip_comm_split_vtable): the all-to-all is grouped point-to-point
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    std::string path;  // what was loaded (lurkhip_comm_library)
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // 1. LURKHIP_RCCL_LIB, when set, is THE library (nothing else is tried: a host that names its RCCL means it).
        // 2. A librccl the process has already mapped -- PyTorch maps its own bundled torch/lib/librccl.so and owns communicators on
        //    the same devices -- is re-used (RTLD_NOLOAD on the mapped path): two RCCL copies in one process would each keep their
        //    own device state.  3. Otherwise the system library by name.
        const char* forced = getenv("LURKHIP_RCCL_LIB");
        if (forced && *forced) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
            r.path = forced;
        } else {
            if (FILE* maps = fopen("/proc/self/maps", "r")) {
                char ln[4352];
                while (!r.lib && fgets(ln, sizeof ln, maps)) {
                    const char* path = strchr(ln, '/');
                    if (!path || !strstr(path, "/librccl.so")) continue;
                    std::string p(path);
                    while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
                    r.lib = dlopen(p.c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
                    if (r.lib) r.path = p + " (already mapped by the process)";
                }
                fclose(maps);
            }
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names) {
                if (r.lib) break;
                r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (r.lib) r.path = n;
            }
        }
        if (!r.lib) {
            const char* e = dlerror();  // one call: dlerror() clears the message it returns
            r.why = std::string("librccl could not be loaded") + (forced && *forced ? std::string(" from LURKHIP_RCCL_LIB=") + forced : std::string()) + ": " +
                    (e ? e : "not found");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p && r.why.empty()) r.why = std::string("librccl has no symbol ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

#define LH_RCCL(ctx, expr)                                                                                          \
    do {                                                                                                            \
        ncclResult_t r__ = (expr);                                                                                  \
        if (r__ != ncclSuccess)                                                                                     \
            return set_error((ctx), LURKHIP_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r__) : "?"); \
    } while (0)

// total[c] = lanes[c] mod p (the all-reduced sums are non-negative: canonical addends)
__global__ void k_lanes_mod_p(const long long* __restrict__ lanes, uint32_t* __restrict__ total) {
    if (threadIdx.x < 4) total[threadIdx.x] = (uint32_t)((unsigned long long)lanes[threadIdx.x] % bb::P);
}
__global__ void k_set_lanes(long long* __restrict__ lanes, long long a, long long b, long long c, long long d) {
    if (threadIdx.x == 0) lanes[0] = a, lanes[1] = b, lanes[2] = c, lanes[3] = d;
}

}  // namespace
}  // namespace lurkhip

using namespace lurkhip;

extern "C" {

const char* lurkhip_comm_library(void) {
    const Rccl& r = rccl();
    return r.why.empty() ? r.path.c_str() : nullptr;
}

int32_t lurkhip_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return set_error(nullptr, LURKHIP_ERR_INVALID_ARG, "null id_out");
    const Rccl& r = rccl();
    if (!r.why.empty()) return set_error(nullptr, LURKHIP_ERR_HIP, "%s", r.why.c_str());
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return LURKHIP_ERR_HIP;
    static_assert(sizeof id == LURKHIP_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return LURKHIP_OK;
}

int32_t lurkhip_comm_create(lurkhip_ctx* ctx, const uint8_t* id, int32_t rank, int32_t world, lurkhip_comm** out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, id && out && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments (rank %d of %d)", rank, world);
    const Rccl& r = rccl();
    if (!r.why.empty()) return set_error(ctx, LURKHIP_ERR_HIP, "%s", r.why.c_str());
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    lurkhip_comm* c = new lurkhip_comm();
    c->rank = rank;
    c->world = world;
    ncclResult_t st = r.CommInitRank(&c->comm, world, uid, rank);  // one rank per device: the context's device is current
    if (st != ncclSuccess) {
        delete c;
        return set_error(ctx, LURKHIP_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r.GetErrorString(st));
    }
    *out = c;
    return LURKHIP_OK;
}

int32_t lurkhip_comm_destroy(lurkhip_ctx* ctx, lurkhip_comm* comm) {
    LH_CHECK_CTX(ctx);
    if (!comm) return LURKHIP_OK;
    (void)stream_wait(ctx);
    if (comm->comm) (void)rccl().CommDestroy(comm->comm);
    delete comm;
    return LURKHIP_OK;
}

int32_t lurkhip_comm_info(const lurkhip_comm* comm, int32_t* rank, int32_t* world) {
    if (!comm) return LURKHIP_ERR_INVALID_ARG;
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return LURKHIP_OK;
}

int32_t lurkhip_exchange_roots_dev(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* records_dev, int32_t n_local, uint32_t* gathered_dev) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm && records_dev && gathered_dev && n_local >= 1, "bad exchange arguments");
    LH_RCCL(ctx, rccl().AllGather(records_dev, gathered_dev, (size_t)n_local * LURKHIP_ROOT_RECORD_WORDS, ncclUint32, comm->comm, ctx->stream));
    return LURKHIP_OK;
}

// The exchange for any number of shards per rank.  Two collectives, both of a size every rank knows without asking: the counts
// (one word per rank), then records padded to the largest count.  A rank that cannot take part properly (bad arguments, an
// allocation that failed) still enters both collectives -- with count -1 -- so that its peers return an error instead of waiting
// for it forever (an early return before a collective leaves the other ranks blocked inside it).
static int32_t exchange_var(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* shard_indices, const uint32_t* roots, int32_t n_local, int64_t n_total,
                            bool equal_counts, uint32_t* roots_out) {
    const int world = comm->world;
    // (a negative n_total from the _var entry point is a bad argument like the others: this rank still enters both collectives, with -1)
    const bool args_ok = n_local >= 0 && n_local <= 4096 && roots_out && (n_local == 0 || (shard_indices && roots)) && n_total >= -1;
    void* cnt = nullptr;  // world + 1 words: [0] mine, [1 ..] everybody's
    LH_TRY(pool_alloc(ctx, ((size_t)world + 1) * 4, &cnt));
    void *send = nullptr, *recv = nullptr;
    auto done = [&](int32_t s) {
        pool_release(ctx, cnt);
        if (send) pool_release(ctx, send);
        if (recv) pool_release(ctx, recv);
        return s;
    };
    const uint32_t mine = args_ok ? (uint32_t)n_local : 0xFFFFFFFFu;
    std::vector<uint32_t> counts((size_t)world);
    if (upload_words(ctx, (uint32_t*)cnt, &mine, 1) != LURKHIP_OK) return done(set_error(ctx, LURKHIP_ERR_HIP, "upload of the shard count failed"));
    ncclResult_t nr = rccl().AllGather(cnt, (uint32_t*)cnt + 1, 1, ncclUint32, comm->comm, ctx->stream);
    if (nr != ncclSuccess) return done(set_error(ctx, LURKHIP_ERR_HIP, "ncclAllGather of the shard counts failed: %s", rccl().GetErrorString(nr)));
    if (hipMemcpyAsync(counts.data(), (uint32_t*)cnt + 1, (size_t)world * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx) != hipSuccess)
        return done(set_error(ctx, LURKHIP_ERR_HIP, "read-back of the gathered shard counts failed"));
    uint32_t most = 0;
    int64_t sum = 0;
    for (int r = 0; r < world; r++) {
        if (counts[(size_t)r] == 0xFFFFFFFFu) return done(set_error(ctx, LURKHIP_ERR_INVALID_ARG, "rank %d entered the root exchange with unusable arguments", r));
        if (equal_counts && counts[(size_t)r] != counts[0])
            return done(set_error(ctx, LURKHIP_ERR_INVALID_ARG, "rank %d passes %u shards, rank 0 passes %u: lurkhip_exchange_roots needs equal counts (use lurkhip_exchange_roots_var)",
                                  r, counts[(size_t)r], counts[0]));
        most = std::max(most, counts[(size_t)r]);
        sum += counts[(size_t)r];
    }
    if (n_total < 0) n_total = sum;
    if (sum != n_total) return done(set_error(ctx, LURKHIP_ERR_INVALID_ARG, "the ranks hold %lld shards in all, the execution has %lld", (long long)sum, (long long)n_total));
    if (n_total == 0) return done(LURKHIP_OK);
    const size_t rec_words = (size_t)most * LURKHIP_ROOT_RECORD_WORDS, all_words = rec_words * (size_t)world;
    std::vector<uint32_t> rec(rec_words, 0xFFFFFFFFu);  // index 0xFFFFFFFF = padding
    for (int i = 0; i < n_local; i++) {
        rec[(size_t)i * LURKHIP_ROOT_RECORD_WORDS] = shard_indices[i];
        memcpy(&rec[(size_t)i * LURKHIP_ROOT_RECORD_WORDS + 1], roots + (size_t)i * 8, 32);
    }
    // (an allocation failure here cannot be reported to the peers any more: they are about to enter the second all-gather.  The two
    // buffers are tens of bytes per shard from the context's pool.)
    int32_t st = pool_alloc(ctx, rec_words * 4, &send);
    if (st == LURKHIP_OK) st = pool_alloc(ctx, all_words * 4, &recv);
    if (st != LURKHIP_OK) return done(st);
    std::vector<uint32_t> all(all_words);
    if (hipMemcpyAsync(send, rec.data(), rec_words * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return done(set_error(ctx, LURKHIP_ERR_HIP, "upload of the root records failed"));
    nr = rccl().AllGather(send, recv, rec_words, ncclUint32, comm->comm, ctx->stream);
    if (nr != ncclSuccess) return done(set_error(ctx, LURKHIP_ERR_HIP, "ncclAllGather of the root records failed: %s", rccl().GetErrorString(nr)));
    if (hipMemcpyAsync(all.data(), recv, all_words * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx) != hipSuccess)
        return done(set_error(ctx, LURKHIP_ERR_HIP, "read-back of the gathered roots failed"));
    // shard order; the indices of all ranks must be a partition of 0 .. n_total - 1
    std::vector<char> seen((size_t)n_total, 0);
    for (int r = 0; r < world; r++)
        for (uint32_t k = 0; k < counts[(size_t)r]; k++) {
            const uint32_t* w = &all[((size_t)r * most + k) * LURKHIP_ROOT_RECORD_WORDS];
            if (w[0] >= (uint64_t)n_total || seen[w[0]])
                return done(set_error(ctx, LURKHIP_ERR_INVALID_ARG, "shard indices of the ranks are not a partition of 0 .. %lld", (long long)n_total - 1));
            seen[w[0]] = 1;
            memcpy(roots_out + (size_t)w[0] * 8, w + 1, 32);
        }
    return done(LURKHIP_OK);
}

int32_t lurkhip_exchange_roots(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* shard_indices, const uint32_t* roots, int32_t n_local,
                               uint32_t* roots_out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm != nullptr, "null communicator");
    return exchange_var(ctx, comm, shard_indices, roots, n_local >= 1 ? n_local : -1, -1, true, roots_out);
}

int32_t lurkhip_exchange_roots_var(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* shard_indices, const uint32_t* roots, int32_t n_local,
                                   int32_t n_total, uint32_t* roots_out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm != nullptr, "null communicator");
    return exchange_var(ctx, comm, shard_indices, roots, n_local, n_total < 0 ? -2 : n_total, false, roots_out);
}

int32_t lurkhip_reduce_sums_dev(lurkhip_ctx* ctx, lurkhip_comm* comm, int64_t* lanes_dev, uint32_t* total_dev) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm && lanes_dev && total_dev, "bad reduce arguments");
    LH_RCCL(ctx, rccl().AllReduce(lanes_dev, lanes_dev, 4, ncclInt64, ncclSum, comm->comm, ctx->stream));
    hipLaunchKernelGGL(k_lanes_mod_p, dim3(1), dim3(64), 0, ctx->stream, (const long long*)lanes_dev, total_dev);
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

// ---- the collectives of one shard over several ranks (include/lurkhip.h: lurkhip_split_comm) on this communicator
namespace {
int32_t sv_fail(lurkhip_comm* c, const char* what, ncclResult_t r) {
    return set_error(c->split_ctx, LURKHIP_ERR_HIP, "%s failed on rank %d: %s", what, c->rank, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
}
int32_t sv_alltoallv_dev(void* user, const uint32_t* send, const uint64_t* soff, uint32_t* recv, const uint64_t* roff, void* stream) {
    lurkhip_comm* c = (lurkhip_comm*)user;
    const Rccl& r = rccl();
    hipStream_t st = (hipStream_t)stream;
    // this rank's own block stays on the device; the others as one group of sends and receives (xGMI is point to point: seven
    // links per GPU, every pair its own)
    const uint64_t own = soff[c->rank + 1] - soff[c->rank];
    if (own && hipMemcpyAsync(recv + roff[c->rank], send + soff[c->rank], own * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return set_error(c->split_ctx, LURKHIP_ERR_HIP, "all-to-all: the local block's copy failed");
    ncclResult_t nr = r.GroupStart();
    if (nr != ncclSuccess) return sv_fail(c, "ncclGroupStart", nr);
    for (int k = 1; k < c->world && nr == ncclSuccess; k++) {
        const int to = (c->rank + k) % c->world, from = (c->rank - k + c->world) % c->world;
        const uint64_t ns = soff[to + 1] - soff[to], nrw = roff[from + 1] - roff[from];
        if (ns) nr = r.Send(send + soff[to], ns, ncclUint32, to, c->comm, st);
        if (nr == ncclSuccess && nrw) nr = r.Recv(recv + roff[from], nrw, ncclUint32, from, c->comm, st);
    }
    const ncclResult_t ne = r.GroupEnd();
    if (nr != ncclSuccess) return sv_fail(c, "ncclSend / ncclRecv", nr);
    if (ne != ncclSuccess) return sv_fail(c, "ncclGroupEnd", ne);
    return 0;
}
int32_t sv_allgather_dev(void* user, const uint32_t* send, uint32_t* recv, uint64_t words, void* stream) {
    lurkhip_comm* c = (lurkhip_comm*)user;
    const ncclResult_t nr = rccl().AllGather(send, recv, words, ncclUint32, c->comm, (hipStream_t)stream);
    return nr == ncclSuccess ? 0 : sv_fail(c, "ncclAllGather", nr);
}
// host buffers: staged through a pooled device block on the context's stream, waited for
int32_t sv_allgather_host(void* user, const void* send, void* recv, uint64_t bytes) {
    lurkhip_comm* c = (lurkhip_comm*)user;
    lurkhip_ctx* ctx = c->split_ctx;
    const uint64_t padded = (bytes + 3) & ~(uint64_t)3;
    void* buf = nullptr;
    LH_TRY(pool_alloc(ctx, padded * ((size_t)c->world + 1), &buf));
    int32_t st = 0;
    if (hipMemcpyAsync(buf, send, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = set_error(ctx, LURKHIP_ERR_HIP, "all-gather: upload failed");
    if (st == 0) {
        const ncclResult_t nr = rccl().AllGather(buf, (uint8_t*)buf + padded, padded / 4, ncclUint32, c->comm, ctx->stream);
        if (nr != ncclSuccess) st = sv_fail(c, "ncclAllGather", nr);
    }
    std::vector<uint8_t> all(padded * (size_t)c->world);
    if (st == 0 && (hipMemcpyAsync(all.data(), (uint8_t*)buf + padded, all.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx) != hipSuccess))
        st = set_error(ctx, LURKHIP_ERR_HIP, "all-gather: read-back failed");
    for (int r = 0; r < c->world && st == 0; r++) memcpy((uint8_t*)recv + (size_t)r * bytes, all.data() + (size_t)r * padded, bytes);
    pool_release(ctx, buf);
    return st;
}
int32_t sv_allreduce_u64_host(void* user, uint64_t* data, uint64_t n) {
    lurkhip_comm* c = (lurkhip_comm*)user;
    lurkhip_ctx* ctx = c->split_ctx;
    if (n == 0) return 0;
    void* buf = nullptr;
    LH_TRY(pool_alloc(ctx, n * 8, &buf));
    int32_t st = 0;
    if (hipMemcpyAsync(buf, data, n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = set_error(ctx, LURKHIP_ERR_HIP, "all-reduce: upload failed");
    if (st == 0) {
        const ncclResult_t nr = rccl().AllReduce(buf, buf, n, ncclUint64, ncclSum, c->comm, ctx->stream);
        if (nr != ncclSuccess) st = sv_fail(c, "ncclAllReduce", nr);
    }
    if (st == 0 && (hipMemcpyAsync(data, buf, n * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx) != hipSuccess))
        st = set_error(ctx, LURKHIP_ERR_HIP, "all-reduce: read-back failed");
    pool_release(ctx, buf);
    return st;
}
}  // namespace

int32_t lurkhip_comm_split_vtable(lurkhip_ctx* ctx, lurkhip_comm* comm, lurkhip_split_comm* out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm && out, "null argument");
    const Rccl& r = rccl();
    if (!r.why.empty()) return set_error(ctx, LURKHIP_ERR_HIP, "%s", r.why.c_str());
    comm->split_ctx = ctx;
    out->rank = comm->rank;
    out->world = comm->world;
    out->user = comm;
    out->alltoallv_dev = sv_alltoallv_dev;
    out->allgather_dev = sv_allgather_dev;
    out->allgather_host = sv_allgather_host;
    out->allreduce_sum_u64_host = sv_allreduce_u64_host;
    return LURKHIP_OK;
}

int32_t lurkhip_reduce_sums(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* local_sums, int32_t n_sums, uint32_t* total) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm && total && n_sums >= 0 && (n_sums == 0 || local_sums), "bad reduce arguments");
    uint64_t acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_sums; i++)
        for (int c = 0; c < 4; c++) {
            LH_ARG(ctx, local_sums[(size_t)i * 4 + c] < bb::P, "a cumulative sum is not canonical");
            acc[c] = (acc[c] + local_sums[(size_t)i * 4 + c]) % bb::P;
        }
    void* buf = nullptr;  // 4 x int64 lanes, then 4 words of total
    LH_TRY(pool_alloc(ctx, 64, &buf));
    auto done = [&](int32_t s) {
        pool_release(ctx, buf);
        return s;
    };
    hipLaunchKernelGGL(k_set_lanes, dim3(1), dim3(64), 0, ctx->stream, (long long*)buf, (long long)acc[0], (long long)acc[1], (long long)acc[2], (long long)acc[3]);
    int32_t st = lurkhip_reduce_sums_dev(ctx, comm, (int64_t*)buf, (uint32_t*)buf + 8);
    if (st != LURKHIP_OK) return done(st);
    if (hipMemcpyAsync(total, (uint32_t*)buf + 8, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx) != hipSuccess)
        return done(set_error(ctx, LURKHIP_ERR_HIP, "read-back of the reduced sums failed"));
    return done(LURKHIP_OK);
}

}  // extern "C"
