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
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {getenv("LURKHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.why = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "not found");
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

int32_t lurkhip_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return LURKHIP_ERR_INVALID_ARG;
    const Rccl& r = rccl();
    if (!r.why.empty()) return LURKHIP_ERR_HIP;
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

int32_t lurkhip_exchange_roots(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* shard_indices, const uint32_t* roots, int32_t n_local,
                               uint32_t* roots_out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm && shard_indices && roots && roots_out && n_local >= 1 && n_local <= 4096, "bad exchange arguments");
    const size_t rec_words = (size_t)n_local * LURKHIP_ROOT_RECORD_WORDS, all_words = rec_words * (size_t)comm->world;
    std::vector<uint32_t> rec(rec_words);
    for (int i = 0; i < n_local; i++) {
        rec[(size_t)i * LURKHIP_ROOT_RECORD_WORDS] = shard_indices[i];
        memcpy(&rec[(size_t)i * LURKHIP_ROOT_RECORD_WORDS + 1], roots + (size_t)i * 8, 32);
    }
    void *send = nullptr, *recv = nullptr;
    LH_TRY(pool_alloc(ctx, rec_words * 4, &send));
    int32_t st = pool_alloc(ctx, all_words * 4, &recv);
    if (st != LURKHIP_OK) {
        pool_release(ctx, send);
        return st;
    }
    std::vector<uint32_t> all(all_words);
    auto done = [&](int32_t s) {
        pool_release(ctx, send);
        pool_release(ctx, recv);
        return s;
    };
    if (hipMemcpyAsync(send, rec.data(), rec_words * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return done(set_error(ctx, LURKHIP_ERR_HIP, "upload of the root records failed"));
    st = lurkhip_exchange_roots_dev(ctx, comm, (const uint32_t*)send, n_local, (uint32_t*)recv);
    if (st != LURKHIP_OK) return done(st);
    if (hipMemcpyAsync(all.data(), recv, all_words * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx) != hipSuccess)
        return done(set_error(ctx, LURKHIP_ERR_HIP, "read-back of the gathered roots failed"));
    // shard order; the indices of all ranks must be a partition of 0 .. n-1
    const size_t n = (size_t)n_local * (size_t)comm->world;
    std::vector<char> seen(n, 0);
    for (size_t k = 0; k < n; k++) {
        const uint32_t idx = all[k * LURKHIP_ROOT_RECORD_WORDS];
        if (idx >= n || seen[idx]) return done(set_error(ctx, LURKHIP_ERR_INVALID_ARG, "shard indices of the ranks are not a partition of 0 .. %zu", n - 1));
        seen[idx] = 1;
        memcpy(roots_out + (size_t)idx * 8, &all[k * LURKHIP_ROOT_RECORD_WORDS + 1], 32);
    }
    return done(LURKHIP_OK);
}

int32_t lurkhip_reduce_sums_dev(lurkhip_ctx* ctx, lurkhip_comm* comm, int64_t* lanes_dev, uint32_t* total_dev) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, comm && lanes_dev && total_dev, "bad reduce arguments");
    LH_RCCL(ctx, rccl().AllReduce(lanes_dev, lanes_dev, 4, ncclInt64, ncclSum, comm->comm, ctx->stream));
    hipLaunchKernelGGL(k_lanes_mod_p, dim3(1), dim3(64), 0, ctx->stream, (const long long*)lanes_dev, total_dev);
    LH_HIP(ctx, hipGetLastError());
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
