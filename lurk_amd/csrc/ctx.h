// Internal context object behind the lurkhip C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lurkhip.h"

struct lurkhip_ctx {
    int device = 0;
    int num_cus = 256;  // compute units of the device (persistent-kernel grids)
    hipStream_t stream = nullptr;
    // the context's own stream, set once at creation and never swapped: `stream` is rerouted to a side / hash stream for the
    // duration of a scope by calls that hold api_mu (SideLane::Guard, early_sponge), so the entry points that run WITHOUT api_mu
    // (LH_CHECK_CTX_NOLOCK: the *_free family) must not read it
    hipStream_t main_stream = nullptr;
    bool owns_stream = false;
    int stream_priority = 0;  // of the context's own streams (lurkhip_ctx_create_with_priority)
    std::string err;      // written under err_mu (set_error), read by lurkhip_last_error into the calling thread's copy
    std::mutex err_mu;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // grow-only scratch arenas for the host-pointer entry points
    void* arena[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t arena_bytes[4] = {0, 0, 0, 0};
    // grow-only page-locked host buffer for the prover's read-backs (host_staging): copies into it are stream-ordered and
    // do not bounce through the runtime's own staging
    void* host_stage = nullptr;
    size_t host_stage_bytes = 0;
    void* pin_small = nullptr;  // 256 page-locked bytes for the few-word read-backs (roots, PoW witness): pinned_small()
    // lazily created per-ctx device state owned by other translation units (commit.h)
    void* merkle_params_dev = nullptr;
    void* merkle_params_host = nullptr;  // P16Params copy for the host-side challenger
    lurkhip_protocol_profile* profile = nullptr;  // every recalled protocol choice (lurkhip.h); default preset until set
    void* ntt_plans[32] = {};
    // coset-shift power tables of the LDE, s^i / N for i < N, keyed by (log_n, s): immutable once filled, so a table is
    // computed once per context instead of once per matrix (commit.hip: extend)
    std::map<std::pair<int, uint32_t>, uint32_t*> lde_scale_tables;
    // quotient-domain selector tables (stark.hip: selector_table) keyed by (log_n, log quotient degree): is_first_row,
    // is_last_row, is_transition at x = g w_Q^bitrev(s), three words per row -- functions of the domain only, immutable
    std::map<std::pair<int, int>, uint32_t*> selector_tables;
    size_t lde_scale_bytes = 0;
    std::vector<std::function<void()>> cleanups;  // run in reverse order by lurkhip_ctx_destroy
    // size-keyed free lists so steady-state proving does no hipMalloc/hipFree (pool_alloc/pool_release)
    std::multimap<size_t, void*> pool_free;
    std::map<void*, size_t> pool_live;
    // pool accounting (lurkhip_pool_stats): bytes handed out, bytes cached on the free list, high-water mark of their sum,
    // hipMalloc calls, OOM retries; inject_alloc_failures is the test hook that forces the retry path
    uint64_t pool_live_bytes = 0, pool_cached_bytes = 0, pool_peak_bytes = 0, pool_mallocs = 0, pool_retries = 0;
    int inject_alloc_failures = 0;
    // Fork / join inside one proof (lurkhip::SideLane): a second stream of the context for the short chips' launches.  While a
    // lane is open every pool_release is deferred to the join, so no block is handed out again while either stream may still use it.
    hipStream_t hash_stream = nullptr;  // commit.hip: early_leaves (a stream measured to run beside the context's own)
    hipEvent_t hash_ready = nullptr, hash_done = nullptr;
    static constexpr int N_SIDE = 4;
    hipStream_t side_stream[N_SIDE] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t side_fork = nullptr, side_join[N_SIDE] = {nullptr, nullptr, nullptr, nullptr};
    bool side_used[N_SIDE] = {false, false, false, false};
    bool defer_releases = false;
    std::vector<void*> deferred;
    // One API call on a context at a time: every entry point holds this lock for its duration (LH_CHECK_CTX), so that calls from
    // several host threads on ONE context serialise instead of racing on its arenas, caches and error string.  Recursive: entry
    // points call entry points.  The entry points that only hand pooled memory back (`*_free`) take pool_mu alone
    // (LH_CHECK_CTX_NOLOCK): a proving thread may release a shard's inputs while the staging thread is inside an upload call.
    std::recursive_mutex api_mu;
    std::mutex pool_mu;  // a streaming prover releases one shard's inputs on its proving thread while the next shard's are allocated on its staging thread
    // page-locked staging of the row-stream uploads (lair_api.cpp: lurkhip_func_trace_prepare_many): grow-only, a buffer is reused by a
    // later call once its `prep_done` (recorded behind the last upload that read it) has passed
    void* prep_stage[2] = {nullptr, nullptr};  // two buffers in turn: one is being filled while the other is still being read by its copies
    size_t prep_stage_bytes[2] = {0, 0};
    hipEvent_t prep_done[2] = {nullptr, nullptr};
    int prep_turn = 0;
    // optional per-span HIP-event timing (lurkhip_profile_*)
    bool profiling = false;
    int profile_level = 0;  // 1: the stage spans (a dozen event records per commitment), 2: also per-chip / per-small-tree spans
    struct Span {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
        double total_ms = 0;
        long count = 0;
    };
    std::map<std::string, Span> spans;
    std::vector<hipEvent_t> event_pool;
    // one shard over several ranks (split.hip): words this rank sent to OTHER ranks in the all-to-alls before / after the LDEs, and
    // the number of all-to-alls (lurkhip_split_stats)
    uint64_t split_words_a = 0, split_words_b = 0, split_exchanges = 0;
    // the last shard proof's permutation traces: cells in all, and in the columns whose LDE was computed (the identically-zero ones
    // are left out: lurkhip_prover_stats -- the bench's "algorithmic bytes transformed")
    uint64_t perm_cells = 0, perm_cells_transformed = 0;
};

namespace lurkhip {

int32_t set_error(lurkhip_ctx* ctx, int32_t code, const char* fmt, ...);
// returns a device buffer of at least `bytes` from scratch slot `slot` (grown on demand)
int32_t arena_get(lurkhip_ctx* ctx, int slot, size_t bytes, void** out);
// waits for the stream like hipStreamSynchronize, polling first: the prover's read-backs come back within tens of
// microseconds and a blocking wait adds its wake-up latency to every transcript round trip
hipError_t stream_wait(lurkhip_ctx* ctx);
// page-locked host buffer of at least `bytes` (grown on demand; the previous contents are dropped, the stream is drained first)
int32_t host_staging(lurkhip_ctx* ctx, size_t bytes, void** out);
// 256 page-locked bytes of the context for read-backs of a few words that are waited for at once (a copy into pageable memory
// goes through the runtime's own staging and its blocking wait; into page-locked memory it is a plain DMA the caller polls for)
int32_t pinned_small(lurkhip_ctx* ctx, void** out);
// a few words from the host to the device as launch arguments of a tiny kernel on the context's stream: no staging buffer, no
// copy packet, nothing for the host to keep alive or wait for
int32_t upload_words(lurkhip_ctx* ctx, uint32_t* dst_dev, const uint32_t* src, size_t n);
// pooled device allocations: released blocks are kept and reused for later requests of the same size
int32_t pool_alloc(lurkhip_ctx* ctx, size_t bytes, void** out);
void pool_release(lurkhip_ctx* ctx, void* ptr);
// Fork / join of the side lanes: between open() and close() work enqueued while an `on_side(true, key)` guard is alive goes to
// side stream `key mod N_SIDE` of the context, ordered after everything queued before open(); close() makes the main stream
// wait for every side stream used.  The short chips of a machine (a few workgroups per launch, latency-bound) run there under
// the tall chips' kernels -- and, since round 3, under one another: work that shares an accumulator must share a key.
// LURKHIP_SIDE_LANE=0 disables it (everything on the main stream); LURKHIP_SIDE_LANES=1 keeps one side stream (round 2).
struct SideLane {
    lurkhip_ctx* ctx;
    bool active = false;
    explicit SideLane(lurkhip_ctx* c) : ctx(c) {}
    int32_t open();
    int32_t close();
    ~SideLane() { (void)close(); }
    struct Guard {  // routes the launches of its scope to the side stream
        lurkhip_ctx* ctx;
        hipStream_t saved;
        Guard(lurkhip_ctx* c, bool on, uint32_t lane) : ctx(c), saved(c->stream) {
            if (on) {
                c->stream = c->side_stream[lane];
                c->side_used[lane] = true;
            }
        }
        ~Guard() { ctx->stream = saved; }
    };
    int lanes = 1;
    // side streams to use (set before open()): 1 when tall chips own the device anyway -- four lanes of short kernels under them
    // measured 0.5 % slower on the 2^20-row step --, N_SIDE for a proof made of short chips only (2^12 rows: 9.8 -> 9.1 ms)
    int want = lurkhip_ctx::N_SIDE;
    Guard on_side(bool on, uint32_t key = 0) { return Guard(ctx, on && active, key % (uint32_t)lanes); }
};
// the context's hash stream (created at first use, measured to run beside the context's own stream) with its two events
int32_t hash_stream_of(lurkhip_ctx* ctx);
// span timing (no-ops unless profiling is enabled)
// `level`: 1 = stage span (recorded whenever profiling is on), 2 = detail span (per chip, per small tree: only at profile level 2 --
// every event record is a marker packet the next kernel waits behind, a few hundred of them cost a millisecond per proof)
void span_begin(lurkhip_ctx* ctx, const char* name, int level = 1);
void span_end(lurkhip_ctx* ctx, const char* name, int level = 1);
void span_switch(lurkhip_ctx* ctx, const char* from, const char* to, int level = 1);  // span_end(from) + span_begin(to) on one event
void host_mark(const char* what);  // LURKHIP_HOST_TRACE=1: the host clock at this point, on stderr (development aid)

}  // namespace lurkhip

// Every entry point starts here.  The context's lock is held until the entry point returns (see lurkhip_ctx::api_mu), and the
// calling thread's current HIP device is the context's for that time: a new host thread starts on device 0 -- the lanes' worker
// threads of a rank that owns GPU 3 would otherwise launch on streams of another device -- and is put back afterwards (ADVICE
// round 3: a caller's later torch.empty(device="cuda") must not land on the context's device because it proved something).
namespace lurkhip {
struct CtxEntry {
    lurkhip_ctx* ctx;
    int prev = -1;
    bool ok = true;
    explicit CtxEntry(lurkhip_ctx* c, bool lock) : ctx(c), locked(lock) {
        if (locked) c->api_mu.lock();
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device && hipSetDevice(c->device) != hipSuccess) ok = false;
    }
    ~CtxEntry() {
        if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
        if (locked) ctx->api_mu.unlock();
    }
    CtxEntry(const CtxEntry&) = delete;
    CtxEntry& operator=(const CtxEntry&) = delete;

   private:
    bool locked;
};
}  // namespace lurkhip
#define LH_ENTRY_CAT2(a, b) a##b
#define LH_ENTRY_CAT(a, b) LH_ENTRY_CAT2(a, b)
#define LH_CHECK_CTX_IMPL(ctx, lock)                                                                              \
    if (!(ctx)) return lurkhip::set_error(nullptr, LURKHIP_ERR_INVALID_ARG, "null ctx");                          \
    lurkhip::CtxEntry LH_ENTRY_CAT(lh_entry_, __LINE__)((ctx), (lock));                                            \
    if (!LH_ENTRY_CAT(lh_entry_, __LINE__).ok)                                                                     \
    return lurkhip::set_error((ctx), LURKHIP_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device)
#define LH_CHECK_CTX(ctx) LH_CHECK_CTX_IMPL(ctx, true)
#define LH_CHECK_CTX_NOLOCK(ctx) LH_CHECK_CTX_IMPL(ctx, false)

#define LH_HIP(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return lurkhip::set_error((ctx), e__ == hipErrorOutOfMemory ? LURKHIP_ERR_OOM : LURKHIP_ERR_HIP, \
                                      "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define LH_TRY(expr)                    \
    do {                                \
        int32_t s__ = (expr);           \
        if (s__ != LURKHIP_OK) return s__; \
    } while (0)

#define LH_ARG(ctx, cond, ...)                                                          \
    do {                                                                                \
        if (!(cond)) return lurkhip::set_error((ctx), LURKHIP_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)
