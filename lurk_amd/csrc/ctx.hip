// Context, memory helpers and the HIP-event stopwatch of the lurkhip C ABI.
#include <iterator>

#include "ctx.h"

#include <chrono>
#include <set>

#include <cstring>

namespace lurkhip {

static thread_local std::string g_tls_err;

int32_t set_error(lurkhip_ctx* ctx, int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->err_mu);  // an unlocked *_free may fail beside a locked call that fails
        ctx->err = buf;
    }
    g_tls_err = buf;
    return code;
}

int32_t arena_get(lurkhip_ctx* ctx, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (ctx->arena_bytes[slot] < bytes) {
        if (ctx->arena[slot]) {
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            LH_HIP(ctx, hipFree(ctx->arena[slot]));
            ctx->arena[slot] = nullptr;
            ctx->arena_bytes[slot] = 0;
        }
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&ctx->arena[slot], want);
        if (e != hipSuccess) {
            want = bytes;
            LH_HIP(ctx, hipMalloc(&ctx->arena[slot], want));
        }
        ctx->arena_bytes[slot] = want;
    }
    *out = ctx->arena[slot];
    return LURKHIP_OK;
}

hipError_t stream_wait(lurkhip_ctx* ctx) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;  // long waits block
    }
    return hipStreamSynchronize(ctx->stream);
}

int32_t host_staging(lurkhip_ctx* ctx, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (ctx->host_stage_bytes < bytes) {
        if (ctx->host_stage) {
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            LH_HIP(ctx, hipHostFree(ctx->host_stage));
            ctx->host_stage = nullptr;
            ctx->host_stage_bytes = 0;
        }
        const size_t want = bytes + bytes / 4;
        LH_HIP(ctx, hipHostMalloc(&ctx->host_stage, want, hipHostMallocDefault));
        ctx->host_stage_bytes = want;
    }
    *out = ctx->host_stage;
    return LURKHIP_OK;
}

// A few words from the host to the device as launch arguments: no staging buffer, no copy packet, nothing for the host to keep
// alive or wait for (the FRI transcript state, the query indices).
constexpr int UPLOAD_WORDS_MAX = 256;
struct UploadWordsArgs {
    uint32_t w[UPLOAD_WORDS_MAX];
};
__global__ void k_upload_words(UploadWordsArgs a, uint32_t* __restrict__ dst, uint32_t n) {
    if (threadIdx.x < n) dst[threadIdx.x] = a.w[threadIdx.x];
}
int32_t upload_words(lurkhip_ctx* ctx, uint32_t* dst_dev, const uint32_t* src, size_t n) {
    for (size_t at = 0; at < n; at += UPLOAD_WORDS_MAX) {
        UploadWordsArgs a;
        const uint32_t m = (uint32_t)std::min<size_t>(UPLOAD_WORDS_MAX, n - at);
        memcpy(a.w, src + at, (size_t)m * 4);
        hipLaunchKernelGGL(k_upload_words, dim3(1), dim3(UPLOAD_WORDS_MAX), 0, ctx->stream, a, dst_dev + at, m);
    }
    LH_HIP(ctx, hipGetLastError());
    return LURKHIP_OK;
}

int32_t pinned_small(lurkhip_ctx* ctx, void** out) {
    if (!ctx->pin_small) LH_HIP(ctx, hipHostMalloc(&ctx->pin_small, 256, hipHostMallocDefault));
    *out = ctx->pin_small;
    return LURKHIP_OK;
}

int32_t pool_alloc(lurkhip_ctx* ctx, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    // most recently released block of this size first (LIFO): the block most likely still in the caches
    auto range = ctx->pool_free.equal_range(bytes);
    if (range.first != range.second) {
        auto it = std::prev(range.second);
        *out = it->second;
        ctx->pool_free.erase(it);
        ctx->pool_live[*out] = bytes;
        ctx->pool_cached_bytes -= bytes;
        ctx->pool_live_bytes += bytes;
        return LURKHIP_OK;
    }
    auto device_malloc = [&]() -> hipError_t {
        if (ctx->inject_alloc_failures > 0) {  // test hook (lurkhip_debug_inject_alloc_failures): pretend the driver is out of memory
            ctx->inject_alloc_failures--;
            *out = nullptr;
            return hipErrorOutOfMemory;
        }
        ctx->pool_mallocs++;
        return hipMalloc(out, bytes);
    };
    hipError_t e = device_malloc();
    if (e != hipSuccess && !ctx->pool_free.empty()) {
        // Give the cached blocks back to the driver and retry once.  Only blocks on the free list go: nobody holds a pointer to
        // them.  The scale / selector table caches stay -- a caller up the stack may be holding one of their pointers.
        (void)hipStreamSynchronize(ctx->stream);
        for (hipStream_t ss : ctx->side_stream)
            if (ss) (void)hipStreamSynchronize(ss);
        if (ctx->hash_stream) (void)hipStreamSynchronize(ctx->hash_stream);
        for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
        ctx->pool_free.clear();
        ctx->pool_cached_bytes = 0;
        ctx->pool_retries++;
        (void)hipGetLastError();  // the failed hipMalloc must not surface at the next launch check
        e = device_malloc();
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error(ctx, e == hipErrorOutOfMemory ? LURKHIP_ERR_OOM : LURKHIP_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes,
                         hipGetErrorString(e));
    }
    ctx->pool_live_bytes += bytes;
    ctx->pool_peak_bytes = std::max(ctx->pool_peak_bytes, ctx->pool_live_bytes + ctx->pool_cached_bytes);
    ctx->pool_live[*out] = bytes;
    return LURKHIP_OK;
}

// Stream-ordered reuse: a released block is only handed out again to work enqueued later on the
// same stream, so no synchronisation is needed here.
void pool_release(lurkhip_ctx* ctx, void* ptr) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    if (ctx->defer_releases) {  // a side lane is open: the block may be in use on either stream until the join
        ctx->deferred.push_back(ptr);
        return;
    }
    auto it = ctx->pool_live.find(ptr);
    if (it == ctx->pool_live.end()) {
        (void)hipFree(ptr);
        return;
    }
    ctx->pool_free.insert({it->second, ptr});
    ctx->pool_cached_bytes += it->second;
    ctx->pool_live_bytes -= it->second;
    ctx->pool_live.erase(it);
}

namespace {
__global__ void k_occupy(uint64_t ticks) {  // one wave busy for `ticks` of the 100 MHz wall clock
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
}
// seconds for one k_occupy on each of the two streams, launched back to back: ~one kernel's time when the streams sit on
// different hardware queues, two when they share one
double occupy_both(hipStream_t a, hipStream_t b, uint64_t ticks) {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_occupy, dim3(1), dim3(64), 0, a, ticks);
    hipLaunchKernelGGL(k_occupy, dim3(1), dim3(64), 0, b, ticks);
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
// a new stream that a kernel was MEASURED to run on beside a kernel on `ref` (nullptr: creation failed); `alone` = seconds of one
// k_occupy(ticks) by itself on `ref`
hipStream_t stream_beside(hipStream_t ref, int priority, uint64_t ticks, double alone) {
    constexpr int MAX_TRIES = 8;
    std::vector<hipStream_t> parked;
    hipStream_t chosen = nullptr;
    for (int t = 0; t < MAX_TRIES && !chosen; t++) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) break;
        (void)occupy_both(ref, s, ticks / 10);  // first launch on a new stream: not timed
        if (occupy_both(ref, s, ticks) < 1.5 * alone) chosen = s;
        else parked.push_back(s);  // kept until the choice is made, so that the next candidate lands elsewhere
    }
    if (!chosen && !parked.empty()) {  // every candidate shared the queue (a runtime with one queue): take one anyway
        chosen = parked.back();
        parked.pop_back();
    }
    for (hipStream_t s : parked) (void)hipStreamDestroy(s);
    return chosen;
}
double occupy_alone(hipStream_t ref, uint64_t ticks) {
    (void)hipStreamSynchronize(ref);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_occupy, dim3(1), dim3(64), 0, ref, ticks);
    (void)hipStreamSynchronize(ref);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
constexpr uint64_t PLACE_TICKS = 50000;  // 0.5 ms of the 100 MHz wall clock
}  // namespace

int32_t hash_stream_of(lurkhip_ctx* ctx) {
    if (ctx->hash_stream) return LURKHIP_OK;
    ctx->hash_stream = stream_beside(ctx->stream, ctx->stream_priority, PLACE_TICKS, occupy_alone(ctx->stream, PLACE_TICKS));
    if (!ctx->hash_stream) return set_error(ctx, LURKHIP_ERR_HIP, "no hash stream could be created");
    LH_HIP(ctx, hipEventCreateWithFlags(&ctx->hash_ready, hipEventDisableTiming));
    LH_HIP(ctx, hipEventCreateWithFlags(&ctx->hash_done, hipEventDisableTiming));
    return LURKHIP_OK;
}

int32_t SideLane::open() {
    static const bool enabled = getenv("LURKHIP_SIDE_LANE") == nullptr || atoi(getenv("LURKHIP_SIDE_LANE")) != 0;
    if (!enabled || active) return LURKHIP_OK;
    static const int n_lanes = getenv("LURKHIP_SIDE_LANES") ? std::max(1, std::min((int)lurkhip_ctx::N_SIDE, atoi(getenv("LURKHIP_SIDE_LANES")))) : (int)lurkhip_ctx::N_SIDE;
    lanes = std::max(1, std::min(n_lanes, want));
    // Streams are created when a lane is first wanted.  The runtime deals streams to its four hardware queues as they are created
    // (a new stream takes the least used queue), and which of a process's streams end up sharing a queue decides whether two
    // proofs in flight overlap: three idle side streams per context were enough to take the step from 40.6 to 44.5 ms, creating
    // this stream together with the context's own instead of here made the same step 45.3 ms under torchrun (RCCL's streams come
    // in between), and a queue set of their own (stream priority -1) 46 ms.  DESIGN.md section 4 has the table; LURKHIP_PAD_STREAMS
    // and LURKHIP_CTX_PRIORITY are the hooks it was measured with.
    if (!ctx->side_fork) LH_HIP(ctx, hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
    for (int k = 0; k < lanes; k++)
        if (!ctx->side_stream[k]) {
            // (the first one is measured to run beside the context's own stream; once per context, ~2 ms)
            static const bool placed = getenv("LURKHIP_SIDE_UNPLACED") == nullptr;
            if (k == 0 && placed) ctx->side_stream[k] = stream_beside(ctx->stream, ctx->stream_priority, PLACE_TICKS, occupy_alone(ctx->stream, PLACE_TICKS));
            if (!ctx->side_stream[k]) LH_HIP(ctx, hipStreamCreateWithPriority(&ctx->side_stream[k], hipStreamNonBlocking, ctx->stream_priority));
            LH_HIP(ctx, hipEventCreateWithFlags(&ctx->side_join[k], hipEventDisableTiming));
        }
    LH_HIP(ctx, hipEventRecord(ctx->side_fork, ctx->stream));
    for (int k = 0; k < lanes; k++) {
        LH_HIP(ctx, hipStreamWaitEvent(ctx->side_stream[k], ctx->side_fork, 0));
        ctx->side_used[k] = false;
    }
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mu);
        ctx->defer_releases = true;
    }
    active = true;
    return LURKHIP_OK;
}

int32_t SideLane::close() {
    if (!active) return LURKHIP_OK;
    active = false;
    hipError_t e = hipSuccess;
    for (int k = 0; k < lurkhip_ctx::N_SIDE && e == hipSuccess; k++) {
        if (!ctx->side_used[k]) continue;
        e = hipEventRecord(ctx->side_join[k], ctx->side_stream[k]);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->side_join[k], 0);
    }
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mu);
        ctx->defer_releases = false;
        blocks.swap(ctx->deferred);
    }
    if (e != hipSuccess) (void)hipDeviceSynchronize();  // could not order the streams: drain before anything is reused
    for (void* b : blocks) pool_release(ctx, b);  // later work on the main stream runs behind the join
    if (e != hipSuccess) return set_error(ctx, LURKHIP_ERR_HIP, "joining the side lane failed: %s", hipGetErrorString(e));
    return LURKHIP_OK;
}

static hipEvent_t get_event(lurkhip_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// LURKHIP_HOST_TRACE=1: host clock of every span edge on stderr (microseconds since the first one) -- where the host thread is
// when it enqueues a stage, i.e. which stages it waits in and what it does between them; a development aid, off by default
static void host_trace(const char* edge, const char* name) {
    static const bool on = getenv("LURKHIP_HOST_TRACE") != nullptr && atoi(getenv("LURKHIP_HOST_TRACE")) != 0;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "[host %10.1f us] %s %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), edge, name);
}

void host_mark(const char* what) { host_trace("mark ", what); }

void span_begin(lurkhip_ctx* ctx, const char* name, int level) {
    if (level <= 1) host_trace("begin", name);
    if (!ctx->profiling || ctx->profile_level < level) return;
    hipEvent_t a = get_event(ctx), b = get_event(ctx);
    (void)hipEventRecord(a, ctx->stream);
    ctx->spans[name].pending.push_back({a, b});
}

void span_end(lurkhip_ctx* ctx, const char* name, int level) {
    if (level <= 1) host_trace("end  ", name);
    if (!ctx->profiling || ctx->profile_level < level) return;
    auto& sp = ctx->spans[name];
    if (sp.pending.empty()) return;
    (void)hipEventRecord(sp.pending.back().second, ctx->stream);
}

// ends `from` and begins `to` on one event: back-to-back spans (the stages of a Merkle tree) cost one record, not two --
// every record is a marker packet the next kernel waits behind
void span_switch(lurkhip_ctx* ctx, const char* from, const char* to, int level) {
    if (level <= 1) host_trace("switch to", to);
    if (!ctx->profiling || ctx->profile_level < level) return;
    auto& f = ctx->spans[from];
    if (f.pending.empty()) {
        span_begin(ctx, to, level);
        return;
    }
    hipEvent_t e = f.pending.back().second;
    (void)hipEventRecord(e, ctx->stream);
    hipEvent_t b = get_event(ctx);
    ctx->spans[to].pending.push_back({e, b});
}

static void spans_resolve(lurkhip_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    std::set<hipEvent_t> used;  // an event shared by two spans (span_switch) returns to the pool once
    for (auto& kv : ctx->spans) {
        for (auto& pr : kv.second.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.count += 1;
            }
            used.insert(pr.first);
            used.insert(pr.second);
        }
        kv.second.pending.clear();
    }
    for (hipEvent_t e : used) ctx->event_pool.push_back(e);
}

static int32_t create_common(int32_t device_id, void* stream, bool borrow, lurkhip_ctx** out, int32_t priority = 0) {
    if (!out) return set_error(nullptr, LURKHIP_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return set_error(nullptr, LURKHIP_ERR_NO_DEVICE,
                         "no usable HIP device (%s); lurkhip has no CPU fallback",
                         e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_id < 0 || device_id >= count)
        return set_error(nullptr, LURKHIP_ERR_INVALID_ARG, "device_id %d out of range [0,%d)", device_id, count);
    lurkhip_ctx* ctx = new lurkhip_ctx();
    ctx->device = device_id;
    auto fail = [&](hipError_t err, const char* what) {
        int32_t s = set_error(nullptr, LURKHIP_ERR_HIP, "%s failed: %s", what, hipGetErrorString(err));
        delete ctx;
        return s;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return fail(e, "hipSetDevice");
    if (borrow) {
        ctx->stream = (hipStream_t)stream;
        ctx->owns_stream = false;
    } else {
        // A/B hook (measurements only): LURKHIP_PAD_STREAMS=n creates n idle streams ahead of this context's own, shifting the
        // hardware queue its streams land on (the runtime deals streams to queues in creation order)
        if (const char* pad = getenv("LURKHIP_PAD_STREAMS"))
            for (int k = 0; k < std::min(16, atoi(pad)); k++) {
                hipStream_t idle = nullptr;
                (void)hipStreamCreateWithFlags(&idle, hipStreamNonBlocking);
            }
        int least = 0, greatest = 0;  // numerically: least >= greatest
        if (priority == 0 && getenv("LURKHIP_CTX_PRIORITY")) priority = atoi(getenv("LURKHIP_CTX_PRIORITY"));  // A/B hook
        if (priority != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
            ctx->stream_priority = std::max(greatest, std::min(least, (int)priority));
        if ((e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, ctx->stream_priority)) != hipSuccess)
            return fail(e, "hipStreamCreateWithPriority");
        ctx->owns_stream = true;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) ctx->num_cus = cus;
    }
    ctx->main_stream = ctx->stream;
    if ((e = hipEventCreate(&ctx->ev_start)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) return fail(e, "hipEventCreate");
    *out = ctx;
    return LURKHIP_OK;
}

}  // namespace lurkhip

using namespace lurkhip;

extern "C" {

int32_t lurkhip_abi_version(void) { return 2; }  // 2: protocol profile, lurkhip_proof_read capacity, bytecode import, lurkhip_open

int32_t lurkhip_ctx_create_with_priority(int32_t device_id, int32_t priority, lurkhip_ctx** out) {
    return create_common(device_id, nullptr, false, out, priority);
}

int32_t lurkhip_ctx_create(int32_t device_id, lurkhip_ctx** out) {
    return create_common(device_id, nullptr, false, out);
}

// A context for work that must overlap `other`'s (a second proof in flight).  The runtime deals streams to its four hardware
// queues as they are created and tells nobody where; two streams on one queue run one kernel at a time.  So: create a stream,
// MEASURE whether a kernel on it runs beside a kernel on the other context's stream, and keep the first stream that does
// (streams that do not are parked until the choice is made, so that the next candidate lands elsewhere).  A few milliseconds,
// once per context.  DESIGN.md section 4 has the numbers this is for (40.6 against 45 ms per step, by placement alone).
int32_t lurkhip_ctx_create_beside(lurkhip_ctx* other, lurkhip_ctx** out) {
    LH_CHECK_CTX(other);
    if (!out) return set_error(other, LURKHIP_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    hipStream_t chosen = stream_beside(other->stream, other->stream_priority, PLACE_TICKS, occupy_alone(other->stream, PLACE_TICKS));
    if (!chosen) return set_error(other, LURKHIP_ERR_HIP, "no stream could be created beside the context's");
    lurkhip_ctx* ctx = nullptr;
    const int32_t st = create_common(other->device, chosen, true, &ctx);
    if (st != LURKHIP_OK) {
        (void)hipStreamDestroy(chosen);
        return st;
    }
    ctx->owns_stream = true;  // created here: destroyed with the context
    ctx->stream_priority = other->stream_priority;
    *out = ctx;
    return LURKHIP_OK;
}

// The placement check behind lurkhip_ctx_create_beside as a query: a 0.5 ms one-wave busy kernel alone on `a`'s stream, then one
// on each stream together.  Streams on different hardware queues take the time of one; streams sharing a queue take two.  Both
// streams are drained first; the probe occupies one wave of the device for about a millisecond.
int32_t lurkhip_ctx_overlap_probe(lurkhip_ctx* a, lurkhip_ctx* b, double* alone_s, double* both_s) {
    LH_CHECK_CTX(a);
    if (!b || !alone_s || !both_s) return set_error(a, LURKHIP_ERR_INVALID_ARG, "null argument");
    if (a->device != b->device) return set_error(a, LURKHIP_ERR_INVALID_ARG, "the contexts are on different devices");
    (void)hipStreamSynchronize(b->stream);
    (void)occupy_both(a->stream, b->stream, PLACE_TICKS / 10);  // warm both
    *alone_s = occupy_alone(a->stream, PLACE_TICKS);
    *both_s = occupy_both(a->stream, b->stream, PLACE_TICKS);
    LH_HIP(a, hipGetLastError());
    return LURKHIP_OK;
}

int32_t lurkhip_ctx_create_on_stream(int32_t device_id, void* hip_stream, lurkhip_ctx** out) {
    return create_common(device_id, hip_stream, true, out);
}

int32_t lurkhip_ctx_destroy(lurkhip_ctx* ctx) {
    if (!ctx) return LURKHIP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto it = ctx->cleanups.rbegin(); it != ctx->cleanups.rend(); ++it) (*it)();
    spans_resolve(ctx);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto& kv : ctx->lde_scale_tables) (void)hipFree(kv.second);
    for (auto& kv : ctx->selector_tables) (void)hipFree(kv.second);
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    for (auto& kv : ctx->pool_live) (void)hipFree(kv.first);
    for (int i = 0; i < 4; i++)
        if (ctx->arena[i]) (void)hipFree(ctx->arena[i]);
    if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
    if (ctx->pin_small) (void)hipHostFree(ctx->pin_small);
    for (int i = 0; i < 2; i++) {
        if (ctx->prep_stage[i]) (void)hipHostFree(ctx->prep_stage[i]);
        if (ctx->prep_done[i]) (void)hipEventDestroy(ctx->prep_done[i]);
    }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    for (int k = 0; k < lurkhip_ctx::N_SIDE; k++)
        if (ctx->side_stream[k]) {
            (void)hipStreamSynchronize(ctx->side_stream[k]);
            (void)hipStreamDestroy(ctx->side_stream[k]);
            (void)hipEventDestroy(ctx->side_join[k]);
        }
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    if (ctx->hash_stream) {
        (void)hipStreamSynchronize(ctx->hash_stream);
        (void)hipStreamDestroy(ctx->hash_stream);
        (void)hipEventDestroy(ctx->hash_ready);
        (void)hipEventDestroy(ctx->hash_done);
    }
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return LURKHIP_OK;
}

int32_t lurkhip_ctx_sync(lurkhip_ctx* ctx) {
    LH_CHECK_CTX(ctx);
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LURKHIP_OK;
}

// Cross-context ordering without a host wait: an event recorded behind the work queued on one context's stream, waited for by
// another context's stream (a staging context's uploads before the proving context's trace kernels).
int32_t lurkhip_event_record(lurkhip_ctx* ctx, void** event) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, event != nullptr, "null argument");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t e = (hipEvent_t)*event;
    if (!e) LH_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    LH_HIP(ctx, hipEventRecord(e, ctx->stream));
    *event = e;
    return LURKHIP_OK;
}

int32_t lurkhip_event_wait(lurkhip_ctx* ctx, void* event) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, event != nullptr, "null argument");
    LH_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
    return LURKHIP_OK;
}

int32_t lurkhip_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
    return LURKHIP_OK;
}

const char* lurkhip_last_error(lurkhip_ctx* ctx) {
    if (!ctx) return g_tls_err.c_str();
    static thread_local std::string copy;  // the caller's own copy: another thread's failing call may rewrite ctx->err at any time
    std::lock_guard<std::mutex> lock(ctx->err_mu);
    copy = ctx->err;
    return copy.c_str();
}

int32_t lurkhip_malloc(lurkhip_ctx* ctx, size_t bytes, void** dev_ptr) {
    LH_CHECK_CTX_NOLOCK(ctx);
    LH_ARG(ctx, dev_ptr != nullptr, "null dev_ptr");
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipMalloc(dev_ptr, bytes ? bytes : 16));
    return LURKHIP_OK;
}

int32_t lurkhip_free(lurkhip_ctx* ctx, void* dev_ptr) {
    LH_CHECK_CTX_NOLOCK(ctx);
    if (!dev_ptr) return LURKHIP_OK;
    LH_HIP(ctx, hipStreamSynchronize(ctx->main_stream));  // not ctx->stream: a locked call on another thread may have rerouted it
    LH_HIP(ctx, hipFree(dev_ptr));
    return LURKHIP_OK;
}

int32_t lurkhip_memcpy_h2d(lurkhip_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
    LH_CHECK_CTX(ctx);
    if (bytes == 0) return LURKHIP_OK;
    LH_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LURKHIP_OK;
}

int32_t lurkhip_memcpy_d2h(lurkhip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    LH_CHECK_CTX(ctx);
    if (bytes == 0) return LURKHIP_OK;
    LH_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LURKHIP_OK;
}

int32_t lurkhip_timer_start(lurkhip_ctx* ctx) {
    LH_CHECK_CTX(ctx);
    LH_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return LURKHIP_OK;
}

int32_t lurkhip_timer_stop(lurkhip_ctx* ctx, float* elapsed_ms) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, elapsed_ms != nullptr, "null elapsed_ms");
    LH_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    LH_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
    LH_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    return LURKHIP_OK;
}

int32_t lurkhip_profile_enable(lurkhip_ctx* ctx, int32_t on) {
    LH_CHECK_CTX(ctx);
    ctx->profiling = on != 0;
    ctx->profile_level = on < 0 ? 0 : on;
    // A span takes two events and keeps them until the spans are read; created on demand each was a 30-40 us host call in the
    // middle of a stage (80 us of idle device at the first span of every Merkle tree).  They are created here, ahead of the
    // region being measured.
    if (on > 0) {
        constexpr size_t PREPARED_EVENTS = 2048;
        while (ctx->event_pool.size() < PREPARED_EVENTS) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            ctx->event_pool.push_back(e);
        }
    }
    return LURKHIP_OK;
}

// a caller's own span on the context's stream (e.g. around the trace generation of all chips of a shard)
int32_t lurkhip_profile_span_begin(lurkhip_ctx* ctx, const char* span) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, span != nullptr, "null span name");
    span_begin(ctx, span);
    return LURKHIP_OK;
}

int32_t lurkhip_profile_span_end(lurkhip_ctx* ctx, const char* span) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, span != nullptr, "null span name");
    span_end(ctx, span);
    return LURKHIP_OK;
}

int32_t lurkhip_profile_reset(lurkhip_ctx* ctx) {
    LH_CHECK_CTX(ctx);
    spans_resolve(ctx);
    ctx->spans.clear();
    return LURKHIP_OK;
}

int32_t lurkhip_profile_read(lurkhip_ctx* ctx, const char* span, double* total_ms, int64_t* count) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, span && total_ms && count, "null argument");
    spans_resolve(ctx);
    auto it = ctx->spans.find(span);
    if (it == ctx->spans.end()) {
        *total_ms = 0;
        *count = 0;
        return LURKHIP_OK;
    }
    *total_ms = it->second.total_ms;
    *count = it->second.count;
    return LURKHIP_OK;
}

int32_t lurkhip_pool_trim(lurkhip_ctx* ctx) {
    LH_CHECK_CTX(ctx);
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    ctx->pool_free.clear();
    ctx->pool_cached_bytes = 0;
    return LURKHIP_OK;
}

int32_t lurkhip_pool_stats(lurkhip_ctx* ctx, uint64_t out[6]) {
    LH_CHECK_CTX_NOLOCK(ctx);
    if (!out) return lurkhip::set_error(ctx, LURKHIP_ERR_INVALID_ARG, "lurkhip_pool_stats: out is null");
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    out[0] = ctx->pool_live_bytes;
    out[1] = ctx->pool_cached_bytes;
    out[2] = ctx->pool_peak_bytes;
    out[3] = ctx->pool_mallocs;
    out[4] = ctx->pool_retries;
    out[5] = ctx->lde_scale_bytes;
    return LURKHIP_OK;
}

int32_t lurkhip_pool_reset_peak(lurkhip_ctx* ctx) {
    LH_CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    ctx->pool_peak_bytes = ctx->pool_live_bytes + ctx->pool_cached_bytes;
    return LURKHIP_OK;
}

int32_t lurkhip_debug_inject_alloc_failures(lurkhip_ctx* ctx, int32_t n) {
    LH_CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    ctx->inject_alloc_failures = n < 0 ? 0 : n;
    return LURKHIP_OK;
}

}  // extern "C"
