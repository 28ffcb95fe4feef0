// Internal context object behind the lurkhip C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/lurkhip.h"

struct lurkhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string err;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // grow-only scratch arenas for the host-pointer entry points
    void* arena[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t arena_bytes[4] = {0, 0, 0, 0};
    // lazily created per-ctx device state owned by other translation units (commit.h)
    void* merkle_params_dev = nullptr;
    void* ntt_plans[32] = {};
    std::vector<std::function<void()>> cleanups;  // run in reverse order by lurkhip_ctx_destroy
};

namespace lurkhip {

int32_t set_error(lurkhip_ctx* ctx, int32_t code, const char* fmt, ...);
// returns a device buffer of at least `bytes` from scratch slot `slot` (grown on demand)
int32_t arena_get(lurkhip_ctx* ctx, int slot, size_t bytes, void** out);

}  // namespace lurkhip

#define LH_CHECK_CTX(ctx)                                                            \
    do {                                                                             \
        if (!(ctx)) return lurkhip::set_error(nullptr, LURKHIP_ERR_INVALID_ARG, "null ctx"); \
    } while (0)

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
