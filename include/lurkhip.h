/* lurkhip -- C ABI of the MI355X (gfx950) proving hot path for Lurk.
 *
 * This is the drop-in boundary: every entry point is what a Rust shim for the
 * reference (argumentcomputer/lurk @ v0.5.0) would bind over FFI to replace the
 * CPU implementation cited next to it.  INTEGRATION.md shows the Rust side.
 *
 * Conventions
 *   - every function returns int32_t: 0 = ok, negative = lurkhip_status error;
 *     nothing unwinds across the boundary; lurkhip_last_error() gives the text;
 *   - field elements are uint32_t BabyBear values; `repr` selects how they are
 *     encoded in caller memory: canonical [0,p) or Montgomery (x * 2^32 mod p, the
 *     in-memory form of p3_baby_bear::BabyBear, so RowMajorMatrix<BabyBear> storage
 *     can be passed as is);
 *   - pointers are host pointers unless the function name ends in _dev, in which
 *     case they are device pointers valid on the ctx's device and the call is
 *     asynchronous on the ctx's HIP stream;
 *   - all buffers are caller-owned; matrices are row-major;
 *   - a ctx owns one HIP stream plus scratch arenas; calls on one ctx are
 *     serialized by the caller, different ctxs may be used concurrently from
 *     different threads (the reference calls these seams from rayon workers).
 *   - there is no CPU fallback: without a usable HIP device every compute call
 *     fails with LURKHIP_ERR_NO_DEVICE.
 */
#ifndef LURKHIP_H
#define LURKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lurkhip_ctx lurkhip_ctx;

typedef enum {
    LURKHIP_OK = 0,
    LURKHIP_ERR_INVALID_ARG = -1,
    LURKHIP_ERR_NO_DEVICE = -2,
    LURKHIP_ERR_HIP = -3,
    LURKHIP_ERR_OOM = -4,
    LURKHIP_ERR_UNSUPPORTED = -5,
    LURKHIP_ERR_EXEC = -6, /* Lair execution error (the reference's `bail!`/panic cases) */
    LURKHIP_ERR_PARSE = -7
} lurkhip_status;

#define LURKHIP_REPR_CANONICAL 0
#define LURKHIP_REPR_MONTY 1

#define LURKHIP_BABYBEAR_P 2013265921u
#define LURKHIP_DIGEST_LANES 8

/* ------------------------------------------------------------------ context */

/* ABI version of this header (bumped on any signature change). */
int32_t lurkhip_abi_version(void);

/* Creates a context on HIP device `device_id` with its own non-blocking stream. */
int32_t lurkhip_ctx_create(int32_t device_id, lurkhip_ctx** out);
/* Same, but all work is enqueued on the caller's hipStream_t (e.g. torch's current stream). */
int32_t lurkhip_ctx_create_on_stream(int32_t device_id, void* hip_stream, lurkhip_ctx** out);
int32_t lurkhip_ctx_destroy(lurkhip_ctx* ctx);
/* Blocks until everything enqueued on the ctx's stream has finished. */
int32_t lurkhip_ctx_sync(lurkhip_ctx* ctx);
/* Text of the last error on this ctx (or of the last ctx-less failure on this thread if ctx == NULL). */
const char* lurkhip_last_error(lurkhip_ctx* ctx);

/* Device memory helpers for hosts that do not bring their own allocator. */
int32_t lurkhip_malloc(lurkhip_ctx* ctx, size_t bytes, void** dev_ptr);
int32_t lurkhip_free(lurkhip_ctx* ctx, void* dev_ptr);
int32_t lurkhip_memcpy_h2d(lurkhip_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int32_t lurkhip_memcpy_d2h(lurkhip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);

/* HIP-event stopwatch on the ctx's stream: start records an event, stop records a
 * second one, waits for it and returns the elapsed milliseconds between them. */
int32_t lurkhip_timer_start(lurkhip_ctx* ctx);
int32_t lurkhip_timer_stop(lurkhip_ctx* ctx, float* elapsed_ms);

/* ---------------------------------------------------------------- Poseidon2 */
/* Widths: 4, 8, ..., 48 (the reference's BabyBearConfig4..48,
 * /root/reference/src/poseidon/config.rs:157-287). */

/* Number of Poseidon2Cols columns for `width` (449 / 603 / 755 for 24 / 32 / 40);
 * negative status if the width is not configured.
 * Replaces Poseidon2Cols::num_cols, /root/reference/src/poseidon/wide/columns.rs:38-40. */
int32_t lurkhip_poseidon2_num_cols(int32_t width);

/* out[k] = Poseidon2_width(in[k]) for k < n; in/out are [n][width].
 * Replaces p3 Poseidon2::permute as built by PoseidonConfig::hasher,
 * /root/reference/src/poseidon/config.rs:75-94. */
int32_t lurkhip_poseidon2_permute(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                  int32_t repr);
int32_t lurkhip_poseidon2_permute_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                      int32_t repr);

/* out[k] = first 8 lanes of Poseidon2_width(in[k]); in [n][width], out [n][8].
 * Replaces PoseidonChipset::{hash, execute_simple}, /root/reference/src/core/poseidon.rs:30-38,61-63,
 * and Hasher::hash, /root/reference/src/core/zstore.rs:241-248. */
int32_t lurkhip_poseidon2_hash8(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                int32_t repr);
int32_t lurkhip_poseidon2_hash8_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                    int32_t repr);

/* One wide-witness row per input: out[k] = [8 output lanes | Poseidon2Cols(in[k])],
 * row stride 8 + num_cols(width).
 * Replaces PoseidonChipset::populate_witness, /root/reference/src/core/poseidon.rs:65-72,
 * and Poseidon2Cols::populate, /root/reference/src/poseidon/wide/trace.rs:12-82. */
int32_t lurkhip_poseidon2_wide_witness(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in,
                                       uint32_t* out, int32_t repr);
int32_t lurkhip_poseidon2_wide_witness_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in,
                                           uint32_t* out, int32_t repr);

#ifdef __cplusplus
}
#endif
#endif /* LURKHIP_H */
