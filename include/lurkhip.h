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
    LURKHIP_ERR_PARSE = -7,
    LURKHIP_ERR_VERIFY = -8 /* lurkhip_machine_verify: the proof is rejected (the reference's `verify` returning Err) */
} lurkhip_status;

#define LURKHIP_REPR_CANONICAL 0
#define LURKHIP_REPR_MONTY 1

#define LURKHIP_BABYBEAR_P 2013265921u
#define LURKHIP_DIGEST_LANES 8

/* ------------------------------------------------------------------ context */

/* ABI version of this header (bumped on any signature change). */
int32_t lurkhip_abi_version(void);

/* Threads.  The reference calls this boundary from rayon workers (`Chipset::populate_witness` per row, `generate_trace` per chip:
 * /root/reference/src/lair/trace.rs:86-132,388-406, /root/reference/src/lair/chipset.rs:9-47).  The contract here:
 *   - contexts are independent: any number of host threads may work on their OWN contexts at once (one stream and one pool each);
 *   - ONE context may also be used from several threads: every entry point holds the context's lock until it returns, so the
 *     calls serialise (results are exactly those of some sequential order).  The `*_free` entry points, lurkhip_malloc /
 *     lurkhip_free and lurkhip_pool_stats only take the pool's lock and never wait for a long call.  lurkhip_last_error is
 *     per context, not per thread: with several threads on one context, read it before the next call of any of them;
 *   - host-side objects (toplevel, record, AIR) are read-only in every entry point that takes them `const` and may be shared;
 *   - an entry point makes the context's HIP device current on the calling thread for its duration and restores the previous
 *     one before it returns.
 * tests/test_concurrency_gpu.py holds the contract; tools/tsan_host.sh runs the host-only threads under ThreadSanitizer. */
/* Creates a context on HIP device `device_id` with its own non-blocking stream. */
int32_t lurkhip_ctx_create(int32_t device_id, lurkhip_ctx** out);
/* Same with a stream priority: 0 = the device's default, > 0 lower, < 0 higher (clamped to the device's range), for a context
 * that proves beside another one (a second prove lane, a staging context).  Measured on two half-shards in flight: no difference
 * between equal and unequal queues on this part, so the Python mirror leaves it at 0. */
int32_t lurkhip_ctx_create_with_priority(int32_t device_id, int32_t priority, lurkhip_ctx** out);
/* A context on `other`'s device whose own stream provably runs BESIDE `other`'s: candidate streams are created and a short
 * busy kernel is timed on both until one overlaps (the runtime deals streams to a few hardware queues without saying which; two
 * streams on one queue run one kernel at a time).  For the second proof in flight (lurk_amd.prover.lane_context). */
int32_t lurkhip_ctx_create_beside(lurkhip_ctx* other, lurkhip_ctx** out);
/* Do the two contexts' streams run BESIDE each other?  alone_s = seconds of a 0.5 ms busy kernel on `a`'s stream, both_s = of one on
 * each stream launched together: about alone_s on different hardware queues, about twice that on a shared one.  (What
 * lurkhip_ctx_create_beside measures while it chooses; here as a query, for a placement check that does not depend on load.) */
int32_t lurkhip_ctx_overlap_probe(lurkhip_ctx* a, lurkhip_ctx* b, double* alone_s, double* both_s);
/* Same, but all work is enqueued on the caller's hipStream_t (e.g. torch's current stream). */
int32_t lurkhip_ctx_create_on_stream(int32_t device_id, void* hip_stream, lurkhip_ctx** out);
int32_t lurkhip_ctx_destroy(lurkhip_ctx* ctx);
/* Blocks until everything enqueued on the ctx's stream has finished. */
int32_t lurkhip_ctx_sync(lurkhip_ctx* ctx);
/* Ordering between two contexts' streams without a host wait: record (*event == NULL: created) behind everything queued on
 * ctx's stream; wait makes everything queued later on ctx's stream run after the event. */
int32_t lurkhip_event_record(lurkhip_ctx* ctx, void** event);
int32_t lurkhip_event_wait(lurkhip_ctx* ctx, void* event);
int32_t lurkhip_event_destroy(void* event);
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

/* Per-span device timing with HIP events on the ctx's stream.  While enabled, the library brackets its
 * named stages with event pairs: on = 1 the stages of a proof ("commit_main", "permutation", "commit_perm", "quotient_all",
 * "commit_quotient", "open", "fri_commit", "fri_query", and inside the commitments "lde", "merkle_leaves", "merkle_levels",
 * "merkle_top" for trees of 2^16 leaves and more), on = 2 also the per-chip and per-small-tree spans ("trace_func", "perm_rows",
 * "perm_scan", "quotient": every event record is a marker packet the next kernel waits behind, so these cost about a
 * millisecond per proof).  lurkhip_profile_read waits for the stream and returns the summed milliseconds and the number of
 * brackets of one span since the last reset; lurkhip_profile_span_begin / _end bracket a caller's own span. */
int32_t lurkhip_profile_enable(lurkhip_ctx* ctx, int32_t on);
int32_t lurkhip_profile_span_begin(lurkhip_ctx* ctx, const char* span);
int32_t lurkhip_profile_span_end(lurkhip_ctx* ctx, const char* span);
int32_t lurkhip_profile_reset(lurkhip_ctx* ctx);
int32_t lurkhip_profile_read(lurkhip_ctx* ctx, const char* span, double* total_ms, int64_t* count);
/* Returns cached device blocks of the ctx's allocation pool to the driver. */
int32_t lurkhip_pool_trim(lurkhip_ctx* ctx);
/* Pool accounting: out = {bytes handed out, bytes cached on the free lists, high-water mark of their sum since the last
 * lurkhip_pool_reset_peak, hipMalloc calls, out-of-memory retries, bytes of cached coset-shift tables}.  The high-water mark
 * is the HBM footprint of whatever was proved in between (the reference keeps its traces in host Vecs; no counterpart). */
int32_t lurkhip_pool_stats(lurkhip_ctx* ctx, uint64_t out[6]);
int32_t lurkhip_pool_reset_peak(lurkhip_ctx* ctx);
/* Test hook: the next n device allocations of the pool fail as if the driver were out of memory, which drives the
 * trim-and-retry path of the allocator. */
int32_t lurkhip_debug_inject_alloc_failures(lurkhip_ctx* ctx, int32_t n);

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

/* The narrow Poseidon2 chip: one trace row per round, R_F + R_P + 1 rows per permutation, row =
 * input[W] | is_init | rounds[R_F + R_P] | add_rc[W] | sbox_deg_3[W] | sbox_deg_7[W] | output[W].
 * lurkhip_poseidon2_trace_shape: width of a row and the padded height (next power of two of n * (R_F + R_P + 1)).
 * lurkhip_poseidon2_trace: in [n][width] -> out [height][row width], rows past the last permutation are zero.
 * Replaces Poseidon2Chip::generate_trace, /root/reference/src/poseidon/trace.rs:14-46 (row contents
 * /root/reference/src/poseidon/columns.rs:16-89; BaseAir::width /root/reference/src/poseidon/air.rs:15-19). */
int32_t lurkhip_poseidon2_trace_shape(int32_t width, size_t n, uint32_t* row_width, uint64_t* height);
int32_t lurkhip_poseidon2_trace(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                int32_t repr);
int32_t lurkhip_poseidon2_trace_dev(lurkhip_ctx* ctx, int32_t width, size_t n, const uint32_t* in, uint32_t* out,
                                    int32_t repr);

/* ------------------------------------------------------------------- commit */
/* The commit stage of the STARK prover: coset low-degree extension of each trace matrix followed by
 * one Poseidon2-width-16 Merkle tree over all of them.
 * Replaces p3 TwoAdicFriPcs::commit (Radix2DitParallel::coset_lde_batch + FieldMerkleTreeMmcs::commit)
 * as run by sphinx-core's `StarkMachine::prove::<LocalProver>` for the main / permutation / quotient
 * traces (call site: /root/reference/benches/fib.rs:114-124, /root/reference/src/core/cli/repl.rs:196).
 * The third-party sources are not in /root/reference: parity for this group is UNPINNED. */

typedef struct lurkhip_commitment lurkhip_commitment;

/* Installs the width-16 Poseidon2 parameters of the Merkle hash (canonical values): 8 x 16 external
 * round constants, `rounds_p` (<= 32) internal ones, the 16-entry internal diagonal.  Default: the
 * reference's BabyBearConfig16 (/root/reference/src/poseidon/config.rs:190-199).  A shim that wants
 * sphinx's BabyBearPoseidon2 hash passes RC_16_30 and DiffusionMatrixBabyBear's diagonal here. */
int32_t lurkhip_set_merkle_poseidon2(lurkhip_ctx* ctx, int32_t rounds_p, const uint32_t* ext_rc,
                                     const uint32_t* int_rc, const uint32_t* diag);

/* ------------------------------------------------------------- protocol profile */
/* EVERY choice of the commit / transcript / FRI layer that is restated from memory of the absent third-party sources
 * (sphinx-core @ 8a39b951 over Plonky3 @ a0b92870, /root/reference/Cargo.lock:1626-1852,2528-2592) [UPSTREAM-RECALL] lives in
 * this one structure, per context; the oracle mirrors it field by field (oracle/stark.py: Profile).  A maintainer with the
 * Rust toolchain dumps vectors from sphinx ("Pinning S1"), drops them into tests/golden/upstream/ and flips
 * fields here until the loader tests pass: nothing else in the library encodes a recalled choice.
 * Values are canonical field elements.  struct_bytes must be sizeof(lurkhip_protocol_profile) (ABI check). */
typedef struct lurkhip_protocol_profile {
    uint32_t struct_bytes;
    /* Poseidon2 width 16 (x^7, 8 external rounds) of the Merkle tree, the sponge and the transcript */
    uint32_t p16_rounds_p;            /* internal rounds, <= 32 */
    uint32_t p16_ext_rc[8 * 16];      /* external round constants, round-major */
    uint32_t p16_int_rc[32];          /* internal round constants (lane 0) */
    uint32_t p16_diag[16];            /* internal layer: y_i = scale * (sum_j x_j + diag_i * x_i) */
    uint32_t p16_internal_scale;      /* 1 = the paper's 1 + diag; p3's Montgomery-shift diffusion layer computes 2^-32 (1 + diag) */
    /* p3 DuplexChallenger<_, _, 16, 8>: absorbs 8 lanes by overwrite */
    uint32_t challenger_squeeze;      /* lanes of the state offered as output after a permutation: 8 (the RATE lanes, default) or
                                         16 (the whole state, capacity lanes included: opt-in, preset "whole-state-squeeze") */
    uint32_t challenger_pop_front;    /* 0: sample() pops the END of the output buffer (p3 Vec::pop), 1: the front */
    /* what the shard transcript observes */
    uint32_t observe_openings;        /* 0: alpha_fri is sampled right after zeta (the pinned revision); 1: every opened value is
                                         observed first (the later upstream soundness fix) */
    uint32_t observe_chip_meta;       /* 1: each chip's (machine index, log height) is observed before the permutation challenges
                                         and its cumulative sum before alpha; 0: neither (the pinned revision) */
    /* folding orders */
    uint32_t constraint_alpha_ascending;  /* 0: Horner, first constraint gets the highest power (sphinx folders); 1: constraint k gets alpha^k */
    uint32_t fri_alpha_global;        /* 0: the alpha_fri power offset restarts per LDE height (p3 num_reduced[log_height]); 1: one running offset */
    uint32_t fri_log_arity;           /* 1 (fold by 2); other arities are refused */
    /* machine defaults (sphinx BabyBearPoseidon2: blow-up 2, 100 queries from FRI_QUERIES, 16 proof-of-work bits) */
    uint32_t fri_log_blowup, fri_num_queries, fri_pow_bits;
    /* wire format of field elements in serialized proofs (lurkhip_proof_serialize): 0 canonical u32, 1 Montgomery u32 */
    uint32_t serialize_montgomery;
} lurkhip_protocol_profile;

/* Fills `out` with a named preset: "default" (this build's best recall of the pinned revision, with the reference's in-tree
 * BabyBearConfig16 constants because sphinx's RC_16_30 are not in /root/reference), "hardened" (default + observe_openings +
 * observe_chip_meta), "whole-state-squeeze" (default + challenger_squeeze 16), "p3-monty-diffusion" (default with
 * diag = [-2, 1, 2, 4, ..., 2^13, 2^15] and scale 2^-32). */
int32_t lurkhip_protocol_profile_preset(const char* name, lurkhip_protocol_profile* out);
/* out[k] = Perm16(in[k]) for k < n, states of 16 words: the profile's width-16 Poseidon2 (Merkle compression, leaf sponge,
 * transcript) on the device, one state per lane.  Replaces sphinx `inner_perm().permute` [UPSTREAM-RECALL]
 * (call site: machine.config(), /root/reference/benches/fib.rs:114-120). */
int32_t lurkhip_perm16(lurkhip_ctx* ctx, size_t n, const uint32_t* in, uint32_t* out, int32_t repr);
int32_t lurkhip_perm16_dev(lurkhip_ctx* ctx, size_t n, const uint32_t* in, uint32_t* out, int32_t repr);
/* Installs / reads the context's profile.  Set it before anything is committed or any challenger is created on the ctx. */
int32_t lurkhip_set_protocol_profile(lurkhip_ctx* ctx, const lurkhip_protocol_profile* profile);
int32_t lurkhip_get_protocol_profile(lurkhip_ctx* ctx, lurkhip_protocol_profile* out);

/* out = LDE of the (1 << log_n) x width matrix `in` (evaluations over the size-2^log_n subgroup, natural
 * order) onto the coset 31 * <w_{2^(log_n+log_blowup)}>, rows in bit-reversed order;
 * out is (1 << (log_n + log_blowup)) x width. */
int32_t lurkhip_coset_lde(lurkhip_ctx* ctx, int32_t log_n, int32_t width, int32_t log_blowup, const uint32_t* in,
                          uint32_t* out, int32_t repr);
int32_t lurkhip_coset_lde_dev(lurkhip_ctx* ctx, int32_t log_n, int32_t width, int32_t log_blowup, const uint32_t* in,
                              uint32_t* out, int32_t repr);

/* Commits to n_mats matrices (matrix i is (1 << log_heights[i]) x widths[i], row-major); returns the
 * 8-lane Merkle root and a handle that keeps the LDE matrices and every tree level on the device for
 * later openings.  `mats` is a host array of host (lurkhip_commit) or device (lurkhip_commit_dev)
 * pointers.  keep_coeffs != 0 also keeps the interpolated coefficient matrices. */
/* The MMCS alone: the Merkle commitment of host matrices AS GIVEN (no interpolation, no coset extension) -- p3
 * FieldMerkleTreeMmcs::commit [UPSTREAM-RECALL]; the handle opens like any other commitment.  (An upstream `mmcs_commit`
 * vector is checked against this.) */
int32_t lurkhip_mmcs_commit(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats, const uint32_t* log_heights, const uint32_t* widths,
                            int32_t repr, lurkhip_commitment** out, uint32_t* root);
int32_t lurkhip_commit(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats, const uint32_t* log_heights,
                       const uint32_t* widths, int32_t log_blowup, int32_t repr, int32_t keep_coeffs,
                       lurkhip_commitment** out, uint32_t* root);
int32_t lurkhip_commit_dev(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev,
                           const uint32_t* log_heights, const uint32_t* widths, int32_t log_blowup, int32_t repr,
                           int32_t keep_coeffs, lurkhip_commitment** out, uint32_t* root);
/* lurkhip_commit_dev for matrices with identically-zero columns (round 5): the columns are found by one pass over the matrices
 * (a word per column, read back: the call waits for the device once before the LDE), and their extension -- the zero polynomial's
 * -- is written as zeros by the last LDE pass instead of being computed; the first two passes run on the other columns only.  Same
 * root, same LDE words, same openings as lurkhip_commit_dev (tests/test_commit_sparse_gpu.py); worth it where a good part of the
 * columns is zero -- the selectors and auxiliary columns of branches a Lair shard never takes: a third of a real `(fib N)` shard's
 * main cells, 42 % of its permutation cells (which the shard prover leaves out by itself, without this pass: its permutation
 * kernels know which columns they computed).  aligned_groups != 0: the LDEs of one height in one buffer with a 128-byte-aligned row
 * pitch (lurkhip_commitment_matrix_pitch), the layout of the prover's own commitments.  *zero_columns (may be NULL) receives the
 * number of columns left out.  Blow-up 2, 2^11 ... 2^20 rows take the route; other shapes are committed as by lurkhip_commit_dev. */
int32_t lurkhip_commit_dev_sparse(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev, const uint32_t* log_heights,
                                  const uint32_t* widths, int32_t log_blowup, int32_t repr, int32_t aligned_groups,
                                  lurkhip_commitment** out, uint32_t* root, uint32_t* zero_columns);
/* As lurkhip_commit_dev, for matrices given as evaluations over cosets: the LDE of matrix i is taken with p3's
 * `shift` = shifts[i] (canonical) instead of the generator, i.e. shifts[i] = 31 / (coset shift of matrix i), so that
 * every committed LDE holds the values of the underlying polynomial on 31 * <w> again.  This is how the quotient
 * chunks (evaluations over 31 * w^c * H) are committed (p3 TwoAdicFriPcs::commit: shift = generator / domain.shift). */
int32_t lurkhip_commit_cosets_dev(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev,
                                  const uint32_t* log_heights, const uint32_t* widths, const uint32_t* shifts,
                                  int32_t log_blowup, int32_t repr, lurkhip_commitment** out, uint32_t* root);
int32_t lurkhip_commitment_free(lurkhip_ctx* ctx, lurkhip_commitment* c);
int32_t lurkhip_commitment_root(lurkhip_ctx* ctx, lurkhip_commitment* c, uint32_t* root, int32_t repr);
/* Device pointer (Montgomery form) and shape of the LDE of matrix `index`. */
int32_t lurkhip_commitment_matrix_dev(lurkhip_ctx* ctx, lurkhip_commitment* c, int32_t index,
                                      const uint32_t** lde_dev, uint32_t* log_height, uint32_t* width);
/* Row pitch of that matrix in words: its width for every commitment made through this header's entry points; the prover's own
 * commitments keep the matrices of one height as column ranges of one buffer with a 128-byte-aligned pitch. */
int32_t lurkhip_commitment_matrix_pitch(lurkhip_ctx* ctx, lurkhip_commitment* c, int32_t index, uint32_t* pitch_words);
/* Opens leaf `index` of the tallest LDE: the opened row of every matrix back to back (caller order;
 * matrix m at row index >> (log_max - log_height_m)) and the log_max sibling digests, leaf level first. */
int32_t lurkhip_commitment_open(lurkhip_ctx* ctx, lurkhip_commitment* c, uint64_t index, uint32_t* rows,
                                uint32_t* path, int32_t repr);

/* -------------------------------------------------------------------- traces */
/* Low-level trace kernels (device pointers, asynchronous on the ctx stream).
 *
 * lurkhip_trace_func_dev fills the `height` x width FuncChip trace of one function from
 *   - a degree-resolved micro-program (lurk_amd/csrc/lair/trace_program.h), resident on the device, with
 *     its TH_WORDS-word header also readable on the host;
 *   - per-row arrays for the n_real real rows: args [n][input], outputs [n][output], provides [n][2]
 *     (last_nonce, last_count), depths [n] (partial functions, else NULL);
 *   - a RowMeta[n] table + a word stream holding, per row, the hints (callee outputs / preimages / pointers /
 *     loaded values / callee depths, in bytecode order), then its require records (nonce, count), then its
 *     depth require records.
 * Row i gets nonce nonce_start + i (padding rows included); everything else of a padding row is zero.
 * Replaces FuncChip::generate_trace + populate_row, /root/reference/src/lair/trace.rs:72-135,145-418. */
int32_t lurkhip_trace_func_dev(lurkhip_ctx* ctx, const uint32_t* program_dev, const uint32_t* program_host_header,
                               uint32_t n_real, uint32_t height, uint32_t nonce_start, const uint32_t* args_dev,
                               const uint32_t* outputs_dev, const uint32_t* provides_dev, const uint32_t* depths_dev,
                               const void* meta_dev, const uint32_t* stream_dev, uint32_t* out_dev, int32_t repr);
/* MemChip rows [is_real, ptr, last_nonce, last_count, values...]: values [n_real][len], provides [n_real][2].
 * Replaces MemChip::generate_trace, /root/reference/src/lair/memory.rs:30-69. */
int32_t lurkhip_trace_mem_dev(lurkhip_ctx* ctx, uint32_t len, uint32_t n_real, uint32_t height,
                              const uint32_t* values_dev, const uint32_t* provides_dev, uint32_t* out_dev, int32_t repr);
/* BytesChip main trace 65536 x 13 from records [65536][6][2] (range_u8, range_u16, less_than, and, xor, or).
 * Replaces BytesChip::generate_trace, /root/reference/src/gadgets/bytes/trace.rs:75-101. */
int32_t lurkhip_trace_bytes_dev(lurkhip_ctx* ctx, const uint32_t* records_dev, int32_t is_real, uint32_t* out_dev,
                                int32_t repr);
/* BytesChip preprocessed trace 65536 x 6 [i1, i2, i1 < i2, and, xor, or].
 * Replaces BaseAir::preprocessed_trace, /root/reference/src/gadgets/bytes/trace.rs:49-72. */
int32_t lurkhip_trace_bytes_preprocessed_dev(lurkhip_ctx* ctx, uint32_t* out_dev, int32_t repr);

/* ------------------------------------------------------------------- ZStore */
/* Content-addressed interning of Lurk data with level-order batched hashing.  Replaces the hashing side of
 * ZStore, /root/reference/src/core/zstore.rs:252-267: the hash3 / hash4 / hash5 memo tables (zstore.rs:305-333), the Merkle
 * DAG `dag` (zstore.rs:207-219,335-356) and memoize_dag (zstore.rs:569-702); the ZDag of a cached proof
 * (/root/reference/src/core/cli/zdag.rs:16-55) is exported from it.  The reference hashes one node per Poseidon2 call,
 * children first; here a caller hands over a whole pending DAG and every height of it is one lurkhip_poseidon2_hash8 launch per
 * preimage width. */
typedef struct lurkhip_zstore lurkhip_zstore;
#define LURKHIP_ZNODE_WORDS 10
#define LURKHIP_ZNODE_ATOM 0     /* tag, digest[8]: a leaf, recorded as ZPtrType::Atom (intern_num / _char / _u64 / _comm ...) */
#define LURKHIP_ZNODE_TUPLE11 1  /* tag, a, b: digest = hash4(flatten(a) | flatten(b)), intern_tuple11 (zstore.rs:335-341) */
#define LURKHIP_ZNODE_TUPLE110 2 /* tag, a, b, c: digest = hash5(flatten(a) | flatten(b) | c.digest), intern_tuple110 (zstore.rs:343-349) */
#define LURKHIP_ZNODE_COMM 3     /* -, a, b: Comm with digest = hash3(a.digest | flatten(b)): `hide` / `commit` */
#define LURKHIP_ZNODE_REF 4      /* tag, digest[8]: a pointer interned earlier; nothing is recorded for it */
int32_t lurkhip_zstore_new(lurkhip_ctx* ctx, lurkhip_zstore** out);
int32_t lurkhip_zstore_free(lurkhip_zstore* zs);
const char* lurkhip_zstore_last_error(const lurkhip_zstore* zs);
/* Interns n_nodes nodes given in topological order (children before parents).  nodes[i] = LURKHIP_ZNODE_WORDS words:
 * kind, tag, then 8 digest words (ATOM / REF) or the indices of the children (earlier nodes).  out_zptrs[i] = tag + 8 digest
 * words.  Tags are those of /root/reference/src/core/tag.rs:23-39; all values canonical. */
int32_t lurkhip_zstore_intern_dag(lurkhip_zstore* zs, uint32_t n_nodes, const uint32_t* nodes, uint32_t* out_zptrs);
/* out[6]: permutations computed at widths 24, 32, 40; kernel launches; memo-table hits; entries of the DAG */
int32_t lurkhip_zstore_stats(const lurkhip_zstore* zs, uint64_t* out);
/* The inverse hash tables memoize_dag walks (`hashes4_inv`, `hashes5_inv`: digest -> preimage, in the reference the inverse
 * queries of the hash4 / hash5 functions of an execution): rows of [digest(8) | preimage(32)] and [digest(8) | preimage(40)]. */
int32_t lurkhip_zstore_set_inverse_tables(lurkhip_zstore* zs, uint64_t n4, const uint32_t* inv4, uint64_t n5, const uint32_t* inv5);
/* ZStore::memoize_dag (zstore.rs:569-702): records the Lurk data dependencies of tag/digest from the inverse tables. */
int32_t lurkhip_zstore_memoize_dag(lurkhip_zstore* zs, uint32_t tag, const uint32_t* digest);
/* out[28] = kind (0 Atom, 1 Tuple11, 2 Tuple110), then the children (9 words each, unused ones zero); error if unknown */
int32_t lurkhip_zstore_fetch(const lurkhip_zstore* zs, const uint32_t* zptr, uint32_t* out);
/* ZDag::populate_with_many (cli/zdag.rs:16-55): the DAG entries reachable from n_roots pointers (9 words each), children
 * before parents, each once; an entry is 37 words: zptr (9), kind (1), children (27).  Returns the number of entries (writes at
 * most cap_entries of them; out may be NULL to count), negative when data is missing from the DAG. */
int64_t lurkhip_zstore_dag_export(const lurkhip_zstore* zs, uint32_t n_roots, const uint32_t* roots, uint32_t* out, uint64_t cap_entries);

/* --------------------------------------------------------------------- Lair */
/* Host side of Lair (C++ here because the reference's Rust toolchain is absent): functions written in the
 * surface syntax of the reference's `func!` macro are compiled to bytecode, executed by the memoising
 * interpreter into a query record, and turned into traces on the device.
 * Replaces Toplevel::new / execute, QueryRecord, FuncChip, MemChip, BytesChip, LairChip::generate_trace
 * (/root/reference/src/lair/{toplevel,execute,func_chip,trace,memory,lair_chip}.rs). */
typedef struct lurkhip_toplevel lurkhip_toplevel;
typedef struct lurkhip_record lurkhip_record;

/* with_lurk_chips != 0 registers the native chips of /root/reference/src/core/chipset.rs:28-63
 * (hasher3/4/5, u64_*, big_num_lessthan) for extern_call. */
int32_t lurkhip_toplevel_new(const char* source, int32_t with_lurk_chips, lurkhip_toplevel** out);
/* A toplevel from COMPILED functions: the index-based bytecode of /root/reference/src/lair/bytecode.rs:12-146
 * (Func / Block / Ctrl / Cases / Op) flattened to u32 words, format "LBC1" (the grammar is the header comment of
 * lurk_amd/csrc/lair/bytecode_io.cpp and INTEGRATION.md section 4).  This is how a host with its own compiler -- the
 * reference's `Toplevel::new`, /root/reference/src/lair/toplevel.rs:38-72 -- obtains trace programs, AIRs and proofs for ITS
 * functions without their source text.  Chip names are resolved against the native chips of
 * /root/reference/src/core/chipset.rs:28-63.  The blob is validated (stack references, arities, selector numbering):
 * a malformed one returns LURKHIP_ERR_PARSE with a message in lurkhip_lair_last_error(). */
int32_t lurkhip_toplevel_from_bytecode(const uint32_t* blob, uint64_t n_words, lurkhip_toplevel** out);
/* The same serialisation of a toplevel's compiled functions.  Returns the number of words (also when `out` is NULL or
 * `capacity_words` is too small: nothing is written then), negative on error. */
int64_t lurkhip_toplevel_to_bytecode(const lurkhip_toplevel* top, uint32_t* out, uint64_t capacity_words);
int32_t lurkhip_toplevel_free(lurkhip_toplevel* top);
int32_t lurkhip_toplevel_num_funcs(const lurkhip_toplevel* top);
int32_t lurkhip_toplevel_func_index(const lurkhip_toplevel* top, const char* name);
/* info[9] = input_size, output_size, partial, invertible, layout {nonce, input, output, aux, sel}
 * (LayoutSizes, /root/reference/src/lair/func_chip.rs:11-26,90-116) */
int32_t lurkhip_toplevel_func_info(const lurkhip_toplevel* top, int32_t func_idx, uint32_t* info);
const char* lurkhip_lair_last_error(void);

int32_t lurkhip_record_new(const lurkhip_toplevel* top, lurkhip_record** out);
int32_t lurkhip_record_free(lurkhip_record* r);
int32_t lurkhip_record_clean(lurkhip_record* r);
/* Toplevel::execute: runs func on canonical args, memoising into the record; out gets output_size values.
 * LURKHIP_ERR_EXEC carries the reference's bail!/panic cases ("Loop detected", failed assertions, ...). */
int32_t lurkhip_execute(lurkhip_record* r, int32_t func_idx, const uint32_t* args, uint32_t n_args, uint32_t* out);
int32_t lurkhip_record_inject_inv_query(lurkhip_record* r, int32_t func_idx, const uint32_t* inp, uint32_t n_inp,
                                        const uint32_t* out, uint32_t n_out);
/* kind 0: queries of func `index`; 1: entries of the mem table of length `index`; 2: number of public
 * values (-1 if unset); 3: distinct byte-pair records; 4: emitted lists */
int64_t lurkhip_record_count(const lurkhip_record* r, int32_t kind, int32_t index);
int32_t lurkhip_record_public_values(const lurkhip_record* r, uint32_t* out);
int64_t lurkhip_record_num_shards(const lurkhip_record* r, uint32_t max_shard_size);

int32_t lurkhip_func_trace_shape(const lurkhip_record* r, int32_t func_idx, uint32_t shard_index,
                                 uint32_t max_shard_size, uint32_t* n_real, uint32_t* height, uint32_t* width);
int32_t lurkhip_generate_trace_func(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r,
                                    int32_t func_idx, uint32_t shard_index, uint32_t max_shard_size,
                                    uint32_t* out_host, int32_t repr);
int32_t lurkhip_generate_trace_func_dev(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r,
                                        int32_t func_idx, uint32_t shard_index, uint32_t max_shard_size,
                                        uint32_t* out_dev, int32_t repr);
/* Compiles the function's trace program (the row interpreter's input) to a straight-line gfx950 row kernel with hiprtc and uses it
 * for every later trace of that function on the context's device (the interpreter otherwise); code objects are cached on disk
 * like the AIR kernels'.  _check compiles without loading (no device needed) and returns the code object size; _source returns
 * the generated text.  No reference counterpart (populate_row walks the bytecode per row, src/lair/trace.rs:145-418). */
int32_t lurkhip_trace_compile(lurkhip_ctx* ctx, lurkhip_toplevel* top, int32_t func_idx);
int32_t lurkhip_trace_compile_check(lurkhip_toplevel* top, int32_t func_idx, char* log, uint32_t log_cap);
int32_t lurkhip_trace_source(lurkhip_toplevel* top, int32_t func_idx, char* out, uint32_t cap);
/* generate_trace split in two for callers that re-run it or want the inputs resident in HBM:
 * prepare flattens one shard of one function into device buffers (program, per-row arrays, row stream),
 * run launches the row kernel into a height x width device buffer (asynchronous on the ctx stream).
 * shape[5] = n_real, height, width, bytes of device-resident inputs, stream words. */
typedef struct lurkhip_func_trace lurkhip_func_trace;
int32_t lurkhip_func_trace_prepare(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r, int32_t func_idx,
                                   uint32_t shard_index, uint32_t max_shard_size, lurkhip_func_trace** out);
/* lurkhip_func_trace_prepare for several functions of one shard at once: host threads (n_threads, 0 = one per usable
 * core -- affinity mask capped by the cgroup CPU quota --, at most 32) write the rows of all of them into one page-locked staging buffer and each function's block is queued for
 * upload on the context's stream as soon as it is complete; the call does not wait for the copies (they are ordered before any
 * later work on that stream).  out[i] = NULL for a function without rows in the shard.  This is the per-row parallelism of
 * FuncChip::generate_trace (src/lair/trace.rs:86-132) applied to the part of it that stays on the host. */
int32_t lurkhip_func_trace_prepare_many(lurkhip_ctx* ctx, lurkhip_toplevel* top, const lurkhip_record* r, uint32_t n_funcs,
                                        const int32_t* func_idx, uint32_t shard_index, uint32_t max_shard_size, uint32_t n_threads,
                                        lurkhip_func_trace** out);
/* the same for a MemChip table and for the BytesChip (run with lurkhip_func_trace_run) */
int32_t lurkhip_mem_trace_prepare(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t mem_len, lurkhip_func_trace** out);
int32_t lurkhip_bytes_trace_prepare(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t shard_index, lurkhip_func_trace** out);
int32_t lurkhip_func_trace_shape_of(const lurkhip_func_trace* p, uint64_t* shape);
int32_t lurkhip_func_trace_run(lurkhip_ctx* ctx, const lurkhip_func_trace* p, uint32_t* out_dev, int32_t repr);
/* lurkhip_func_trace_run for every chip of a shard (function, memory and byte chips alike) in one call: the launches are
 * independent, the short chips' kernels go to the context's side streams and run under the tall ones (src/lair/trace.rs:86-132
 * generates the chips of a shard with rayon: par_iter over chips); everything queued on the context afterwards is ordered
 * behind all of them. */
int32_t lurkhip_func_trace_run_many(lurkhip_ctx* ctx, uint32_t n, const lurkhip_func_trace* const* ps, uint32_t* const* outs_dev, int32_t repr);
/* Row pitches (ABI 2, round 5).  RowMajorMatrix::new(values, width) (src/lair/trace.rs:133) is dense; a caller that owns the
 * allocation can do better: the traces of one height as column ranges of ONE device buffer [height][pitch], pitch a multiple of
 * 32 words, so that every row and every 32-column tile of the coset LDE's first pass starts on a 128-byte line (dense 78- and
 * 148-word rows straddle lines on every tile: 1.5 x read amplification measured).  lurkhip_trace_group_layout says how the
 * library would lay out n matrices of the given shapes: matrix i belongs to buffer groups[i] (0 .. *n_groups - 1; a buffer
 * holds 2^log_heights[i] rows of pitches[i] words) from column col_starts[i] on; a matrix that is not worth padding gets a
 * buffer of its own with pitches[i] = widths[i].  The _pitched variants of run / run_many / shard_commit take outs_dev[i] =
 * buffer + col_starts[i] and the pitch in words (>= the trace's width; NULL or 0: dense); words of a row beyond a matrix's
 * columns are never read or written by the library.  Any layout is valid -- results do not depend on it.  Measured on the
 * fib-mix step: the first LDE pass does not gain from aligned sources and the trace kernels lose what their
 * contiguous tile stores had, so lurkhip_trace_group_layout answers "dense" unless LURKHIP_SRC_PADDED=1 is set. */
int32_t lurkhip_trace_group_layout(uint32_t n, const uint32_t* log_heights, const uint32_t* widths, uint32_t* pitches,
                                   uint32_t* col_starts, int32_t* groups, int32_t* n_groups);
int32_t lurkhip_func_trace_run_pitched(lurkhip_ctx* ctx, const lurkhip_func_trace* p, uint32_t* out_dev, uint32_t out_pitch, int32_t repr);
/* rows [first_row, first_row + n_rows) of the trace only, into out_dev[n_rows][out_pitch] (out_pitch 0: dense): a rank's block of
 * rows of one shard proved by several ranks (lurkhip_shard_commit_split, main_row_blocks).  Bit-identical to the same rows of
 * lurkhip_func_trace_run's matrix (tests/test_split_gpu.py). */
int32_t lurkhip_func_trace_run_rows(lurkhip_ctx* ctx, const lurkhip_func_trace* p, uint32_t first_row, uint32_t n_rows, uint32_t* out_dev,
                                    uint32_t out_pitch, int32_t repr);
int32_t lurkhip_func_trace_run_many_pitched(lurkhip_ctx* ctx, uint32_t n, const lurkhip_func_trace* const* ps, uint32_t* const* outs_dev,
                                            const uint32_t* out_pitches, int32_t repr);
/* A prepared trace as bytes (round 5): the process that executed the program hands a shard's kernel inputs to the process
 * (GPU) that proves the shard -- `Shard::shard` (src/lair/execute.rs:186-216) cuts ONE QueryRecord, which only the executing
 * process holds.  export: the handle's numbers and its device block into a host buffer of lurkhip_func_trace_export_size bytes
 * (waits for the context's stream); import: a new handle on `ctx` (any device) that lurkhip_func_trace_run* take like one made
 * by the prepare calls.  The blob is validated for shape, not for content: it is as trusted as a bytecode blob. */
int32_t lurkhip_func_trace_export_size(const lurkhip_func_trace* p, uint64_t* bytes);
int32_t lurkhip_func_trace_export(lurkhip_ctx* ctx, const lurkhip_func_trace* p, void* out_host, uint64_t bytes);
int32_t lurkhip_func_trace_import(lurkhip_ctx* ctx, const void* blob_host, uint64_t bytes, lurkhip_func_trace** out);
int32_t lurkhip_func_trace_free(lurkhip_ctx* ctx, lurkhip_func_trace* p);
int32_t lurkhip_mem_trace_shape(const lurkhip_record* r, uint32_t mem_len, uint32_t* n_real, uint32_t* height,
                                uint32_t* width);
int32_t lurkhip_generate_trace_mem(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t mem_len, uint32_t* out_host,
                                   int32_t repr);
int32_t lurkhip_generate_trace_bytes(lurkhip_ctx* ctx, const lurkhip_record* r, uint32_t shard_index,
                                     uint32_t* out_host, int32_t repr);

/* ---------------------------------------------------------------- AIR + prover stages */
/* The constraints and lookup interactions of one chip, collected once on the host by a symbolic walk of the
 * reference's AIR code and lowered to a register program the GPU evaluates per row.
 * Replaces `Air::eval` of FuncChip (/root/reference/src/lair/air.rs:158-552), MemChip
 * (/root/reference/src/lair/memory.rs:71-109), BytesChip (/root/reference/src/gadgets/bytes/trace.rs:117-143) and the
 * Entrypoint chip (/root/reference/src/lair/lair_chip.rs:166-191) as sphinx-core's symbolic / debug / folding builders
 * run them; provide / require follow /root/reference/src/air/builder.rs:42-104. */
typedef struct lurkhip_air lurkhip_air;
int32_t lurkhip_air_func(const lurkhip_toplevel* top, int32_t func_idx, lurkhip_air** out);
int32_t lurkhip_air_mem(uint32_t len, lurkhip_air** out);
int32_t lurkhip_air_bytes(lurkhip_air** out);
int32_t lurkhip_air_entrypoint(uint32_t func_idx, uint32_t num_public_values, lurkhip_air** out);
/* Air::eval of the narrow Poseidon2 chip, /root/reference/src/poseidon/air.rs:21-165 (no lookups) */
int32_t lurkhip_air_poseidon2(int32_t width, lurkhip_air** out);
int32_t lurkhip_air_free(lurkhip_air* air);
const char* lurkhip_air_name(const lurkhip_air* air);
/* info[16]: 0 width, 1 preprocessed width, 2 #constraints, 3 #sends, 4 #receives, 5 max constraint degree,
 * 6 log_quotient_degree, 7 permutation-trace width in extension-field columns, 8 words per row of the interaction
 * dump (sum of 1 + tuple length), 9 #public values, 10/11 registers / instructions of the constraint program,
 * 12/13 of the interaction program, 14 number of pieces the interaction program is cut into for the prover kernels (one
 * wave per piece), 15 instructions summed over the pieces */
int32_t lurkhip_air_info(const lurkhip_air* air, uint32_t* info);
/* Compiles the chip's AIR program pieces to straight-line device code (hiprtc) and uses the compiled kernels for its permutation
 * traces and quotients on this context's device from then on (the interpreter otherwise).  Costs seconds to tens of seconds of
 * host time per chip: for traces of 2^17 rows and more.  No reference counterpart (sphinx evaluates `Air::eval` natively). */
int32_t lurkhip_air_compile(lurkhip_ctx* ctx, lurkhip_air* air);
/* Generates and compiles the chip's kernels without loading them (no device needed): code object size in bytes, or a negative
 * error with the compiler's message in log. */
int32_t lurkhip_air_compile_check(const lurkhip_air* air, char* log, uint32_t log_cap);
/* The lowered register programs (csrc/air_program.h), for inspection and tests.  which: 0 constraints, 1 interactions (whole),
 * 2 interaction pieces of the permutation-trace kernel, 3 pieces of the quotient kernel, 4 constraint pieces of the quotient kernel
 * (header word 10 = index of the piece's first constraint).  Returns the word count (copies at most cap words), negative for an
 * unknown program. */
int32_t lurkhip_air_program(const lurkhip_air* air, int32_t which, uint32_t index, uint32_t* out, uint32_t cap);
/* tuple length of each interaction, sends first then receives; returns their number */
int32_t lurkhip_air_interaction_sizes(const lurkhip_air* air, uint32_t* sizes, uint32_t cap);
/* Debug / parity entry: evaluates every constraint and every interaction on explicit (local, next) row pairs with
 * explicit selector values [n][3] = (is_first_row, is_last_row, is_transition).  Host pointers, canonical values.
 * constraints_out [n][#constraints] in assertion order; interactions_out [n][info[8]]: per interaction (sends
 * first) the multiplicity followed by the tuple. */
int32_t lurkhip_air_eval_rows(lurkhip_ctx* ctx, lurkhip_air* air, uint32_t n_rows, const uint32_t* local,
                              const uint32_t* next, const uint32_t* prep_local, const uint32_t* prep_next,
                              const uint32_t* public_values, const uint32_t* selectors, uint32_t* constraints_out,
                              uint32_t* interactions_out);
/* Row-by-row constraint check of a whole trace on the device (the twin of sphinx's machine.debug_constraints /
 * /root/reference/src/air/debug.rs:161-206): main_dev / prep_dev are height x width Montgomery matrices in natural
 * row order; public_values host canonical.  first_bad_row = -1 when every constraint vanishes on every row. */
int32_t lurkhip_air_check_trace_dev(lurkhip_ctx* ctx, lurkhip_air* air, uint32_t height, const uint32_t* main_dev,
                                    const uint32_t* prep_dev, const uint32_t* public_values, int64_t* first_bad_row,
                                    int32_t* first_bad_constraint);
/* LogUp permutation trace of one chip (sphinx generate_permutation_trace [UPSTREAM-RECALL]; spec twin
 * /root/reference/src/logup/trace.rs:53-151).  main_dev / prep_dev: height x width Montgomery, natural order.
 * challenges[8] = alpha, beta (canonical extension-field coefficients, host).  out_dev: height x (4 * info[7])
 * Montgomery words (extension-field columns flattened to base), last column = running sum.  cumulative_sum[4]
 * (host, canonical, may be NULL) receives the last row's running sum; passing it makes the call synchronous. */
int32_t lurkhip_permutation_trace_dev(lurkhip_ctx* ctx, lurkhip_air* air, uint32_t height, const uint32_t* main_dev,
                                      const uint32_t* prep_dev, const uint32_t* challenges, uint32_t* out_dev,
                                      uint32_t* cumulative_sum);

/* Quotient values of one chip on the quotient domain 31 * <w_Q>, Q = 2^(log_n + info[6])
 * (sphinx quotient_values [UPSTREAM-RECALL]): every constraint of the chip, the permutation constraints of its
 * interactions and the three running-sum constraints folded with powers of `alpha`, divided by Z_H.
 * main / prep / perm _lde_dev: the committed LDE matrices (lurkhip_commitment_matrix_dev: bit-reversed rows, Montgomery);
 * their height must be at least Q.  perm_challenges[8] as in lurkhip_permutation_trace_dev; alpha[4],
 * cumulative_sum[4], public_values: host, canonical.  out_dev: 2^info[6] chunk matrices back to back, each
 * 2^log_n x 4 Montgomery words: chunk c row r = quotient at 31 * w_Q^(r * 2^info[6] + c), i.e. the evaluations of
 * chunk c over the coset 31 * w_Q^c * H that lurkhip_commit_cosets_dev takes with shift w_Q^-c. */
int32_t lurkhip_quotient_dev(lurkhip_ctx* ctx, lurkhip_air* air, uint32_t log_n, const uint32_t* main_lde_dev,
                             const uint32_t* prep_lde_dev, const uint32_t* perm_lde_dev, const uint32_t* perm_challenges,
                             const uint32_t* alpha, const uint32_t* cumulative_sum, const uint32_t* public_values,
                             uint32_t* out_dev);

/* ---------------------------------------------------------------- transcript + shard prover */
/* Fiat-Shamir transcript (p3 DuplexChallenger<BabyBear, Poseidon2-16, 16, 8> [UPSTREAM-RECALL]) over the same width-16
 * permutation as the Merkle tree; host state.  Values are canonical. */
typedef struct lurkhip_challenger lurkhip_challenger;
int32_t lurkhip_challenger_new(lurkhip_ctx* ctx, lurkhip_challenger** out);
int32_t lurkhip_challenger_clone(const lurkhip_challenger* src, lurkhip_challenger** out);
int32_t lurkhip_challenger_free(lurkhip_challenger* c);
int32_t lurkhip_challenger_observe(lurkhip_challenger* c, const uint32_t* values, uint32_t n);
int32_t lurkhip_challenger_sample(lurkhip_challenger* c, uint32_t* out, uint32_t n);
int32_t lurkhip_challenger_sample_bits(lurkhip_challenger* c, uint32_t bits, uint32_t* out);

/* StarkMachine::setup: commits the preprocessed traces (height-sorted, tallest first; device, Montgomery, natural row
 * order; they must outlive the key).  n_prep may be 0.  Replaces machine.setup(&LairMachineProgram),
 * /root/reference/benches/fib.rs:120. */
typedef struct lurkhip_pk lurkhip_pk;
int32_t lurkhip_setup(lurkhip_ctx* ctx, int32_t n_prep, const uint32_t* const* prep_traces_dev, const uint32_t* log_heights,
                      const uint32_t* widths, int32_t log_blowup, lurkhip_pk** out, uint32_t* root);
int32_t lurkhip_pk_free(lurkhip_ctx* ctx, lurkhip_pk* pk);

/* LocalProver::commit_shards for one shard: orders the chips by height (tallest first, stable) and commits their
 * main traces (device, Montgomery, natural row order, 2^log_heights[i] x width(airs[i]); they must outlive the shard).
 * prep_indices[i] = index of chip i's preprocessed trace in the key or -1 (may be NULL). */
typedef struct lurkhip_shard lurkhip_shard;
int32_t lurkhip_shard_commit(lurkhip_ctx* ctx, int32_t n_chips, lurkhip_air* const* airs, const uint32_t* log_heights,
                             const uint32_t* const* main_traces_dev, const int32_t* prep_indices, int32_t log_blowup,
                             lurkhip_shard** out, uint32_t* root);
/* the same with main_pitches[i] words between the rows of main_traces_dev[i] (lurkhip_trace_group_layout; NULL: dense) */
int32_t lurkhip_shard_commit_pitched(lurkhip_ctx* ctx, int32_t n_chips, lurkhip_air* const* airs, const uint32_t* log_heights,
                                     const uint32_t* const* main_traces_dev, const uint32_t* main_pitches, const int32_t* prep_indices,
                                     int32_t log_blowup, lurkhip_shard** out, uint32_t* root);
int32_t lurkhip_shard_free(lurkhip_ctx* ctx, lurkhip_shard* shard);

/* LocalProver::prove_shard: permutation traces, quotient, openings and FRI for a committed shard.  The challenger
 * must already have observed what the machine observes before per-shard challenges are drawn (preprocessed root, 0,
 * every shard's main root and public values); it is advanced exactly as the verifier's will be.  The proof is a flat
 * array of canonical words (layout: lurk_amd/prover.py). Replaces machine.prove::<LocalProver>(..) per shard,
 * /root/reference/benches/fib.rs:124, /root/reference/src/core/cli/repl.rs:196. */
typedef struct lurkhip_proof lurkhip_proof;
int32_t lurkhip_shard_prove(lurkhip_ctx* ctx, const lurkhip_pk* pk, lurkhip_shard* shard, lurkhip_challenger* challenger,
                            const uint32_t* public_values, uint32_t n_public, uint32_t num_queries, uint32_t pow_bits,
                            lurkhip_proof** out);
/* out[0] = base-field cells of the permutation traces of the context's last shard proof, out[1] = those of them in columns whose
 * extension was computed (the identically-zero columns -- interactions no row of the shard uses -- are left out of the LDE and
 * stored as zeros): what a bench's "algorithmic bytes" and "bytes transformed" differ by. */
int32_t lurkhip_prover_stats(lurkhip_ctx* ctx, uint64_t* out /* [2] */);
/* StarkMachine::verify on the HOST (no device, no context): rebuilds the transcript from the verifying key and the shard proofs
 * (in shard order), checks every Merkle opening, every FRI query with its proof of work, the constraint identity of every chip
 * at zeta -- evaluated from airs[machine index], the same AIR handles the prover was given -- and that the cumulative sums of all
 * chips of all shards cancel.  `profile` NULL = the "default" preset; prep_log_heights / prep_widths describe the preprocessed
 * traces committed by lurkhip_setup (n_prep of them, in key order); proofs[s] / proof_words[s] = the words of shard s
 * (lurkhip_proof_words).  Returns LURKHIP_OK, or LURKHIP_ERR_VERIFY with the reason in `err` (NUL-terminated, at most err_cap
 * bytes).  Replaces `machine.verify(&vk, &proof, &mut challenger)` (sphinx StarkMachine::verify [UPSTREAM-RECALL];
 * /root/reference/benches/fib.rs:105-133, /root/reference/src/lair/lair_chip.rs:246-276). */
int32_t lurkhip_machine_verify(const struct lurkhip_protocol_profile* profile, const lurkhip_air* const* airs, uint32_t n_airs,
                               const uint32_t* vk_root, const uint32_t* prep_log_heights, const uint32_t* prep_widths, uint32_t n_prep,
                               const uint32_t* const* proofs, const uint64_t* proof_words, uint32_t n_proofs, char* err,
                               uint32_t err_cap);
/* The same verification from the reference's serialised form: `bytes` = bincode of a `CryptoProof` (lurkhip_crypto_proof_bincode;
 * /root/reference/src/core/cli/proofs.rs:22-35), which carries neither the public values (the caller rebuilds the 44 lanes from
 * the claim: proofs.rs:46-56), nor the FRI parameters (the machine's: num_queries, pow_bits, log_blowup), nor the query indices and
 * the queried half of each FRI pair (re-derived).  chip_names[k] = name of machine index k (lurkhip_air_name), resolving the
 * proof's `chip_ordering`.  The `verify` path of the reference's CLI: load a CachedProof, rebuild the public values, then
 * `machine.verify` (/root/reference/src/core/cli/proofs.rs:94-131).  Host only. */
int32_t lurkhip_crypto_proof_verify(const struct lurkhip_protocol_profile* profile, const lurkhip_air* const* airs, const char* const* chip_names,
                                    uint32_t n_airs, const uint32_t* vk_root, const uint32_t* prep_log_heights, const uint32_t* prep_widths,
                                    uint32_t n_prep, const uint8_t* bytes, uint64_t n_bytes, const uint32_t* public_values, uint32_t n_public,
                                    uint32_t num_queries, uint32_t pow_bits, uint32_t log_blowup, char* err, uint32_t err_cap);
/* ... and from a `CachedProof { crypto_proof, expr, env, result, zdag }` (lurkhip_cached_proof_bincode;
 * /root/reference/src/core/cli/proofs.rs:137-169): the claim travels with the proof, so the 44 public values are rebuilt from
 * expr / env / result / depth as `into_machine_proof` does (proofs.rs:46-56,94-131) and returned in public_values_out (44 words,
 * may be NULL); the ZDag is checked for its framing only.  What `lurk verify <key>` does after loading the file.  Host only. */
int32_t lurkhip_cached_proof_verify(const struct lurkhip_protocol_profile* profile, const lurkhip_air* const* airs, const char* const* chip_names,
                                    uint32_t n_airs, const uint32_t* vk_root, const uint32_t* prep_log_heights, const uint32_t* prep_widths,
                                    uint32_t n_prep, const uint8_t* bytes, uint64_t n_bytes, uint32_t num_queries, uint32_t pow_bits,
                                    uint32_t log_blowup, uint32_t* public_values_out, char* err, uint32_t err_cap);
/* Pcs::open on its own (p3 TwoAdicFriPcs::open as sphinx calls it from prove_shard [UPSTREAM-RECALL]): opens the
 * matrices of n_rounds commitments (lurkhip_commit / lurkhip_commit_dev / lurkhip_commit_cosets_dev handles, all with the same
 * blow-up) at caller-chosen extension-field points and proves the openings with FRI.  n_points[k] (1 or 2) is the number of
 * points of the k-th matrix, counting matrices round by round in committed order; `points` holds those points in the same
 * order, 4 canonical words each.  The challenger must be in the state the verifier's will be in before it reads the opened
 * values; it is advanced through alpha, the FRI betas, the proof-of-work check and the query indices.  The result is a flat
 * array of canonical words read with lurkhip_proof_words / lurkhip_proof_read (layout: lurk_amd/commit.py parse_opening):
 * opened values [round][matrix][point][column], FRI layer roots, final polynomial, proof-of-work witness, query indices, the
 * Merkle openings of every round and layer. */
int32_t lurkhip_open(lurkhip_ctx* ctx, int32_t n_rounds, lurkhip_commitment* const* commitments, const uint32_t* n_points,
                     const uint32_t* points, lurkhip_challenger* challenger, uint32_t num_queries, uint32_t pow_bits,
                     lurkhip_proof** out);
int64_t lurkhip_proof_words(const lurkhip_proof* proof);
/* copies the proof's words; capacity_words (the size of `out`) must be at least lurkhip_proof_words(proof) */
int32_t lurkhip_proof_read(const lurkhip_proof* proof, uint32_t* out, uint64_t capacity_words);
int32_t lurkhip_proof_free(lurkhip_proof* proof);

/* ------------------------------------------------------------------- the in-tree LogUp module */
/* Device twin of /root/reference/src/logup/ -- dead code upstream (`// pub mod logup;`, src/lib.rs:6); the permutation argument
 * that runs is sphinx's (lurkhip_permutation_trace_dev).  Host pointers, canonical words, extension elements as 4 words.
 * `program` is the flat form of the interactions (PairColLC, src/air/symbolic/virtual_col.rs:8-13), provides first:
 *   n_provides, n_requires, then per interaction: has_is_real, [lc], n_values, lc * n_values;
 *   lc = n_terms, (kind, index, weight) * n_terms, constant;  kind 0 = identity column, 1 = preprocessed, 2 = main.
 *
 * lurkhip_logup_multiplicities     = generate_multiplicities_trace (logup/trace.rs:10-50): out[row][i] = sum_j z^(traces_i[j] + 1)
 *   counts_i[row][j]; `traces` holds the n_traces[i] trace indices of every provide back to back, counts[i] is [height][n_traces[i]].
 * lurkhip_logup_permutation_trace  = generate_permutation_trace (logup/trace.rs:53-151): rows [s, 1/d_0, 1/d_1, ...] with
 *   d_k = r + sum_j gamma^j v_kj (gamma^0 = 1, logup/air.rs:79-109), 0 where is_real evaluates to 0, column 0 the running sum of
 *   sum_k m_k / d_k (m_k = the multiplicity witness for provides, -z for requires): inclusive as upstream computes it, or --
 *   exclusive != 0 -- the s_0 = 0 form its constraints ask for.  `sum` (may be null) receives the total.
 * lurkhip_logup_eval_constraints   = eval_logup_constraints (logup/air.rs:11-77) on n_rows row pairs: per pair the n_int inverse
 *   constraints, then first-row, transition and last-row constraints (extension values).  air_order != 0 walks the interactions
 *   requires-then-provides as the AIR does (air.rs:37; its multiplicities still come provides-first, air.rs:39-43), 0 provides
 *   first like the trace generator.  selectors[row] = is_first_row, is_last_row, is_transition. */
int32_t lurkhip_logup_multiplicities(lurkhip_ctx* ctx, uint32_t height, uint32_t n_provides, const uint32_t* n_traces, const uint32_t* traces,
                                     const uint32_t* const* counts, const uint32_t* z, uint32_t* out);
int32_t lurkhip_logup_permutation_trace(lurkhip_ctx* ctx, uint32_t height, uint32_t prep_width, uint32_t main_width, const uint32_t* identity,
                                        const uint32_t* prep, const uint32_t* main, const uint32_t* multiplicities, const uint32_t* program,
                                        uint64_t program_words, const uint32_t* z, const uint32_t* r, const uint32_t* gamma, int32_t exclusive,
                                        uint32_t* out, uint32_t* sum);
int32_t lurkhip_logup_eval_constraints(lurkhip_ctx* ctx, uint32_t n_rows, uint32_t prep_width, uint32_t main_width, const uint32_t* perm_local,
                                       const uint32_t* perm_next, const uint32_t* multiplicities, const uint32_t* identity, const uint32_t* prep,
                                       const uint32_t* main, const uint32_t* program, uint64_t program_words, const uint32_t* z, const uint32_t* r,
                                       const uint32_t* gamma, const uint32_t* final_sum, const uint32_t* selectors, int32_t air_order, uint32_t* out);

/* ------------------------------------------------------------------- multi-GPU: the two collectives of a sharded proof */
/* One process per GPU.  What the reference does inside one process -- sphinx's prover observes the main-trace
 * commitment of EVERY shard (`Shard::shard`, /root/reference/src/lair/execute.rs:186-241) into the shared challenger before any
 * shard's challenges are drawn, and the verifier requires the chips' cumulative sums over all shards to cancel
 * (/root/reference/src/lair/lair_chip.rs:104-139) -- becomes an all-gather of (shard index, root) records and an all-reduce of
 * extension-field sums over RCCL (xGMI), both enqueued on the context's stream.  librccl is loaded when the first
 * communicator is asked for.  Rank 0 draws the id and the HOST distributes it (the Rust prover: over whatever channel
 * started its ranks; the Python mirror: torch.distributed.broadcast_object_list). */
typedef struct lurkhip_comm lurkhip_comm;
#define LURKHIP_COMM_ID_BYTES 128      /* ncclUniqueId */
#define LURKHIP_ROOT_RECORD_WORDS 9    /* shard index, then the 8 words of the shard's main-trace root */
int32_t lurkhip_comm_unique_id(uint8_t* id_out /* [LURKHIP_COMM_ID_BYTES] */);
/* Which librccl the library bound (loading it if need be): LURKHIP_RCCL_LIB when set (and nothing else), else a librccl the
 * process has already mapped (PyTorch's bundled copy: one RCCL per process), else the system's by name.  NULL when none could be
 * loaded; lurkhip_last_error(NULL) then says why. */
const char* lurkhip_comm_library(void);
/* Collective over all `world` ranks (ncclCommInitRank); the context's device is the rank's GPU. */
int32_t lurkhip_comm_create(lurkhip_ctx* ctx, const uint8_t* id, int32_t rank, int32_t world, lurkhip_comm** out);
int32_t lurkhip_comm_destroy(lurkhip_ctx* ctx, lurkhip_comm* comm);
int32_t lurkhip_comm_info(const lurkhip_comm* comm, int32_t* rank, int32_t* world);
/* gathered_dev[world * n_local][9] <- every rank's records_dev[n_local][9], in rank order; every rank passes the same n_local.
 * Device pointers; returns when the all-gather is enqueued. */
int32_t lurkhip_exchange_roots_dev(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* records_dev, int32_t n_local, uint32_t* gathered_dev);
/* Host convenience: this rank's (shard_indices[i], roots[i][8]) in, roots_out[shard][8] of ALL world * n_local shards out, in shard
 * order (what every shard's transcript observes); fails unless the ranks' indices are a partition of 0 .. world * n_local - 1.
 * Every rank passes the same n_local (checked: the ranks all-gather their counts first, so a disagreement is an error on every
 * rank, not a hang). */
int32_t lurkhip_exchange_roots(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* shard_indices, const uint32_t* roots, int32_t n_local,
                               uint32_t* roots_out);
/* The same for ANY number of shards per rank (`Shard::shard` cuts an execution into ceil(rows / max_shard_size) shards,
 * /root/reference/src/lair/execute.rs:186-216 -- nine shards on eight GPUs is an ordinary case): n_local >= 0 may differ from rank
 * to rank and be 0; n_total = the number of shards of the execution, the same on every rank; roots_out[n_total][8].  Two all-gathers:
 * the counts (one word per rank; a rank whose arguments are unusable sends -1 and EVERY rank returns LURKHIP_ERR_INVALID_ARG),
 * then records padded to the largest count.  Fails on every rank alike unless the indices are a partition of 0 .. n_total - 1. */
int32_t lurkhip_exchange_roots_var(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* shard_indices, const uint32_t* roots, int32_t n_local,
                                   int32_t n_total, uint32_t* roots_out);
/* lanes_dev[4] (int64: sums of canonical coefficients, one lane per extension-field coefficient) all-reduced in place, then
 * total_dev[4] <- lanes mod p.  RCCL has no modular reduction: addends below 2^31 cannot overflow 64 bits over any node. */
int32_t lurkhip_reduce_sums_dev(lurkhip_ctx* ctx, lurkhip_comm* comm, int64_t* lanes_dev, uint32_t* total_dev);
/* Host convenience: local_sums[n_sums][4] canonical extension-field elements (the cumulative sums of every chip of every shard
 * this rank proved) -> total[4], the machine-wide sum on every rank: a consistent proof set gives zero. */
int32_t lurkhip_reduce_sums(lurkhip_ctx* ctx, lurkhip_comm* comm, const uint32_t* local_sums, int32_t n_sums, uint32_t* total);

/* ------------------------------------------------------------------- multi-GPU: ONE shard over G ranks */
/* `Shard::shard` (/root/reference/src/lair/execute.rs:186-241) only cuts an execution above 2^22 rows, so
 * `machine.prove::<LocalProver>` (/root/reference/benches/fib.rs:124, /root/reference/src/core/cli/repl.rs:196) of anything
 * smaller is ONE shard: shard -> rank leaves all GPUs but one idle.  These entry points let G = 2, 4, 8 .. ranks prove one shard
 * together and return, on every rank, the words lurkhip_shard_prove returns on one (second bullet):
 * coset LDEs on column tiles, ONE all-to-all to contiguous storage-row blocks (lurk_amd/csrc/split_plan.h), leaf hashing and tree
 * levels on the rank's rows, the G subtree roots all-gathered and the top log2 G levels computed by every rank; permutation traces
 * by row blocks with the block totals of the running sum all-gathered; quotient values on the rank's rows; opened values as partial
 * sums all-reduced (64-bit lanes, reduced mod p locally); reduced openings row-local, all-gathered, FRI on every rank; query rows
 * and paths from the rank that owns them.
 *
 * The collectives are callbacks so that the host decides what carries them: lurkhip_comm_split_vtable fills the struct for an
 * RCCL communicator of this library (device buffers stay on the device, everything enqueued on the context's stream); a test
 * fills it with functions that stage through host memory and torch.distributed's gloo.  Every callback returns 0 or a negative
 * status; all ranks call the same collectives in the same order.  `*_dev` pointers are device memory of the calling rank and
 * `hip_stream` the stream the buffers were produced on: an implementation that leaves the device synchronises it first and
 * returns with the result in place. */
typedef struct lurkhip_split_comm {
    int32_t rank, world; /* world: a power of two >= 2 */
    void* user;
    /* rank d receives send_dev[send_off[d] .. send_off[d + 1]) of every rank s at recv_dev[recv_off[s] ..]; offsets in 32-bit words */
    int32_t (*alltoallv_dev)(void* user, const uint32_t* send_dev, const uint64_t* send_off, uint32_t* recv_dev, const uint64_t* recv_off,
                             void* hip_stream);
    int32_t (*allgather_dev)(void* user, const uint32_t* send_dev, uint32_t* recv_dev, uint64_t words_per_rank, void* hip_stream);
    int32_t (*allgather_host)(void* user, const void* send, void* recv, uint64_t bytes_per_rank);
    int32_t (*allreduce_sum_u64_host)(void* user, uint64_t* buf, uint64_t n);
} lurkhip_split_comm;
/* the four collectives on an RCCL communicator of lurkhip_comm_create (ncclSend / ncclRecv pairs, ncclAllGather, ncclAllReduce on
 * the context's stream; the host variants stage through the context's pool).  `out->user` refers to ctx and comm: both must
 * outlive every key / shard made with it. */
int32_t lurkhip_comm_split_vtable(lurkhip_ctx* ctx, lurkhip_comm* comm, lurkhip_split_comm* out);
/* lurkhip_setup / lurkhip_shard_commit_pitched / lurkhip_shard_prove for one shard over comm->world ranks.  Chips of at least
 * 2^split_min_log_n rows (>= log2 world) are cut, the shorter ones are proved whole by every rank.  main_row_blocks = 0: every
 * rank passes the same whole traces (2 % of a step's work, repeated on every rank); != 0: main_traces_dev[i] of a cut chip holds
 * only this rank's block of rows [rank N / G, (rank + 1) N / G) (lurkhip_func_trace_run_rows), the other chips' their whole trace.
 * The preprocessed traces are whole on every rank.  The struct is copied.  Roots, proofs and the challenger's final state are identical on every rank and equal to the one-rank
 * entry points' (tests/test_split_gpu.py). */
int32_t lurkhip_setup_split(lurkhip_ctx* ctx, const lurkhip_split_comm* comm, int32_t split_min_log_n, int32_t n_prep,
                            const uint32_t* const* prep_traces_dev, const uint32_t* log_heights, const uint32_t* widths, int32_t log_blowup,
                            lurkhip_pk** out, uint32_t* root);
int32_t lurkhip_shard_commit_split(lurkhip_ctx* ctx, const lurkhip_split_comm* comm, int32_t split_min_log_n, int32_t n_chips,
                                   lurkhip_air* const* airs, const uint32_t* log_heights, const uint32_t* const* main_traces_dev,
                                   const uint32_t* main_pitches, const int32_t* prep_indices, int32_t log_blowup, int32_t main_row_blocks,
                                   lurkhip_shard** out, uint32_t* root);
int32_t lurkhip_shard_prove_split(lurkhip_ctx* ctx, const lurkhip_pk* pk, lurkhip_shard* shard, lurkhip_challenger* challenger,
                                  const uint32_t* public_values, uint32_t n_public, uint32_t num_queries, uint32_t pow_bits,
                                  lurkhip_proof** out);
/* out[0] / out[1] = bytes this rank has sent to other ranks in the all-to-alls before / after the LDEs, out[2] = the number of
 * all-to-alls, since the context was created or last reset (the bench's bytes-per-link figure). */
int32_t lurkhip_split_stats(lurkhip_ctx* ctx, uint64_t* out /* [3] */, int32_t reset);
/* The index arithmetic of the two exchanges alone (host only; tests/test_split_plan.py runs it on host arrays over gloo with
 * ragged widths).  Matrix i: 2^log_heights[i] x widths[i], kinds[i] = 0 every rank holds all rows, 1 rank r holds natural rows
 * [r N / G, (r + 1) N / G), 2 chunk chunks[i] of a quotient of degree 2^lqds[i] held as the quotient kernel leaves it; n_next[i]
 * next-row copies (of columns 0 ..) travel with exchange B; run_counts[i] (first column, width) pairs of `runs`, taken in matrix
 * order, are the columns of matrix i that are not identically zero (0 pairs, or run_counts NULL: all of them).  Writes the plan as
 * 64-bit words (layout: lurk_amd/split.py parse_plan) and returns their number, or a negative status; out may be NULL to size the
 * buffer. */
int64_t lurkhip_split_plan(int32_t world, int32_t rank, int32_t split_min_log_n, int32_t n_mats, const uint32_t* log_heights,
                           const uint32_t* widths, const int32_t* kinds, const uint32_t* lqds, const uint32_t* chunks, const uint32_t* n_next,
                           const uint32_t* run_counts, const uint32_t* runs, uint64_t* out, uint64_t capacity);

/* ------------------------------------------------------------------- proof wire format */
/* The reference's serialised proofs.  `CryptoProof { shard_proofs, verifier_version, depth }` with
 * `CryptoShardProof { commitment, opened_values, opening_proof, chip_ordering }` as `bincode::serialize` writes them
 * (/root/reference/src/core/cli/proofs.rs:22-35, /root/reference/src/core/cli/repl.rs:200-203), built from the flat words of
 * lurkhip_proof_read, one array per shard.  chip_names[machine index] = the chips' `name()` (lurkhip_air_name) for
 * `chip_ordering`; depth = the last four public values as little-endian bytes (proofs.rs:115-124).  The inner sphinx / Plonky3
 * types are [UPSTREAM-RECALL] (field order in lurk_amd/csrc/wire.cpp); serialize_montgomery mirrors the profile field of that
 * name.  Returns the byte count (written only when capacity suffices), negative on malformed input. */
/* One sphinx `ShardProof { commitment, opened_values, opening_proof, chip_ordering, public_values }` as bincode: the
 * CryptoShardProof of the entry point below plus the trailing `public_values: Vec<Val>` it drops
 * (/root/reference/src/core/cli/proofs.rs:61-74,94-101).  This is the byte string an upstream vector
 * (`bincode::serialize(&proof.shard_proofs[0])`, tests/golden/upstream/README.md key "shard_proof") is compared with. */
int64_t lurkhip_shard_proof_bincode(const uint32_t* words, uint64_t n_words, int32_t n_chip_names, const char* const* chip_names,
                                    int32_t serialize_montgomery, uint8_t* out, uint64_t capacity);
int64_t lurkhip_crypto_proof_bincode(int32_t n_shards, const uint32_t* const* shard_words, const uint64_t* shard_n_words, int32_t n_chip_names,
                                     const char* const* chip_names, const char* verifier_version, int32_t serialize_montgomery, uint8_t* out,
                                     uint64_t capacity);
/* `CachedProof { crypto_proof, expr, env, result, zdag }` (proofs.rs:137-143): crypto_proof = bytes of the call above, the
 * three pointers as tag + digest (9 words), zdag = entries of lurkhip_zstore_dag_export for [expr, env, result]
 * (/root/reference/src/core/cli/zdag.rs:12, a map ZPtr -> ZPtrType). */
int64_t lurkhip_cached_proof_bincode(const uint8_t* crypto_proof, uint64_t crypto_len, const uint32_t* expr, const uint32_t* env,
                                     const uint32_t* result, uint64_t n_dag_entries, const uint32_t* dag_entries, int32_t serialize_montgomery,
                                     uint8_t* out, uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* LURKHIP_H */
