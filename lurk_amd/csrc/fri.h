// Internal interfaces of the opening-phase kernels (fri.hip).
#pragma once
#include <stdint.h>

#include <vector>

#include "babybear.h"
#include "ctx.h"

namespace lurkhip {

// out[s] (EF, 4 Montgomery words), s < 2^log_m, for the point x_s = 31 * w_M^bitrev(s):
//   mode 0: w_M^bitrev(s) / (z - x_s)   (barycentric weights of the coset 31 * <w_M>)
//   mode 1: 1 / (x_s - z)
int32_t point_weights(lurkhip_ctx* ctx, int mode, int log_m, const bb::ef& z, uint32_t* out_dev);
// several tables in one launch (mode 0 tables centred, as point_weights writes them)
constexpr int PW_BATCH_MAX = 48;
struct WeightJob {
    int mode, log_m;
    bb::ef z;
    uint32_t* out;
};
int32_t point_weights_batch(lurkhip_ctx* ctx, const std::vector<WeightJob>& jobs);
// Opened values.  partial[blk][p][c] = sum over the block's rows of mat[s][c] * u_p[s] (p = 0, and 1 when u1 != null) for
// one matrix; column_dot_finish then sums the blocks of every matrix of the proof in one launch:
// out_dev[out_off + (p * w + c) * 4 ..] = sum_{s < n_rows} mat[s][c] * u_p[s].
size_t column_dot_partial_words(uint32_t w, size_t n_rows);
int32_t column_dot_partial(lurkhip_ctx* ctx, const uint32_t* mat, uint32_t w, uint32_t pitch /* words between rows */, size_t n_rows, const uint32_t* u0, const uint32_t* u1,
                           uint32_t* partial_dev);
// the narrow matrices (column_dot_is_narrow: the ones column_dot_partial gives to the slab kernel) of an opening in one launch
constexpr int NARROW_DOT_MAX = 64;
struct NarrowDot {
    const uint32_t* mat;
    uint32_t w;
    size_t n_rows;
    const uint32_t *u0, *u1;
    uint32_t* partial;
    uint32_t pitch = 0;  // words between rows; 0: w
};
bool column_dot_is_narrow(uint32_t w);
int32_t column_dot_partial_batch(lurkhip_ctx* ctx, const std::vector<NarrowDot>& items);
struct DotJob {
    const uint32_t* partial;
    uint32_t w;
    size_t n_rows;
    uint32_t out_off;  // word offset of the matrix's [2][w][4] block in out_dev
    bool two_points;
};
constexpr uint32_t DOT_FINISH_MAX = 64;
struct DotFinishArgs {
    uint32_t n;
    uint32_t col_start[DOT_FINISH_MAX + 1];
    const uint32_t* partial[DOT_FINISH_MAX];
    uint32_t w[DOT_FINISH_MAX], n_blocks[DOT_FINISH_MAX], out_off[DOT_FINISH_MAX];
    uint32_t* out;
};
int32_t column_dot_finish(lurkhip_ctx* ctx, const std::vector<DotJob>& jobs, uint32_t* out_dev);
// reduced openings of the narrow matrices of one height in one launch (fri.hip: k_reduce_openings_narrow)
constexpr uint32_t NARROW_MAX_W = 16, NARROW_MAX_MATS = 16;
struct NarrowMat {
    const uint32_t* mat;
    uint32_t w;
    uint32_t two;  // opened at both points (else only at the first)
    bb::ef ys0, ys1, apow0, apow1;
    uint32_t pitch = 0;  // words between rows; 0: w
};
struct NarrowArgs {
    NarrowMat m[NARROW_MAX_MATS];
    uint32_t n_mats;
    uint32_t m_rows;
    const uint32_t* alpha_pows;  // centred table (k_ef_powers)
    const uint32_t* d0;
    const uint32_t* d1;  // nullable: no matrix of the group is opened at a second point
    uint32_t* ro;
};
int32_t reduce_openings_narrow(lurkhip_ctx* ctx, const NarrowArgs& args);
// Reduced openings of the matrices of one height, any width from 4 columns up, any row pitch, in one launch (fri.hip:
// k_reduce_openings_rows, round 4): four lanes to a row, every lane reads 16-byte pieces of its row straight from memory.
constexpr uint32_t ROWS_MAX_MATS = 12, ROWS_MAX_W = 2048;
struct RowsMat {
    const uint32_t* mat;  // first column of row 0
    uint32_t pitch;       // words between rows
    uint32_t w;           // >= 4
    uint32_t two;         // opened at both points (else only at the first)
    bb::ef ys0, ys1, apow0, apow1;
};
struct RowsArgs {
    RowsMat m[ROWS_MAX_MATS];
    uint32_t n_mats;
    uint32_t m_rows;
    uint32_t max_w;
    const uint32_t* alpha_pows;  // centred table (k_ef_powers), at least max_w entries
    const uint32_t* d0;
    const uint32_t* d1;  // nullable: no matrix of the group is opened at a second point
    uint32_t* ro;
};
int32_t reduce_openings_rows(lurkhip_ctx* ctx, RowsArgs args);
// reduced openings of one wide matrix (w > 128) as column slices of at most 128 words in one launch (fri.hip:
// k_reduce_openings_wide).  Slice i covers columns c0[i] .. c0[i] + sw[i]; ys_p[i] = sum_j alpha^j y_p[c0[i] + j] and
// apow_p[i] = (the matrix's alpha offset at point p) * alpha^c0[i].
constexpr uint32_t WIDE_MAX_SLICES = 24;  // (round 4: a padded height group is reduced as slices of one wide matrix)
struct WideArgs {
    const uint32_t* mat;
    uint32_t w;       // row pitch in words
    uint32_t m_rows;
    const uint32_t* alpha_pows;  // centred table (k_ef_powers), at least max(sw) entries
    const uint32_t* d0;
    const uint32_t* d1;  // nullable
    uint32_t* ro;
    uint32_t n_slices;
    uint32_t sw[WIDE_MAX_SLICES], c0[WIDE_MAX_SLICES], magic[WIDE_MAX_SLICES];  // magic: filled by reduce_openings_wide
    bb::ef ys0[WIDE_MAX_SLICES], ys1[WIDE_MAX_SLICES], apow0[WIDE_MAX_SLICES], apow1[WIDE_MAX_SLICES];
};
int32_t reduce_openings_wide(lurkhip_ctx* ctx, WideArgs args);
// ro[s] += apow0 * (rr_s - ys0) * d0[s] (+ apow1 * (rr_s - ys1) * d1[s]),  rr_s = sum_c alpha_pows[c] * mat[s][c]
int32_t reduce_openings(lurkhip_ctx* ctx, const uint32_t* mat, uint32_t w, uint32_t m_rows, const uint32_t* alpha_pows,
                        const uint32_t* alpha_pows_centred /* 8 words per power (ef_powers centred), or null */, const uint32_t* d0, const uint32_t* d1, const bb::ef& ys0, const bb::ef& ys1, const bb::ef& apow0,
                        const bb::ef& apow1, uint32_t* ro);
// p3 fold_even_odd on 2^log_len bit-reversed evaluations (+ add[j] when given); out has 2^(log_len-1) elements
int32_t fri_fold(lurkhip_ctx* ctx, const uint32_t* cur, int log_len, const uint32_t* beta_dev /* 4 words, device */, const uint32_t* add,
                 uint32_t* out, uint32_t pair_base = 0 /* a block of the layer: cur / add / out start at pair pair_base .. */,
                 uint32_t n_pairs = 0 /* .. and hold n_pairs pairs (0: the whole layer) */);
// Transcript state as the device keeps it during the FRI commit phase (challenger.h: Challenger, same semantics)
struct DevChallenger {
    uint32_t state[16];
    uint32_t input[8];
    uint32_t output[16];  // the not yet sampled outputs are output[out_head .. out_head + n_out)
    uint32_t n_in, n_out, out_head;
    uint32_t squeeze;     // lanes offered after a permutation: 8 or 16 (lurkhip_protocol_profile::challenger_squeeze)
    uint32_t pop_front;   // sample from the front instead of the end
};
// observes the 8-word digest at root_dev and samples one extension element into beta_dev, all on the context's stream
int32_t fri_challenge(lurkhip_ctx* ctx, DevChallenger* ch_dev, const uint32_t* root_dev, uint32_t* beta_dev, uint32_t* root_copy_dev);
// smallest canonical witness w such that a challenger whose permutation input is `state` (pending inputs already
// written over lanes [0, n_pending)) samples `bits` zero bits after observing w; sample_lane = the lane of the permuted state
// that the first sample() returns (challenger.h: Challenger::first_sample_lane)
int32_t pow_grind(lurkhip_ctx* ctx, const uint32_t state_with_pending_m[16], int n_pending, int bits, int sample_lane, uint32_t* witness);

struct OpenMat {
    const uint32_t* base;
    uint32_t width;
    uint32_t log_h;
    uint32_t pitch = 0;  // row pitch in words; 0: the width (lurkhip_commitment::pitch)
};
// MMCS open_batch for n_queries indices at once: record q = [row of every matrix at (indices[q] >> shift) >>
// (log_max - log_h) | log_max sibling digests].  out_dev == null only computes record_words.
int32_t gather_openings(lurkhip_ctx* ctx, const std::vector<OpenMat>& mats, const uint32_t* digests, const std::vector<size_t>& level_off,
                        uint32_t log_max, const uint32_t* indices_dev, uint32_t n_queries, uint32_t shift, uint32_t* out_dev,
                        uint32_t* record_words);

}  // namespace lurkhip
