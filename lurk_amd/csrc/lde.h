// Coset LDE of a height group of device-resident matrices (lde.hip).
#pragma once
#include <stdint.h>

#include "ctx.h"

namespace lurkhip {

constexpr int LDE_MAX_MATS = 32;    // matrices (or column runs of matrices) of one launch group (one virtual row)
constexpr int LDE_MAX_CLASSES = 2;  // distinct coset shifts inside a group (the quotient chunks of one height)
constexpr int LDE_GROUP_MIN_LOG_N = 5, LDE_GROUP_MAX_LOG_N = 20;

// heights lde_group takes (LURKHIP_LDE_V2=0 turns the route off: every matrix goes through ntt.hip's transforms)
bool lde_group_takes(int log_n);
// Blow-up 2: ldes[m] (2N x widths[m]) <- evaluations of matrix m's column polynomials on shift_m * <w_2N>, rows in bit-reversed
// order (block q = coset q).  cls[m] < n_cls is the matrix's shift class, scales[q][cls] the table (shift_cls * w_2N^q)^k / N,
// k < N (commit.hip: cached_scale_table).  in_canonical / out_canonical: the caller's words are canonical (converted on the first load /
// last store); between the kernels everything is Montgomery.  Everything on ctx->stream; scratch comes from the context's pool.
bool lde_group_enabled();
int32_t lde_group(lurkhip_ctx* ctx, int log_n, int n_mats, const uint32_t* const* evals, const uint32_t* widths, uint32_t* const* ldes,
                  const uint32_t* cls, int n_cls, const uint32_t* const (*scales)[LDE_MAX_CLASSES], bool in_canonical, bool out_canonical,
                  const uint32_t* lde_pitches = nullptr,   // row pitch of ldes[m] in words (null: widths[m]): column ranges of one padded buffer
                  const uint32_t* src_pitches = nullptr,   // row pitch of evals[m] in words (null: widths[m])
                  // Dead columns (round 5): entry m's columns are columns out_starts[m] .. of an output row of out_width columns; the
                  // columns no entry covers are identically zero in the caller's matrices and written as zeros by the last pass, at
                  // ldes[m] + (column - out_starts[m]) of the last entry that starts at or before them (an entry of width 0 marks a
                  // matrix whose first columns are dead).  Null: the entries ARE the output row.
                  const uint32_t* out_starts = nullptr, uint32_t out_width = 0);

}  // namespace lurkhip
