// Internal interfaces of the commit pipeline (coset LDE + Poseidon2-16 Merkle tree).
#pragma once
#include <stdint.h>

#include <utility>
#include <vector>

#include "ctx.h"
#include "merkle.h"

// Device-resident result of a commit: the LDE matrices and every level of the Merkle tree.
struct lurkhip_commitment {
    int n_mats = 0;
    int log_blowup = 0;
    std::vector<uint32_t*> lde;       // device, Montgomery, (1 << log_h[i]) x width[i], bit-reversed rows
    bool owns_lde = true;
    std::vector<int> log_h;           // log2 of the LDE height
    std::vector<uint32_t> width;
    // Row pitch of lde[i] in words.  Equal to width[i] for a matrix with its own buffer; the matrices of one height that went through
    // the grouped LDE together (commit_impl with padded_groups) are column ranges of ONE buffer [2N][pitch], pitch = the group's
    // total width rounded up to 32 words: every row, and every 32-column tile of the LDE's last pass, starts on a 128-byte line.
    std::vector<uint32_t> pitch;
    std::vector<char> lde_is_view;    // lde[i] points into a group buffer (owned through `owned`): not released on its own
    std::vector<int> group;           // index of the matrix's group buffer, -1: its own buffer
    std::vector<uint32_t> col_start;  // first column of the matrix inside its group buffer
    std::vector<uint32_t*> group_base;
    std::vector<uint32_t*> coeffs;    // device, Montgomery, natural-order coefficients (N x w), may be null
    uint32_t* digests = nullptr;      // all levels back to back, level 0 first
    std::vector<size_t> level_off;    // in digests (units of 8 words)
    int log_max = 0;
    std::vector<void*> owned;         // extra device allocations (column tables)
    // The row sponge of ONE height group launched ahead on the context's hash stream, under the LDE passes of the other groups
    // (commit.hip: early_sponge): early_level = the tree level its rows are injected at (0: the leaves), -1: none; for a level
    // above the leaves early_digests holds the group's row digests (pooled, released with the tree's scratch).
    int early_level = -1;
    uint32_t* early_digests = nullptr;
    std::vector<void*> early_scratch;
    // ---- a commitment made by G = 2^split_log_g ranks together (split.hip; 0: an ordinary commitment).  log_h / log_max stay the
    // GLOBAL heights.  A matrix taller than G rows is "local": lde[i] holds only this rank's 2^(log_h[i] - split_log_g) storage rows,
    // starting at global row split_rank << (log_h[i] - split_log_g); digests / level_off describe the LOCAL subtree over them
    // (log_max - split_log_g levels).  The other matrices ("tiny", at most G rows) are whole on every rank and enter the tree in its
    // top split_log_g levels, which every rank computes on the host from the all-gathered subtree roots.
    int split_log_g = 0, split_rank = 0;
    std::vector<uint32_t*> full_lde;                  // per matrix: the whole LDE when this rank has it (small chips: computed by every rank), else null
    std::vector<uint32_t> next_off;                   // per matrix: words from a local row's first column to its next-row copies (0: none)
    std::vector<std::vector<uint32_t>> top_levels_m;  // host, Montgomery: top_levels_m[t] = the G >> t final node digests t levels above the subtree roots
    std::vector<std::vector<uint32_t>> tiny_rows_m;   // host, Montgomery: the tiny matrices' LDE rows (dense), empty for the others
    lurkhip_commitment* aux = nullptr;                // owns the LDE buffers of the small chips' matrices
    bool is_local(int m) const { return split_log_g > 0 && log_h[m] > split_log_g; }
    int rows_log(int m) const { return is_local(m) ? log_h[m] - split_log_g : log_h[m]; }  // log2 of the rows at lde[m]
    size_t row_base(int m) const { return is_local(m) ? (size_t)split_rank << rows_log(m) : 0; }  // global storage row of lde[m]'s first row
};

namespace lurkhip {

// ---- NTT ---------------------------------------------------------------------
struct NttPlan {
    int log_n = 0;
    uint32_t* tw_fwd = nullptr;  // w_N^i, i < N/2, Montgomery
    uint32_t* tw_inv = nullptr;  // w_N^-i
    int32_t init(lurkhip_ctx* ctx, int log_n);
    void destroy();
};

uint32_t two_adic_generator_monty(int bits);
int32_t fill_powers(lurkhip_ctx* ctx, uint32_t* out, uint32_t root_m, uint32_t scale_m, size_t count);
int32_t ntt_dif(lurkhip_ctx* ctx, const NttPlan& plan, bool inverse, const uint32_t* src, uint32_t* dst,
                uint32_t* scratch, int w, const uint32_t* row_scale, bool in_canonical, bool out_canonical,
                bool bitrev_store);
// several matrices of one shape (height, width), one launch per pass
constexpr int NTT_MAX_BATCH = 8;
struct NttBatch {
    int n;
    const uint32_t* src[NTT_MAX_BATCH];
    uint32_t* dst[NTT_MAX_BATCH];
    uint32_t* scratch[NTT_MAX_BATCH];
    const uint32_t* row_scale[NTT_MAX_BATCH];
    // Chunk-tiled layouts (ntt.hip: PassArgs::in_tiled): src / dst hold the matrix as [column chunk][N][32 words] instead of
    // row-major; scratch_tiled: scratch[m] holds ntt_tiled_words(log_n, w) words and the matrix between two passes goes there
    // tiled.  Only for shapes with ntt_tiled_words != 0.
    bool src_tiled = false, dst_tiled = false, scratch_tiled = false;
    // For the fused LDE pass (ntt_lde_fused): reversed_schedule cuts the row bits into the same pass sizes, smallest first (so
    // that the LAST pass of an inverse transform is as tall as the FIRST of the forward one); skip_last_pass stops before the last
    // pass (its output goes to dst, in bit-reversed order so far); skip_first_pass starts at the second pass, src holding the
    // first pass's output.
    bool reversed_schedule = false, skip_last_pass = false, skip_first_pass = false;
};
// The last pass of the inverse transform of b.n matrices (inter[m]: output of its earlier passes), the coset scalings and the
// first pass of each of the two forward transforms in one launch per column part (ntt.hip: k_ntt_fused); out[m][q] receives the
// first-pass output of coset q.  *done = false when the shape is not eligible (nothing launched).
int32_t ntt_lde_fused(lurkhip_ctx* ctx, const NttPlan& plan, int n, const uint32_t* const* inter, uint32_t* const (*out)[2],
                      const uint32_t* const (*scale)[2], int w, bool* done);
bool ntt_lde_fused_eligible(int log_n, int w);
// words of the chunk-tiled form of a 2^log_n x w matrix under the NTT's column-chunk plan, or 0 when the shape is not tiled
// (a single chunk, odd widths, chunks wider than a 128-byte line, one-pass transforms)
size_t ntt_tiled_words(int log_n, int w);
int32_t ntt_dif_batch(lurkhip_ctx* ctx, const NttPlan& plan, bool inverse, const NttBatch& b, int w, bool in_canonical,
                      bool out_canonical, bool bitrev_store);

int32_t get_merkle_params(lurkhip_ctx* ctx, const P16Params** out_dev);
// the context's protocol profile (created with the "default" preset on first use)
const lurkhip_protocol_profile& profile_of(lurkhip_ctx* ctx);

using ColumnRuns = std::vector<std::pair<uint32_t, uint32_t>>;
// ---- commit pipeline entry points shared with the prover (commit.hip)
int32_t commit_impl(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats, bool mats_on_host,
                    const uint32_t* log_heights, const uint32_t* widths, int32_t log_blowup, int32_t repr,
                    int32_t keep_coeffs, lurkhip_commitment** out, uint32_t* root, const uint32_t* shifts = nullptr,
                    bool raw = false /* the matrices as given: no interpolation, no coset extension (lurkhip_mmcs_commit) */,
                    bool padded_groups = false /* the prover's own commitments: height groups in one aligned-pitch buffer (lurkhip_commitment::pitch) */,
                    const uint32_t* src_pitches = nullptr /* words between rows of mats[i] (device matrices only; null: widths[i]) */,
                    const std::vector<ColumnRuns>* live_runs = nullptr /* per matrix: the ascending, disjoint (first column, width) runs outside
                    which the matrix is identically zero -- those columns' extension is zero-filled instead of computed */,
                    bool lde_only = false /* stop after the extension: no tree, no root (split.hip hashes row blocks it receives from other ranks) */);
// the Merkle tree over c's matrices as they are (build_tree: every level into c->digests / level_off)
int32_t commitment_build_tree(lurkhip_ctx* ctx, lurkhip_commitment* c);
// Row pitches for the matrices of a commitment-to-be (the prover's own traces): the matrices of one height that the grouped LDE
// takes become column ranges of ONE buffer [N][pitch], pitch = the group's width rounded up to a 128-byte line, when that costs at
// most half more memory (transient scratch whose padding is never read); otherwise a matrix keeps its own dense buffer.
// group[i] = index of the matrix's buffer (0 .. *n_groups - 1), col_start[i] = its first column there, pitch[i] = the buffer's pitch.
void plan_source_groups(int n, const uint32_t* log_heights, const uint32_t* widths, uint32_t* pitch, uint32_t* col_start, int32_t* group,
                        int32_t* n_groups);
int32_t nonzero_column_runs(lurkhip_ctx* ctx, int32_t n_mats, const uint32_t* const* mats_dev, const uint32_t* log_heights, const uint32_t* widths,
                            const uint32_t* pitches /* null: dense */, std::vector<ColumnRuns>* runs, uint32_t* zero_columns);
int32_t commit_raw(lurkhip_ctx* ctx, const std::vector<uint32_t*>& mats, const std::vector<int>& log_heights,
                   const std::vector<uint32_t>& widths, lurkhip_commitment** out);
int32_t commitment_root_m(lurkhip_ctx* ctx, const lurkhip_commitment* c, uint32_t* root_m);
void free_commitment(lurkhip_ctx* ctx, lurkhip_commitment* c);
int32_t get_ntt_plan(lurkhip_ctx* ctx, int log_n, const NttPlan** out);

}  // namespace lurkhip
