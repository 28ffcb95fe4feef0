// Internal interfaces of the AIR-driven prover stages (stark.hip).
#pragma once
#include <stdint.h>

#include <vector>

#include "babybear.h"
#include "ctx.h"
#include "lair/air.h"

struct lurkhip_air;

namespace lurkhip {

// device copies (per HIP device) of the chip's two programs
int32_t air_programs_dev(lurkhip_ctx* ctx, lurkhip_air* a, const uint32_t** constraints, const uint32_t** interactions,
                         const std::vector<uint32_t*>** interaction_parts = nullptr, bool coarse = false,
                         const uint32_t** interaction_static = nullptr, const std::vector<uint32_t*>** constraint_parts = nullptr);
const lair::ChipAir& air_of(const lurkhip_air* a);
const lair::AirPrograms& programs_of(const lurkhip_air* a);
int vm_block(uint32_t n_regs, size_t* lds_bytes);
// out[i] = base^i for i < count (extension field, Montgomery words)
int32_t ef_powers(lurkhip_ctx* ctx, const uint32_t base_m[4], uint32_t* out_dev, uint32_t count, bool centred = false, bool reversed = false);
// in-place inclusive scan of the EF elements data[r * stride_words .. +4], r < n
int32_t scan_ef_column(lurkhip_ctx* ctx, uint32_t* data, size_t stride_words, size_t n);

// Montgomery-form entry points shared by the C ABI wrappers and the shard prover.
// `shared_beta_pows` / `shared_starts`: tables of one proof kept by the caller across its chips and stages -- the centred powers of
// the permutation challenge beta (at least air_beta_pows(a) entries of 8 words; every chip reads a prefix of the same table) and
// the chip's interaction start values (air_num_interactions(a) x 4 words; written by permutation_trace_impl, read by
// quotient_impl).  nullptr: each call builds its own (two or three more launches per chip and stage).
uint32_t air_beta_pows(const lurkhip_air* a);
uint32_t air_num_interactions(const lurkhip_air* a);
int32_t permutation_trace_impl(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t height, const uint32_t* main_dev, const uint32_t* prep_dev,
                               const bb::ef& alpha, const bb::ef& beta, uint32_t* out_dev, bb::ef* cumulative_sum_m,
                               const uint32_t* shared_beta_pows = nullptr, uint32_t* shared_starts = nullptr,
                               uint32_t main_pitch = 0 /* words between rows of main_dev; 0: the chip's width */,
                               uint32_t out_pitch = 0 /* words between rows of out_dev; 0: 4 x permutation width */,
                               uint32_t* col_live = nullptr /* device, [permutation width - 1], zeroed by the caller: receives a 1 per batch column
                               some wave computed; a column left at 0 is identically zero (stark_kernels.h: PermSink) */,
                               bool starts_ready = false /* shared_starts already holds the chip's start values (interaction_starts_batch) */,
                               bool defer_scan = false /* leave the last column as the rows' sums: the caller scans it (scan_ef_columns_one_chunk) */);
// the running sums of several short columns (scan_is_one_chunk(n) each) in one launch: data[i] = the column's first element
bool scan_is_one_chunk(size_t n);
int32_t scan_ef_columns_one_chunk(lurkhip_ctx* ctx, int n_cols, uint32_t* const* data, const uint32_t* strides, const uint32_t* ns);
// the interaction start values of n chips (starts[i]: air_num_interactions(airs[i]) x 4 words) in as few launches as their number allows
int32_t interaction_starts_batch(lurkhip_ctx* ctx, int n, lurkhip_air* const* airs, const bb::ef& alpha, const uint32_t* beta_pows, uint32_t* const* starts);

uint32_t* selector_table_of(lurkhip_ctx* ctx, uint32_t log_n, uint32_t lqd);  // stark.hip: written on the current stream at first use
// One shard over several ranks: the quotient values of the n_rows storage rows from s_base on (stark_kernels.h: QuotientArgs::split).
// The three LDE pointers point at storage row s_base; out_dev receives [2^log_rows][4] words in brev(local row) order.
struct QuotientSplit {
    uint32_t s_base, n_rows, next_off, log_rows;
};
uint32_t air_next_columns(const lurkhip_air* a);  // 1 + the highest main column the chip's constraints read on the next row (0: none)
bool air_reads_prep_next(const lurkhip_air* a);
int32_t quotient_impl(lurkhip_ctx* ctx, lurkhip_air* a, uint32_t log_n, const uint32_t* main_lde_dev, const uint32_t* prep_lde_dev,
                      const uint32_t* perm_lde_dev, const bb::ef& perm_alpha, const bb::ef& perm_beta, const bb::ef& alpha_m,
                      const bb::ef& cumsum_m, const uint32_t* public_values, uint32_t* out_dev,
                      const uint32_t* shared_beta_pows = nullptr, const uint32_t* shared_starts = nullptr,
                      const uint32_t* pitches = nullptr /* {main, prep, perm} row pitches in words, 0 or null: the matrix's width */,
                      bool honest_running_sum = false /* the permutation LDE is the LDE of a trace this prover built: its last column IS the
                      running sum of the row sums and ends in cumsum_m, so the next row's sum need not be read (stark_kernels.h) */,
                      const uint32_t* shared_alpha_pows = nullptr /* centred powers alpha^j, j < at least the chip's constraint count, natural
                      order (ef_powers(.., centred = true, reversed = false)): one table for all chips of a proof; null: the call builds its own */,
                      const uint32_t* shared_public_m = nullptr /* the public values on the device, Montgomery; null: uploaded by the call */,
                      const uint32_t* cumsum_dev = nullptr /* the chip's cumulative sum on the device (4 words, Montgomery): read by the kernel
                      instead of cumsum_m, which the host may not know yet (with shared_alpha_pows only) */,
                      const QuotientSplit* split = nullptr /* this rank's rows only (honest_running_sum required) */,
                      const uint32_t* col_live = nullptr /* device, one word per batch column as permutation_trace_impl left them (0: nobody computed
                      the column: identically zero); the interaction waves step over those columns without reading their entries */);
int32_t ef_powers_dev(lurkhip_ctx* ctx, const uint32_t* base_dev /* 4 words, device */, uint32_t* out_dev, uint32_t count, bool centred, bool reversed = false);
uint32_t air_total_constraints(const lurkhip_air* a);  // constraints + batch columns + the three running-sum constraints

}  // namespace lurkhip

extern "C" int32_t lurkhip_air_from_chip(lair::ChipAir&& air, lurkhip_air** out);
