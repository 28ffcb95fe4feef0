// One shard over G = 2^log_g ranks: internal interfaces (split.hip; plan: split_plan.h; public entry points: include/lurkhip.h).
#pragma once
#include <stdint.h>

#include <utility>
#include <vector>

#include "commit.h"
#include "split_plan.h"

namespace lurkhip {

struct SplitEnv {
    lurkhip_split_comm comm{};  // a copy of the caller's callbacks
    int log_g = 0, rank = 0;
    int min_log_n = 0;          // chips of at least 2^min_log_n rows are cut, the shorter ones proved whole by every rank
    bool on() const { return log_g > 0; }
    int world() const { return 1 << log_g; }
};
int32_t split_env_init(lurkhip_ctx* ctx, const lurkhip_split_comm* comm, int32_t min_log_n, SplitEnv* out);

// One matrix of a commitment-to-be (natural row order, Montgomery, device).
struct SplitMat {
    const uint32_t* src;  // K_FULL: all 2^log_n rows; K_BLOCK: this rank's N / G rows; K_QUOTIENT: this rank's 2N / G quotient values (brev(j) order)
    uint32_t log_n, width, pitch;
    uint32_t shift;       // canonical coset shift of the extension, 0: the generator
    int kind;             // split::Kind (a matrix below 2^min_log_n rows is K_FULL)
    uint32_t lqd, chunk;  // K_QUOTIENT
    uint32_t n_next, next_lqd;
    // the column runs outside which the matrix is zero on every rank (split::MatDesc::runs); null: every column
    const std::vector<std::pair<uint32_t, uint32_t>>* runs = nullptr;
};
// p3 TwoAdicFriPcs::commit of `mats` by all ranks together: *out is this rank's part (lurkhip_commitment::split_log_g), root_m the
// root every rank computes (Montgomery).
int32_t split_commit(lurkhip_ctx* ctx, const SplitEnv& env, int n, const SplitMat* mats, int log_blowup, lurkhip_commitment** out, uint32_t* root_m);

// the collectives with the library's error convention
int32_t split_allgather_host(lurkhip_ctx* ctx, const SplitEnv& env, const void* send, void* recv, uint64_t bytes_per_rank);
int32_t split_allreduce_u64_host(lurkhip_ctx* ctx, const SplitEnv& env, uint64_t* buf, uint64_t n);
int32_t split_allgather_dev(lurkhip_ctx* ctx, const SplitEnv& env, const uint32_t* send_dev, uint32_t* recv_dev, uint64_t words_per_rank);
// column[r * stride_words .. +4] += offset for r < n (extension elements, Montgomery): the running sum of a row block continues the previous ranks'
int32_t add_ef_to_column(lurkhip_ctx* ctx, uint32_t* data, size_t stride_words, size_t n, const uint32_t offset_m[4]);

}  // namespace lurkhip
