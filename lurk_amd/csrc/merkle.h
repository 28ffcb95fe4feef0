// What the Merkle hashing kernels are made of (merkle.hip): the width-16 Poseidon2 parameters, the leaf-column descriptor and the
// launch interfaces.  Split out of commit.h in round 6 so that the files whose code the hashing kernels contain -- merkle.hip,
// merkle.h, poseidon2_dev.h, p16_coop.h, babybear.h: the list bench.py hashes before it prices a run with the committed instruction
// count -- do not change when the commit pipeline's host-side declarations do.
#pragma once
#include <stdint.h>

#include "ctx.h"

namespace lurkhip {

// ---- Poseidon2 width-16 parameters for the Merkle hash (device resident, Montgomery) ----
constexpr int P16_MAX_RP = 32;
struct P16Params {
    uint32_t ext_rc[8 * 16];
    uint32_t int_rc[P16_MAX_RP];
    uint32_t diag[16];           // Montgomery form of scale * diag_i: the internal layer is y_i = diag[i] x_i + sum_mult * sum_j x_j
    int32_t rounds_p;
    uint32_t ext_rc_mp[8 * 16];  // rc - p (mod 2^32), for the signed S-box chain
    uint32_t int_rc_mp[P16_MAX_RP];
    int32_t diag_c[16];          // diag, centred in (-p/2, p/2]: multiplier of the lazy internal rounds
    uint32_t sum_mult;           // Montgomery form of the internal layer's scale (lurkhip_protocol_profile::p16_internal_scale): R mod p for scale 1
    int32_t sum_mult_c;          // the same, centred: integer multiplier of the lazily reduced lane sum
    void finish() {
        for (int i = 0; i < 128; i++) ext_rc_mp[i] = ext_rc[i] - 2013265921u;
        for (int i = 0; i < P16_MAX_RP; i++) int_rc_mp[i] = int_rc[i] - 2013265921u;
        for (int i = 0; i < 16; i++) diag_c[i] = diag[i] > 2013265921u / 2 ? (int32_t)(diag[i] - 2013265921u) : (int32_t)diag[i];
        sum_mult_c = sum_mult > 2013265921u / 2 ? (int32_t)(sum_mult - 2013265921u) : (int32_t)sum_mult;
    }
};

// One matrix column of the concatenated leaf row (uniform descriptor, read through the scalar cache)
struct LeafCol {
    const uint32_t* base;
    uint32_t width;
    uint32_t col;
};

// digests[level] has (n_leaves >> level) entries of 8 words; stored back to back
int32_t merkle_leaves(lurkhip_ctx* ctx, const P16Params* params_dev, const LeafCol* cols_dev, uint32_t total_w,
                      size_t n_rows, uint32_t* digests_out);
// parents[i] = compress(children[2i], children[2i+1]); if inject_cols: then compress(that, hash(row i))
int32_t merkle_level(lurkhip_ctx* ctx, const P16Params* params_dev, const uint32_t* children, size_t n_parents,
                     const LeafCol* inject_cols_dev, uint32_t inject_w, uint32_t* parents);
// The row sponges of several height groups of one tree in ONE launch (a group = the matrices of one height, concatenated:
// the leaves, or the rows injected at a level), longest rows first: a level kernel that hashes its injected rows itself runs
// 40 permutations per lane for the 2^20-row group of a fib shard, sixteen such waves per SIMD at five or six resident -- the
// last ones alone on the chip; hashed ahead of the levels, all groups share one grid whose tail is made of the shortest rows.
// out[g] receives n_rows[g] digests of 8 words.
constexpr int SPONGE_MAX_GROUPS = 16;
constexpr size_t MERKLE_COOP_MAX_PARENTS = 16384;  // levels of at most this many parents run lane-cooperatively (merkle.hip)
struct SpongeGroups {
    int n = 0;
    const LeafCol* cols[SPONGE_MAX_GROUPS];
    uint32_t total_w[SPONGE_MAX_GROUPS];
    uint64_t n_rows[SPONGE_MAX_GROUPS];
    uint32_t* out[SPONGE_MAX_GROUPS];
    uint32_t first_block[SPONGE_MAX_GROUPS + 1];  // filled by merkle_row_sponges
    uint8_t coop[SPONGE_MAX_GROUPS];              // filled by merkle_row_sponges: the group's rows are hashed by 16 lanes each
};
int32_t merkle_row_sponges(lurkhip_ctx* ctx, const P16Params* params_dev, SpongeGroups groups);
// parents[i] = compress(children[2i], children[2i+1]); if inject_digests: then compress(that, inject_digests[i])
int32_t merkle_level_digests(lurkhip_ctx* ctx, const P16Params* params_dev, const uint32_t* children, size_t n_parents,
                             const uint32_t* inject_digests, uint32_t* parents);
// collapses the levels below `n` nodes down to the root inside one workgroup (n <= 2048)
// matrices injected at the levels merkle_top collapses: entry t describes the rows absorbed into the parents of step t
// (n >> (t + 1) of them); cols[t] == nullptr where nothing is injected
struct TopInject {
    const LeafCol* cols[11];
    uint32_t w[11];
    const uint32_t* dig[11];  // the rows' sponge digests when they were hashed ahead (merkle_row_sponges): 8 words per row; then cols[t] is not used
};
int32_t merkle_top(lurkhip_ctx* ctx, const P16Params* params_dev, uint32_t* level_base, size_t n, const TopInject& inject);
// `levels` (1 .. 5) consecutive cooperative levels above the n_children nodes at `children` in one launch; the levels' digests
// lie back to back behind them (inject as for merkle_top: entry t = the rows absorbed into the parents of step t)
int32_t merkle_levels_coop(lurkhip_ctx* ctx, const P16Params* params_dev, uint32_t* children, size_t n_children, int levels,
                           const TopInject& inject);

}  // namespace lurkhip
