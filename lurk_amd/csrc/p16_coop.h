// Lane-cooperative width-16 Poseidon2 (16 lanes hold one state): shared by the Merkle tree tails (merkle.hip) and the
// device-side transcript of the FRI commit phase (fri.hip).
#pragma once
#include "babybear.h"
#include "commit.h"

namespace lurkhip {

// ---- lane-cooperative permutation: 16 lanes hold one width-16 state (lane j = element j).
// The tail of a tree is a latency chain (one permutation per level with nothing else to run): spreading a
// permutation over 16 lanes cuts its dependent-instruction count from ~5 k to ~1 k.  Data moves with DPP inside
// the 16-lane rows of the wave (quad permutes for M4, row rotations for the column / full sums).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
constexpr int DPP_QUAD_ROT1 = 0x39, DPP_QUAD_ROT2 = 0x4E, DPP_QUAD_ROT3 = 0x93;  // lane k reads lane (k + n) & 3 of its quad
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

__device__ __forceinline__ uint32_t coop_external_layer(uint32_t a) {
    // M4 = circ(2, 3, 1, 1) on each quad: y_k = 2 x_k + 3 x_{k+1} + x_{k+2} + x_{k+3}
    const uint32_t b = dpp<DPP_QUAD_ROT1>(a), c = dpp<DPP_QUAD_ROT2>(a), d = dpp<DPP_QUAD_ROT3>(a);
    const uint32_t t = bb::add(a, b), u = bb::add(c, d);
    uint32_t y = bb::add(bb::add(bb::add(t, u), t), b);
    // plus the sum of the same position over the four quads
    uint32_t v = bb::add(y, dpp<DPP_ROW_ROR4>(y));
    v = bb::add(v, dpp<DPP_ROW_ROR8>(v));
    return bb::add(y, v);
}

__device__ __forceinline__ uint32_t coop_perm16(uint32_t x, const P16Params* __restrict__ p, int j) {
    const uint32_t diag = p->diag[j];
    x = coop_external_layer(x);
#pragma unroll 1
    for (int r = 0; r < 4; r++) x = coop_external_layer(bb::add_pow7_mp(x, p->ext_rc_mp[r * 16 + j]));
#pragma unroll 1
    for (int r = 0; r < p->rounds_p; r++) {
        const uint32_t sb = bb::add_pow7_mp(x, p->int_rc_mp[r]);
        x = j == 0 ? sb : x;
        uint32_t sum = bb::add(x, dpp<DPP_ROW_ROR8>(x));
        sum = bb::add(sum, dpp<DPP_ROW_ROR4>(sum));
        sum = bb::add(sum, dpp<DPP_ROW_ROR2>(sum));
        sum = bb::add(sum, dpp<DPP_ROW_ROR1>(sum));
        x = bb::add(bb::mul(x, diag), sum);
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) x = coop_external_layer(bb::add_pow7_mp(x, p->ext_rc_mp[r * 16 + j]));
    return x;
}

}  // namespace lurkhip
