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
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128, DPP_ROW_ROR12 = 0x12C;

// The layers run on lazily reduced signed lanes (babybear.h: sred), like the one-permutation-per-lane kernels: a lane is an
// int32 x, |x| < p, standing for x mod p.  With R1 = 2^32 mod p = 0.133 p every linear combination is a chain of 64-bit
// multiply-adds by small multiples of R1 followed by one signed reduction, which gives the plain sum back:
//   M4 row     y = sred(R1 (2a + 3b + c + d)),        |.| <= 7 R1 |a| < 0.97 p^2 for |a| <= 1.03 p,  |y| < 0.95 p
//   quad sums  o = sred(R1 (2y + y4 + y8 + y12)),     |.| <= 5 R1 |y| < 0.64 p^2,                    |o| < 0.80 p
// (sred needs |t| < 1.2 p^2 and returns |r| <= |t| / 2^32 + p / 2.)
__device__ __forceinline__ int32_t coop_external_layer(int32_t a) {
    constexpr int32_t R1 = (int32_t)bb::R1;
    // M4 = circ(2, 3, 1, 1) on each quad: y_k = 2 x_k + 3 x_{k+1} + x_{k+2} + x_{k+3}
    const int32_t b = (int32_t)dpp<DPP_QUAD_ROT1>((uint32_t)a), c = (int32_t)dpp<DPP_QUAD_ROT2>((uint32_t)a),
                  d = (int32_t)dpp<DPP_QUAD_ROT3>((uint32_t)a);
    const int32_t y = bb::sred(bb::mad_i64_u(a, 2 * R1, bb::mad_i64_u(b, 3 * R1, bb::mad_i64_u(c, R1, bb::mad_i64_u(d, R1, 0)))));
    // plus the sum of the same position over the four quads (its own quad counted twice in all)
    const int32_t y4 = (int32_t)dpp<DPP_ROW_ROR4>((uint32_t)y), y8 = (int32_t)dpp<DPP_ROW_ROR8>((uint32_t)y),
                  y12 = (int32_t)dpp<DPP_ROW_ROR12>((uint32_t)y);
    return bb::sred(bb::mad_i64_u(y, 2 * R1, bb::mad_i64_u(y4, R1, bb::mad_i64_u(y8, R1, bb::mad_i64_u(y12, R1, 0)))));
}

// (x + rc)^7 of a lazy lane: brought to [0, p) first so that x + (rc - p) lies in (-p, p); the chain stays signed, |x^7| < 0.94 p
__device__ __forceinline__ int32_t coop_sbox(int32_t x, uint32_t rc_mp) {
    const uint32_t c = bb::umin((uint32_t)x, (uint32_t)x + bb::P);
    const int32_t y = (int32_t)(c + rc_mp);
    const int32_t y2 = bb::smul(y, y), y3 = bb::smul(y2, y), y6 = bb::smul(y3, y3);
    return bb::smul(y6, y);
}

__device__ __forceinline__ uint32_t coop_perm16(uint32_t x_in, const P16Params* __restrict__ p, int j) {
    const int32_t sum_mult = p->sum_mult_c;  // R mod p for the plain layer (commit.h: P16Params)
    const int32_t diag = p->diag_c[j];
    int32_t x = coop_external_layer((int32_t)x_in);
#pragma unroll 1
    for (int r = 0; r < 4; r++) x = coop_external_layer(coop_sbox(x, p->ext_rc_mp[r * 16 + j]));
    // (Measured and rejected in round 2: every lane carrying a copy of element 0 and running its S-box itself, so that the lane
    // sum of the other fifteen starts beside the S-box instead of behind it -- 25 dependent instructions per round instead of 37,
    // three more in all -- made the tree tops 11 % slower: a lone wave is bound by the instructions it issues, not by their
    // dependences.)
#pragma unroll 1
    for (int r = 0; r < p->rounds_p; r++) {
        const int32_t sb = coop_sbox(x, p->int_rc_mp[r]);
        x = j == 0 ? sb : x;
        // the sum of the lanes on canonical values (four modular adds), then x <- sred(x d + m sum), m = sum_mult (R mod p for
        // the plain layer): |x d + m sum| < 0.94 p * p / 2 + p / 2 * p = 0.97 p^2 < 1.2 p^2 for any centred m, |x| < 0.99 p
        const uint32_t c = bb::umin((uint32_t)x, (uint32_t)x + bb::P);
        uint32_t sum = bb::add(c, dpp<DPP_ROW_ROR8>(c));
        sum = bb::add(sum, dpp<DPP_ROW_ROR4>(sum));
        sum = bb::add(sum, dpp<DPP_ROW_ROR2>(sum));
        sum = bb::add(sum, dpp<DPP_ROW_ROR1>(sum));
        x = bb::sred(bb::mad_i64(x, diag, bb::mad_i64_u((int32_t)sum, sum_mult, 0)));
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) x = coop_external_layer(coop_sbox(x, p->ext_rc_mp[r * 16 + j]));
    return bb::umin((uint32_t)x, (uint32_t)x + bb::P);
}

}  // namespace lurkhip
