// Micro-benchmark (GPU box only): the width-16 Poseidon2 permutation of the Merkle kernels on registers only -- NPERM
// dependent permutations per lane, no memory traffic except the round constants -- in SIMD cycles per wave-permutation
// (2.4 GHz, 1024 SIMDs), next to the sum of its parts from tools/ubench_issue.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ilurk_amd/csrc -Iinclude tools/ubench_perm.hip -o /tmp/ubench_perm
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "commit.h"
#include "poseidon2_dev.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using lurkhip::P16Params;

#ifndef VARIANT
#define VARIANT 0
#endif

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_perm(const P16Params* __restrict__ p, uint32_t* out, int nperm) {
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = (threadIdx.x * 16 + i + blockIdx.x * 977u) % bb::P;
    for (int n = 0; n < nperm; n++) {
        p2::NoRecord rec;
        p2::permute_core<16>(s, p->rounds_p, p->ext_rc, p->int_rc, p->diag, p->ext_rc_mp, p->int_rc_mp, p->diag_c, rec, p->sum_mult_c);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= s[i];
    out[blockIdx.x * BLOCK + threadIdx.x] = acc;
}

// VARIANT 1: two independent permutations per lane, interleaved round by round (does instruction-level parallelism inside a
// wave buy what more waves per SIMD do not?  76 -> ~150 VGPRs, so half the waves)
template <int W>
__device__ __forceinline__ void internal_rounds_lazy2(uint32_t (&s)[W], uint32_t (&t)[W], int rounds_p, const uint32_t* __restrict__ int_rc_mp,
                                                      const int32_t* __restrict__ diag_c) {
    int32_t x[W], y[W];
#pragma unroll
    for (int i = 0; i < W; i++) x[i] = (int32_t)s[i], y[i] = (int32_t)t[i];
#pragma unroll 1
    for (int r = 0; r < rounds_p; r++) {
        {
            const uint32_t c0 = bb::umin((uint32_t)x[0], (uint32_t)x[0] + bb::P), d0 = bb::umin((uint32_t)y[0], (uint32_t)y[0] + bb::P);
            const int32_t a = (int32_t)(c0 + int_rc_mp[r]), b = (int32_t)(d0 + int_rc_mp[r]);
            const int32_t a2 = bb::smul(a, a), b2 = bb::smul(b, b);
            const int32_t a3 = bb::smul(a2, a), b3 = bb::smul(b2, b);
            const int32_t a6 = bb::smul(a3, a3), b6 = bb::smul(b3, b3);
            x[0] = bb::smul(a6, a);
            y[0] = bb::smul(b6, b);
        }
        int64_t v = 0, w = 0;
#pragma unroll
        for (int g = 0; g < W; g += 8) {
            int64_t u = 0, q = 0;
#pragma unroll
            for (int j = g; j < g + 8 && j < W; j++) u = bb::mad_i64(x[j], (int32_t)bb::R1, u), q = bb::mad_i64(y[j], (int32_t)bb::R1, q);
            v = bb::mad_i64(bb::sred(u), (int32_t)bb::R1, v);
            w = bb::mad_i64(bb::sred(q), (int32_t)bb::R1, w);
        }
#pragma unroll
        for (int i = 0; i < W; i++) x[i] = bb::sred(bb::mad_i64_u(x[i], diag_c[i], v)), y[i] = bb::sred(bb::mad_i64_u(y[i], diag_c[i], w));
    }
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = bb::umin((uint32_t)x[i], (uint32_t)x[i] + bb::P), t[i] = bb::umin((uint32_t)y[i], (uint32_t)y[i] + bb::P);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_perm2(const P16Params* __restrict__ p, uint32_t* out, int nperm) {
    uint32_t s[16], t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = (threadIdx.x * 16 + i + blockIdx.x * 977u) % bb::P, t[i] = (threadIdx.x * 16 + i + blockIdx.x * 1009u + 5) % bb::P;
    for (int n = 0; n < nperm; n++) {
        p2::NoRecord rec;
        p2::external_layer<16>(s);
        p2::external_layer<16>(t);
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = bb::add_pow7_mp(s[i], p->ext_rc_mp[r * 16 + i]), t[i] = bb::add_pow7_mp(t[i], p->ext_rc_mp[r * 16 + i]);
            p2::external_layer<16>(s);
            p2::external_layer<16>(t);
        }
        internal_rounds_lazy2<16>(s, t, p->rounds_p, p->int_rc_mp, p->diag_c);
#pragma unroll 1
        for (int r = 4; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = bb::add_pow7_mp(s[i], p->ext_rc_mp[r * 16 + i]), t[i] = bb::add_pow7_mp(t[i], p->ext_rc_mp[r * 16 + i]);
            p2::external_layer<16>(s);
            p2::external_layer<16>(t);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= s[i] ^ t[i];
    out[blockIdx.x * BLOCK + threadIdx.x] = acc;
}

int main() {
    P16Params hp{};
    for (int i = 0; i < 128; i++) hp.ext_rc[i] = (i * 2654435761u) % bb::P;
    for (int i = 0; i < lurkhip::P16_MAX_RP; i++) hp.int_rc[i] = (i * 40503u + 7) % bb::P;
    for (int i = 0; i < 16; i++) hp.diag[i] = (i * 69069u + 11) % bb::P;
    hp.rounds_p = 13;
    hp.sum_mult = bb::R1;
    hp.finish();
    P16Params* dp;
    uint32_t* dout;
    CK(hipMalloc(&dp, sizeof(hp)));
    CK(hipMemcpy(dp, &hp, sizeof(hp), hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    const int nperm = 64;
    for (int wg_per_cu : {1, 2, 4, 6, 8}) {
        const int blocks = 256 * wg_per_cu;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_perm<256>), dim3(blocks), dim3(256), 0, 0, dp, dout, nperm);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_perm<256>), dim3(blocks), dim3(256), 0, 0, dp, dout, nperm);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wperm_per_simd = (double)wg_per_cu * nperm;  // 4 waves per workgroup on 4 SIMDs
        printf("%d waves/SIMD: %8.3f ms  %7.0f cycles per wave-permutation  %6.2f Gperm/s\n", wg_per_cu, best, best * 2.4e6 / wperm_per_simd,
               (double)blocks * 256 * nperm / best * 1e-6);
    }
    printf("two interleaved permutations per lane (k_perm2):\n");
    for (int wg_per_cu : {1, 2, 3, 4}) {
        const int blocks = 256 * wg_per_cu;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_perm2<256>), dim3(blocks), dim3(256), 0, 0, dp, dout, nperm);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_perm2<256>), dim3(blocks), dim3(256), 0, 0, dp, dout, nperm);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wperm_per_simd = (double)wg_per_cu * nperm * 2;
        printf("%d waves/SIMD x 2: %8.3f ms  %7.0f cycles per wave-permutation  %6.2f Gperm/s\n", wg_per_cu, best, best * 2.4e6 / wperm_per_simd,
               (double)blocks * 256 * nperm * 2 / best * 1e-6);
    }
    return 0;
}
