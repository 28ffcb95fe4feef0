// Micro-benchmark (GPU box only): cycles per radix-2 DIF butterfly of lde.hip's register stage groups, data in registers, no
// global memory -- what the VALU alone allows the LDE's tile kernels.  Variants: per-thread twiddles read from LDS inside the
// group (as shipped), per-thread twiddles preloaded, scalar twiddles, butterflies without the two range corrections.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lurk_amd/csrc tools/ubench_bfly.hip -o tools/ubench_bfly.bin && tools/ubench_bfly.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "babybear.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int U = 32, LOG_U = 5, S = 32;
constexpr int ITERS = 64;

__device__ __forceinline__ void bfly(uint32_t& x, uint32_t& y, uint32_t tw) {
    const uint32_t sum = bb::add(x, y);
    const uint32_t r = (uint32_t)bb::smul((int32_t)(x - y), (int32_t)tw);
    y = bb::umin(r, r + bb::P);
    x = sum;
}
__device__ __forceinline__ void bfly_u(uint32_t& x, uint32_t& y, uint32_t tw_uniform) {
    const uint32_t sum = bb::add(x, y);
    const uint32_t r = (uint32_t)bb::sred(bb::mad_i64_u((int32_t)(x - y), (int32_t)tw_uniform, 0));
    y = bb::umin(r, r + bb::P);
    x = sum;
}
__device__ __forceinline__ void bfly_raw(uint32_t& x, uint32_t& y, uint32_t tw) {  // no corrections: NOT a correct butterfly, the instruction floor
    const uint32_t sum = x + y;
    const uint32_t r = (uint32_t)bb::smul((int32_t)(x - y), (int32_t)tw);
    y = r;
    x = sum;
}

// K butterflies in lockstep: every step of the dependent chain (difference, product, m = lo * p^-1, reduction, corrections) is issued for
// all K before the next step, so that a wave's consecutive instructions do not wait for one another
template <int K>
__device__ __forceinline__ void bfly_k(uint32_t* (&xs)[K], uint32_t* (&ys)[K], const uint32_t (&tw)[K]) {
    int32_t d[K];
    int64_t t[K];
    int32_t m[K];
    uint32_t sum[K];
#pragma unroll
    for (int i = 0; i < K; i++) d[i] = (int32_t)(*xs[i] - *ys[i]);
#pragma unroll
    for (int i = 0; i < K; i++) t[i] = bb::mad_i64(d[i], (int32_t)tw[i], 0);
#pragma unroll
    for (int i = 0; i < K; i++) sum[i] = *xs[i] + *ys[i];
#pragma unroll
    for (int i = 0; i < K; i++) m[i] = (int32_t)((uint32_t)t[i] * bb::MU);
#pragma unroll
    for (int i = 0; i < K; i++) t[i] = bb::mad_i64(m[i], -(int32_t)bb::P, t[i]);
#pragma unroll
    for (int i = 0; i < K; i++) *xs[i] = bb::umin(sum[i], sum[i] - bb::P);
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t r = (uint32_t)(t[i] >> 32);
        *ys[i] = bb::umin(r, r + bb::P);
    }
}

template <int MODE>
__global__ __launch_bounds__(1024, 4) void k(uint32_t* out, const uint32_t* tw_g) {
    __shared__ uint32_t tw[1024];
    tw[threadIdx.x] = tw_g[threadIdx.x];
    __syncthreads();
    const int s = threadIdx.x >> 5;
    uint32_t x[U];
#pragma unroll
    for (int j = 0; j < U; j++) x[j] = (threadIdx.x * 2654435761u + j * 40503u + blockIdx.x) % bb::P;
    uint32_t twr[U];
    if (MODE == 1) {
#pragma unroll
        for (int m = 1; m < U; m++) twr[m] = tw[s + S * m];
    }
    uint32_t tws[U];
    if (MODE == 2) {
#pragma unroll
        for (int m = 1; m < U; m++) tws[m] = (uint32_t)__builtin_amdgcn_readfirstlane((int)tw[m]);
    }
    for (int it = 0; it < ITERS; it++) {
        if (MODE >= 4) {
            constexpr int K = MODE == 4 ? 2 : (MODE == 5 ? 4 : 8);
#pragma unroll
            for (int g = LOG_U - 1; g >= 0; g--) {
#pragma unroll
                for (int p0 = 0; p0 < U / 2; p0 += K) {
                    uint32_t *xs[K], *ys[K], tws[K];
#pragma unroll
                    for (int i = 0; i < K; i++) {
                        const int p = p0 + i;  // pair index -> j with bit g clear
                        const int j = ((p >> g) << (g + 1)) | (p & ((1 << g) - 1));
                        xs[i] = &x[j];
                        ys[i] = &x[j | (1 << g)];
                        tws[i] = tw[s + S * ((1 << g) + (j & ((1 << g) - 1)))];
                    }
                    bfly_k<K>(xs, ys, tws);
                }
            }
            asm volatile("" ::: "memory");
            continue;
        }
#pragma unroll
        for (int g = LOG_U - 1; g >= 0; g--) {
#pragma unroll
            for (int j = 0; j < U; j++) {
                if (j & (1 << g)) continue;
                const int m = (1 << g) + (j & ((1 << g) - 1));
                if (MODE == 0) bfly(x[j], x[j | (1 << g)], tw[s + S * m]);
                else if (MODE == 1) bfly(x[j], x[j | (1 << g)], twr[m]);
                else if (MODE == 2) bfly_u(x[j], x[j | (1 << g)], tws[m]);
                else bfly_raw(x[j], x[j | (1 << g)], tw[s + S * m]);
            }
        }
        asm volatile("" ::: "memory");
    }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < U; j++) acc ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
int run(const char* name, uint32_t* out, const uint32_t* tw) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, tw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, tw);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bflies = (double)blocks * 1024 * ITERS * 80;   // per lane
    const double wave_bflies_per_simd = bflies / 64 / 1024;     // 256 CUs x 4 SIMDs
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-44s %8.3f ms  %6.1f cycles per wave-butterfly per SIMD  (%5.1f G butterflies/s)\n", name, ms, cycles / wave_bflies_per_simd, bflies / ms * 1e-6);
    return 0;
}

int main() {
    uint32_t *out, *tw;
    CK(hipMalloc(&out, 256 * 4 * 1024 * 4));
    CK(hipMalloc(&tw, 1024 * 4));
    uint32_t h[1024];
    for (int i = 0; i < 1024; i++) h[i] = (uint32_t)((i * 2654435761ull + 12345) % bb::P);
    CK(hipMemcpy(tw, h, sizeof h, hipMemcpyHostToDevice));
    if (run<0>("twiddles read from LDS inside the group", out, tw)) return 1;
    if (run<1>("twiddles preloaded into 31 VGPRs", out, tw)) return 1;
    if (run<2>("twiddles in scalar registers", out, tw)) return 1;
    if (run<3>("no range corrections (floor), LDS twiddles", out, tw)) return 1;
    if (run<4>("2 butterflies in lockstep, LDS twiddles", out, tw)) return 1;
    if (run<5>("4 butterflies in lockstep, LDS twiddles", out, tw)) return 1;
    if (run<6>("8 butterflies in lockstep, LDS twiddles", out, tw)) return 1;
    return 0;
}
