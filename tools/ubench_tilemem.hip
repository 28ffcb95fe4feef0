// Micro-benchmark (GPU box only): what the LDE's tile kernels' MEMORY patterns alone sustain -- persistent 1024-thread workgroups,
// a tile = 2^10 rows x 32 columns, thread (slot s, lane c) reads rows s + 32 j and writes rows 32 s + j (j < 32), no arithmetic,
// no LDS.  Source / destination are row-major with a pitch; rows of a tile are contiguous (stride 1) or strided (stride 2^10).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_tilemem.hip -o tools/ubench_tilemem.bin && tools/ubench_tilemem.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Args {
    const uint32_t* src;
    uint32_t* dst;
    uint32_t src_pitch, dst_pitch;      // words
    uint32_t src_row_stride, dst_row_stride;  // rows between a tile's consecutive rows (1 = contiguous, 1024 = strided pass)
    uint32_t n_chunks;                  // 32-column chunks per row
    uint32_t n_tiles;
};

// V = dwords per lane and access: V = 1: lane = column (32 lanes per row, 2 rows per wave); V = 4: lane = 4 columns (8 lanes per row, 8 rows per wave)
template <int V>
__global__ __launch_bounds__(1024, 4) void k(Args a) {
    typedef uint32_t vec __attribute__((ext_vector_type(V)));
    constexpr int LANES = 32 / V;            // lanes per row segment
    constexpr int SLOTS = 1024 / LANES;      // row slots
    constexpr int U = 1024 / SLOTS;          // rows per thread
    const int s = threadIdx.x / LANES, c = (threadIdx.x % LANES) * V;
    for (uint32_t id = blockIdx.x; id < a.n_tiles; id += gridDim.x) {
        const uint32_t blk = id / a.n_chunks, chunk = id - blk * a.n_chunks;
        // contiguous tiles: rows blk * 1024 + t; strided: rows t * stride + blk
        const size_t src_row0 = a.src_row_stride == 1 ? (size_t)blk * 1024 : blk;
        const size_t dst_row0 = a.dst_row_stride == 1 ? (size_t)blk * 1024 : blk;
        vec x[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const size_t row = src_row0 + (size_t)(s + SLOTS * j) * a.src_row_stride;
            x[j] = *reinterpret_cast<const vec*>(a.src + row * a.src_pitch + chunk * 32 + c);
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const size_t row = dst_row0 + (size_t)(U * s + j) * a.dst_row_stride;
            *reinterpret_cast<vec*>(a.dst + row * a.dst_pitch + chunk * 32 + c) = x[j];
        }
    }
}

template <int V>
int run(const char* name, Args a, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(1024), 0, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(1024), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double bytes = 2.0 * a.n_tiles * 1024 * 128;
    printf("%-86s V=%d blocks=%4d %7.3f ms %7.1f GB/s\n", name, V, blocks, ms, bytes / ms * 1e-6);
    return 0;
}

int main() {
    const size_t N = 1u << 20;
    uint32_t *src, *dst;
    CK(hipMalloc(&src, N * 128 * 4));
    CK(hipMalloc(&dst, N * 128 * 4));
    CK(hipMemset(src, 1, N * 128 * 4));
    // 2^20 rows x 96 columns (3 chunks): 3072 tiles
    for (int blocks : {256, 512}) {
        Args a{src, dst, 96, 96, 1, 1, 3, 3072};
        if (run<1>("contiguous rows -> contiguous rows, pitch 96 (aligned segments)", a, blocks)) return 1;
        if (run<4>("contiguous rows -> contiguous rows, pitch 96 (aligned segments)", a, blocks)) return 1;
        Args b{src, dst, 32, 78, 1, 1, 1, 1024};  // one slab -> one chunk of a 78-wide matrix
        b.n_chunks = 1;
        Args b3{src, dst, 96, 80, 1, 1, 2, 2048};
        if (run<1>("k_out: slab rows (pitch 96, 2 chunks) -> matrix rows pitch 80 (16-byte aligned segments)", b3, blocks)) return 1;
        if (run<4>("k_out: slab rows (pitch 96, 2 chunks) -> matrix rows pitch 80 (16-byte aligned segments)", b3, blocks)) return 1;
        Args b4{src, dst, 96, 78, 1, 1, 2, 2048};
        if (run<1>("k_out: slab rows (pitch 96, 2 chunks) -> matrix rows pitch 78 (8-byte aligned segments)", b4, blocks)) return 1;
        Args c{src, dst, 78, 96, 1024, 1024, 2, 2048};
        if (run<1>("k_in: strided matrix rows pitch 78 -> strided slab rows pitch 96", c, blocks)) return 1;
        Args c2{src, dst, 96, 96, 1024, 1024, 3, 3072};
        if (run<1>("strided rows pitch 96 -> strided rows pitch 96 (aligned segments)", c2, blocks)) return 1;
        if (run<4>("strided rows pitch 96 -> strided rows pitch 96 (aligned segments)", c2, blocks)) return 1;
        Args d{src, dst, 96, 96, 1, 1024, 3, 3072};
        if (run<1>("k_mid: contiguous rows -> strided rows, pitch 96", d, blocks)) return 1;
        if (run<4>("k_mid: contiguous rows -> strided rows, pitch 96", d, blocks)) return 1;
    }
    return 0;
}
