// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this repository's two big kernel
// families (MI355X_MICROARCH.md "HBM": "calibrate on a known byte count in your own access pattern").
// Every kernel moves a KNOWN number of bytes through one pattern over buffers far larger than the 256 MiB Infinity Cache;
// tools/calibrate_fetch.sh runs this binary under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and
// tools/calibrate_fetch.py divides: factor = known bytes / counter bytes, per pattern -> profiles/pmc_traffic.json.
//
//   k_read_wide16     16 B per lane, fully coalesced streaming read (the guide's reference pattern: FETCH_SIZE = 1/2)
//   k_read_row16      ONE ROW PER LANE, two 16-byte loads per 32-byte chunk, chunks in turn (merkle.hip: load_chunk); rows of
//                     78 words = 312 bytes, so a wave's 64 loads of one chunk touch 64 different rows
//   k_read_pair8      8 bytes per lane, column pair fastest across 16 lanes then the next row (ntt.hip tile loads: a 32-column
//                     chunk of a 78-column matrix = 128-byte row segments at a 312-byte row stride)
//   k_write_pair8     the NTT pass's stores: same map, 8-byte stores
//   k_write_digest32  one 32-byte digest per lane, two 16-byte stores (the sponges' output)
//   k_write_wide16    16 B per lane coalesced streaming write
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_fetch.hip -o tools/ubench_fetch.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                                    \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                          \
            exit(1);                                                                                \
        }                                                                                           \
    } while (0)

constexpr uint32_t W = 78;          // words per row (the eval chip)
constexpr uint32_t CHUNK_COLS = 32;  // columns of an NTT column chunk

__global__ __launch_bounds__(256) void k_read_wide16(const uint4* __restrict__ in, size_t n_vec, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;  // never true for the fill pattern: keeps the loads alive, writes nothing
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void k_read_row16(const uint32_t* __restrict__ in, size_t n_rows, uint32_t* __restrict__ sink) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    uint32_t acc = 0;
    const uint32_t* r = in + row * W;
    for (uint32_t g = 0; g + 8 <= W; g += 8) {
        const u32x4_a4* a = (const u32x4_a4*)(r + g);
        const u32x4_a4 lo = a[0], hi = a[1];
        acc ^= lo.x ^ lo.y ^ lo.z ^ lo.w ^ hi.x ^ hi.y ^ hi.z ^ hi.w;
    }
    for (uint32_t c = W & ~7u; c < W; c++) acc ^= r[c];
    if (acc == 0x12345678u) sink[0] = acc;
}

// rows x 32-column chunk of a W-column matrix: thread t of a 256-thread block handles (row = t / 16, pair = t % 16)
__global__ __launch_bounds__(256) void k_read_pair8(const uint32_t* __restrict__ in, size_t n_rows, uint32_t col0, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t row = (size_t)blockIdx.x * 16 + threadIdx.x / 16; row < n_rows; row += (size_t)gridDim.x * 16) {
        const uint2 v = *(const uint2*)(in + row * W + col0 + 2 * (threadIdx.x % 16));
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the same map over a matrix of WD words per row (compile-time): WD = 32 makes the rows contiguous (dense 8-byte-per-lane
// streaming), WD = 96 keeps 128-byte ALIGNED segments at a 384-byte stride -- known bytes == touched lines in both, which
// separates "how a request width is tallied" from "how many lines an unaligned segment touches"
template <uint32_t WD>
__global__ __launch_bounds__(256) void k_read_pair8_w(const uint32_t* __restrict__ in, size_t n_rows, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t row = (size_t)blockIdx.x * 16 + threadIdx.x / 16; row < n_rows; row += (size_t)gridDim.x * 16) {
        const uint2 v = *(const uint2*)(in + row * WD + 2 * (threadIdx.x % 16));
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_dense4(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= in[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_write_pair8(uint32_t* __restrict__ out, size_t n_rows, uint32_t col0) {
    for (size_t row = (size_t)blockIdx.x * 16 + threadIdx.x / 16; row < n_rows; row += (size_t)gridDim.x * 16)
        *(uint2*)(out + row * W + col0 + 2 * (threadIdx.x % 16)) = make_uint2((uint32_t)row, threadIdx.x);
}

__global__ __launch_bounds__(256) void k_write_digest32(uint32_t* __restrict__ out, size_t n_rows) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    uint4* d = (uint4*)(out + row * 8);
    d[0] = make_uint4((uint32_t)row, 1, 2, 3);
    d[1] = make_uint4(4, 5, 6, (uint32_t)row);
}

__global__ __launch_bounds__(256) void k_write_wide16(uint4* __restrict__ out, size_t n_vec) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

__global__ void k_fill(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u) | 1u;
}

int main() {
    const size_t n_rows = (size_t)1 << 22;           // 2^22 rows x 78 words = 1.31 GB: five times the Infinity Cache
    const size_t words = n_rows * W;
    uint32_t *buf = nullptr, *sink = nullptr, *dig = nullptr;
    CHECK(hipMalloc(&buf, words * 4));
    CHECK(hipMalloc(&dig, n_rows * 32));
    CHECK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, words);
    CHECK(hipDeviceSynchronize());
    // known bytes per launch, printed for tools/calibrate_fetch.py
    const size_t rows32 = words / 32, rows96 = words / 96;
    printf("{\"k_read_wide16\": %zu, \"k_read_row16\": %zu, \"k_read_pair8\": %zu, \"k_write_pair8\": %zu, \"k_write_digest32\": %zu, \"k_write_wide16\": %zu, "
           "\"k_read_pair8_w<32>\": %zu, \"k_read_pair8_w<96>\": %zu, \"k_read_dense4\": %zu}\n",
           words * 4, words * 4, n_rows * CHUNK_COLS * 4, n_rows * CHUNK_COLS * 4, n_rows * 32, words * 4, rows32 * 128, rows96 * 128, words * 4);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read_wide16, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, words / 4, sink);
        hipLaunchKernelGGL(k_read_row16, dim3((unsigned)(n_rows / 256)), dim3(256), 0, 0, buf, n_rows, sink);
        hipLaunchKernelGGL(k_read_pair8, dim3(8192), dim3(256), 0, 0, buf, n_rows, 0u, sink);
        hipLaunchKernelGGL(k_read_pair8_w<32>, dim3(8192), dim3(256), 0, 0, buf, words / 32, sink);
        hipLaunchKernelGGL(k_read_pair8_w<96>, dim3(8192), dim3(256), 0, 0, buf, words / 96, sink);
        hipLaunchKernelGGL(k_read_dense4, dim3(8192), dim3(256), 0, 0, buf, words, sink);
        hipLaunchKernelGGL(k_write_pair8, dim3(8192), dim3(256), 0, 0, buf, n_rows, 32u);
        hipLaunchKernelGGL(k_write_digest32, dim3((unsigned)(n_rows / 256)), dim3(256), 0, 0, dig, n_rows);
        hipLaunchKernelGGL(k_write_wide16, dim3(8192), dim3(256), 0, 0, (uint4*)buf, words / 4);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, words);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipFree(buf));
    CHECK(hipFree(dig));
    CHECK(hipFree(sink));
    return 0;
}
