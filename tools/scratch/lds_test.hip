#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(uint32_t* out, uint32_t words) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = i * 3u;
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) s += lds[words - 1 - i];
    atomicAdd(out, s);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    uint32_t* d; hipMalloc(&d, 4);
    for (size_t kb : {32, 64, 65, 96, 128, 160}) {
        hipMemset(d, 0, 4);
        hipError_t e0 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kb * 1024));
        hipLaunchKernelGGL(k, dim3(4), dim3(256), kb * 1024, 0, d, (uint32_t)(kb * 256));
        hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
        uint32_t h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%zu KB: attr %s launch %s sync %s out %u\n", kb, hipGetErrorString(e0), hipGetErrorString(e), hipGetErrorString(e2), h);
    }
    return 0;
}
