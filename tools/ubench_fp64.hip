// Micro-benchmark (GPU box only): issue rate of the gfx950 instructions a double-precision / 64-bit-accumulator
// formulation of the Poseidon2 linear layers would use, next to the int32 add baseline.  Prints T instr/s (lanes) per op.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_fp64.hip -o /tmp/ubench64 && /tmp/ubench64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ILP = 8;
constexpr int ITERS = 4096;

struct OpAddU32 {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAddF64 {
    using T = double;
    __device__ static T init(uint32_t s) { return (double)s; }
    __device__ static void f(T& x, T y) { asm volatile("v_add_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return (uint32_t)(long long)x; }
};
struct OpFmaF64 {
    using T = double;
    __device__ static T init(uint32_t s) { return (double)s; }
    __device__ static void f(T& x, T y) { asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return (uint32_t)(long long)x; }
};
struct OpMulF64 {
    using T = double;
    __device__ static T init(uint32_t s) { return (double)s; }
    __device__ static void f(T& x, T y) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return (uint32_t)(long long)x; }
};
struct OpRndneF64 {
    using T = double;
    __device__ static T init(uint32_t s) { return (double)s * 0.37; }
    __device__ static void f(T& x, T) { asm volatile("v_rndne_f64 %0, %1" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return (uint32_t)(long long)x; }
};
// i32 -> f64 -> i32 round trip: two instructions per call
struct OpCvtRound {
    using T = int32_t;
    __device__ static T init(uint32_t s) { return (int32_t)s; }
    __device__ static void f(T& x, T) {
        double d;
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(x));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(x) : "v"(d));
    }
    __device__ static uint32_t fin(T x) { return (uint32_t)x; }
};
struct OpMadI64 {
    using T = long long;
    __device__ static T init(uint32_t s) { return (long long)s; }
    __device__ static void f(T& x, T y) {
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"((int32_t)y), "v"((int32_t)(y >> 3)) : "vcc");
    }
    __device__ static uint32_t fin(T x) { return (uint32_t)x ^ (uint32_t)(x >> 32); }
};
struct OpMulLo {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s | 1u; }
    __device__ static void f(T& x, T y) { asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMulHiI32 {
    using T = int32_t;
    __device__ static T init(uint32_t s) { return (int32_t)(s | 1u); }
    __device__ static void f(T& x, T y) { asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return (uint32_t)x; }
};


struct OpMinU32 {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_min_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpSubU32 {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAdd3U32 {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_add3_u32 %0, %1, %2, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAddConst {  // v_add_u32 with an inline constant operand (one VGPR source)
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T) { asm volatile("v_add_u32 %0, 17, %1" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAddSgpr {  // v_add_u32 with an SGPR operand
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T) { asm volatile("v_add_u32 %0, s4, %1" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpBBAdd {  // add, sub p, min: the canonical modular add
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s % 2013265921u; }
    __device__ static void f(T& x, T y) {
        uint32_t s_, t_;
        asm volatile("v_add_u32 %0, %2, %3\n\tv_add_u32 %1, 0x87ffffff, %0\n\tv_min_u32 %0, %0, %1" : "=&v"(s_), "=&v"(t_) : "v"(x), "v"(y));
        x = s_;
    }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpSmul {  // signed Montgomery product as the compiler emits it: mad_i64_i32, mul_lo, mul_hi_i32, sub
    using T = int32_t;
    __device__ static T init(uint32_t s) { return (int32_t)(s % 2013265921u); }
    __device__ static void f(T& x, T y) {
        const long long t = (long long)x * y;
        const int32_t m = (int32_t)((uint32_t)t * 0x88000001u);
        const int32_t u = __mulhi(m, (int32_t)0x78000001);
        int32_t hi = (int32_t)(uint32_t)((unsigned long long)t >> 32);
        asm("" : "+v"(hi));
        x = hi - u;
    }
    __device__ static uint32_t fin(T x) { return (uint32_t)x; }
};
struct OpMulAdd2 {  // one multiply followed by two simple ops
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s | 1u; }
    __device__ static void f(T& x, T y) {
        asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_min_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    __device__ static uint32_t fin(T x) { return x; }
};

// smul variants: where the two constants live, and how the 64-bit product is formed
template <int VARIANT>
struct OpSmulV {
    using T = int32_t;
    __device__ static T init(uint32_t s) { return (int32_t)(s % 2013265921u); }
    __device__ static void f(T& x, T y) {
        uint32_t mu = 0x88000001u;
        int32_t pp = 0x78000001;
        if (VARIANT == 1 || VARIANT == 3) {  // opaque VGPR constants
            asm volatile("" : "+v"(mu));
            asm volatile("" : "+v"(pp));
        }
        if (VARIANT <= 1) {
            const long long t = (long long)x * y;
            const int32_t m = (int32_t)((uint32_t)t * mu);
            const int32_t u = __mulhi(m, pp);
            int32_t hi = (int32_t)(uint32_t)((unsigned long long)t >> 32);
            asm("" : "+v"(hi));
            x = hi - u;
        } else {  // separate low / high products (5 instructions)
            uint32_t lo = (uint32_t)x * (uint32_t)y;
            int32_t hi = __mulhi(x, y);
            asm("" : "+v"(lo));
            asm("" : "+v"(hi));
            const int32_t m = (int32_t)(lo * mu);
            const int32_t u = __mulhi(m, pp);
            x = hi - u;
        }
    }
    __device__ static uint32_t fin(T x) { return (uint32_t)x; }
};
struct OpMulLoSgpr {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s | 1u; }
    __device__ static void f(T& x, T) { asm volatile("v_mul_lo_u32 %0, %1, s4" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMulHiSgpr {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s | 1u; }
    __device__ static void f(T& x, T) { asm volatile("v_mul_hi_i32 %0, %1, s4" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAshr {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_ashrrev_i32 %0, 3, %1" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAnd {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpXor {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMaxI {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_max_i32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMinI {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_min_i32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMaxU {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_max_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpLshl {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpSubrev {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_subrev_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAddF32 {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMinF32 {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_min_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMed3 {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_med3_i32 %0, %1, %2, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpAndOr {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_and_or_b32 %0, %1, %2, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpLshlAdd {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpMov {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpBfe {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_bfe_i32 %0, %1, 31, 1" : "=v"(x) : "v"(x)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpCndmask {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(y)); }
    __device__ static uint32_t fin(T x) { return x; }
};
struct OpSubCo {
    using T = uint32_t;
    __device__ static T init(uint32_t s) { return s; }
    __device__ static void f(T& x, T y) { asm volatile("v_sub_co_u32 %0, vcc, %1, %2" : "=v"(x) : "v"(x), "v"(y) : "vcc"); }
    __device__ static uint32_t fin(T x) { return x; }
};

template <class Op>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    typename Op::T x[ILP], y = Op::init(seed | 3u);
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = Op::init((threadIdx.x + 1u) * 2654435761u + i * 97u + blockIdx.x);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) Op::f(x[i], y);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc ^= Op::fin(x[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class Op>
int run(const char* name, int instr_per_call, uint32_t* dout) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<Op>), dim3(blocks), dim3(threads), 0, 0, dout, 12345u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<Op>), dim3(blocks), dim3(threads), 0, 0, dout, 12345u + rep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double n = (double)blocks * threads * ILP * ITERS * instr_per_call;
    printf("%-28s %8.3f ms  %8.2f T lane-instr/s\n", name, best, n / best * 1e-9);
    return 0;
}

int main() {
    uint32_t* dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    run<OpAddU32>("v_add_u32", 1, dout);
    run<OpAshr>("v_ashrrev_i32", 1, dout);
    run<OpAnd>("v_and_b32", 1, dout);
    run<OpXor>("v_xor_b32", 1, dout);
    run<OpMaxI>("v_max_i32", 1, dout);
    run<OpMinI>("v_min_i32", 1, dout);
    run<OpMaxU>("v_max_u32", 1, dout);
    run<OpLshl>("v_lshlrev_b32", 1, dout);
    run<OpSubrev>("v_subrev_u32", 1, dout);
    run<OpAddF32>("v_add_f32", 1, dout);
    run<OpMinF32>("v_min_f32", 1, dout);
    run<OpMed3>("v_med3_i32", 1, dout);
    run<OpAndOr>("v_and_or_b32", 1, dout);
    run<OpLshlAdd>("v_lshl_add_u32", 1, dout);
    run<OpMov>("v_mov_b32", 1, dout);
    run<OpBfe>("v_bfe_i32", 1, dout);
    run<OpCndmask>("v_cndmask_b32", 1, dout);
    run<OpSubCo>("v_sub_co_u32", 1, dout);
    run<OpMinU32>("v_min_u32", 1, dout);
    run<OpSubU32>("v_sub_u32", 1, dout);
    run<OpAdd3U32>("v_add3_u32", 1, dout);
    run<OpAddConst>("v_add_u32 inline const", 1, dout);
    run<OpAddSgpr>("v_add_u32 sgpr", 1, dout);
    run<OpBBAdd>("bb::add (add,add,min)", 3, dout);
    run<OpSmul>("smul (mad64,mullo,mulhi,sub)", 4, dout);
    run<OpSmulV<0>>("smul mad64, literal consts", 4, dout);
    run<OpSmulV<1>>("smul mad64, VGPR consts", 4, dout);
    run<OpSmulV<2>>("smul lo/hi, literal consts", 5, dout);
    run<OpSmulV<3>>("smul lo/hi, VGPR consts", 5, dout);
    run<OpMulLoSgpr>("v_mul_lo_u32 sgpr", 1, dout);
    run<OpMulHiSgpr>("v_mul_hi_i32 sgpr", 1, dout);
    run<OpMulAdd2>("mul_lo + add + min", 3, dout);
    run<OpMulLo>("v_mul_lo_u32", 1, dout);
    run<OpMulHiI32>("v_mul_hi_i32", 1, dout);
    run<OpMadI64>("v_mad_i64_i32", 1, dout);
    run<OpAddF64>("v_add_f64", 1, dout);
    run<OpMulF64>("v_mul_f64", 1, dout);
    run<OpFmaF64>("v_fma_f64", 1, dout);
    run<OpRndneF64>("v_rndne_f64", 1, dout);
    run<OpCvtRound>("v_cvt_f64_i32 + v_cvt_i32_f64", 2, dout);
    return 0;
}
