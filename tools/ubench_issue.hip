// Micro-benchmark (GPU box only): VALU issue cost of the instruction sequences of the Poseidon2 / NTT inner loops on gfx950,
// with long unrolled bodies (UNROLL x ILP instructions between two loop branches) so that the loop overhead does not dilute
// the figure the way it does in tools/ubench_fp64.hip (8 instructions per trip there).  Prints cycles per wave-instruction
// and SIMD (2.4 GHz, 1024 SIMDs) for several occupancies.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ILP = 8;
constexpr int UNROLL = 16;
constexpr int ITERS = 512;

struct St {
    uint32_t x[ILP];
    uint64_t t[ILP];
    uint32_t y, mu, np;
};

// ---- single instructions
struct OpAdd { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpAddLit { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_add_u32 %0, 0x87ffffff, %0" : "+v"(s.x[i])); } };
struct OpMin { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpMulLo { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpMulLoS { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mul_lo_u32 %0, %0, s4" : "+v"(s.x[i])); } };
struct OpMulHi { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpMad64Vcc { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(s.t[i]) : "v"(s.x[i]), "v"(s.y) : "vcc"); } };
struct OpMad64S { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mad_i64_i32 %0, s[6:7], %1, %2, %0" : "+v"(s.t[i]) : "v"(s.x[i]), "v"(s.y) : "s6", "s7"); } };
struct OpMadU64 { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(s.t[i]) : "v"(s.x[i]), "v"(s.y) : "vcc"); } };
struct OpMad64Zero { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(s.t[i]) : "v"(s.x[i]), "v"(s.y) : "vcc"); } };
struct OpMulI24 { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpMadU24 { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpSubCo { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(s.x[i]) : "v"(s.y) : "vcc"); } };
struct OpCndmask { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s.x[i]) : "v"(s.y) : "vcc"); } };
struct OpPkAdd { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpAdd3 { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(s.x[i]) : "v"(s.y)); } };
struct OpSNop { static constexpr int N = 1; __device__ static void f(St& s, int i) { asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(s.x[i]) : "v"(s.y)); } };

// ---- sequences
// canonical modular add: add, add literal, min
struct OpBBAdd { static constexpr int N = 3; __device__ static void f(St& s, int i) {
    uint32_t t;
    asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, 0x87ffffff, %0\n\tv_min_u32 %0, %0, %1" : "+v"(s.x[i]), "=&v"(t) : "v"(s.y)); } };
// modular add with carry select: add, sub_co, cndmask
struct OpBBAddSel { static constexpr int N = 3; __device__ static void f(St& s, int i) {
    uint32_t t;
    asm volatile("v_add_u32 %0, %0, %2\n\tv_subrev_co_u32 %1, vcc, %3, %0\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(s.x[i]), "=&v"(t) : "v"(s.y), "v"(s.np) : "vcc"); } };
// signed Montgomery square: mad64 (zero addend), mul_lo by MU (VGPR), mad64 by -P (VGPR); the result is the high half
struct OpSmulV { static constexpr int N = 3; __device__ static void f(St& s, int i) {
    uint64_t t; uint32_t m;
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, 0" : "=v"(t) : "v"(s.x[i]) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(m) : "v"((uint32_t)t), "v"(s.mu));
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(m), "v"(s.np) : "vcc");
    s.x[i] = (uint32_t)(t >> 32); } };
// same with the constants in SGPRs / as literals the way the compiler places them
struct OpSmulS { static constexpr int N = 3; __device__ static void f(St& s, int i) {
    uint64_t t; uint32_t m;
    asm volatile("v_mad_i64_i32 %0, s[6:7], %1, %1, 0" : "=v"(t) : "v"(s.x[i]) : "s6", "s7");
    asm volatile("v_mul_lo_u32 %0, %1, s4" : "=v"(m) : "v"((uint32_t)t));
    asm volatile("v_mad_i64_i32 %0, s[6:7], %1, %2, %0" : "+v"(t) : "v"(m), "v"(s.np) : "s6", "s7");
    s.x[i] = (uint32_t)(t >> 32); } };
// x^7 + correction: the external-round S-box (add rc, 4 products, add p, min)
struct OpSbox { static constexpr int N = 15; __device__ static void f(St& s, int i) {
    auto mul = [&](uint32_t a, uint32_t b) {
        uint64_t t; uint32_t m;
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(a), "v"(b) : "vcc");
        asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(m) : "v"((uint32_t)t), "v"(s.mu));
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(m), "v"(s.np) : "vcc");
        return (uint32_t)(t >> 32); };
    uint32_t x;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(s.x[i]), "v"(s.y));
    uint32_t x2 = mul(x, x), x3 = mul(x2, x), x6 = mul(x3, x3), x7 = mul(x6, x), c;
    asm volatile("v_add_u32 %0, 0x78000001, %1" : "=v"(c) : "v"(x7));
    asm volatile("v_min_u32 %0, %1, %2" : "=v"(s.x[i]) : "v"(x7), "v"(c)); } };
// mul_lo / mul_hi formulation: lo = a*a, hi = mulhi(a,a), m = lo*MU, u = mulhi(m, P), r = hi - u
struct OpSmulLoHi { static constexpr int N = 5; __device__ static void f(St& s, int i) {
    uint32_t lo, hi;
    asm volatile("v_mul_lo_u32 %1, %0, %0\n\tv_mul_hi_i32 %2, %0, %0\n\tv_mul_lo_u32 %1, %1, %3\n\tv_mul_hi_i32 %1, %1, %4\n\tv_sub_u32 %0, %2, %1"
                 : "+v"(s.x[i]), "=&v"(lo), "=&v"(hi) : "v"(s.mu), "v"(s.np)); } };

template <class Op>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    St s;
#pragma unroll
    for (int i = 0; i < ILP; i++) {
        s.x[i] = (threadIdx.x + 1u) * 2654435761u + i * 97u + blockIdx.x;
        s.t[i] = ((uint64_t)s.x[i] << 32) | (s.x[i] * 7u);
    }
    s.y = seed | 3u;
    s.mu = 0x88000001u + (seed & 0u);
    s.np = 0x87ffffffu + (seed & 0u);
    asm volatile("" : "+v"(s.y), "+v"(s.mu), "+v"(s.np));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) Op::f(s, i);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc ^= s.x[i] ^ (uint32_t)s.t[i] ^ (uint32_t)(s.t[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class Op>
int run(const char* name, uint32_t* dout, int iters = ITERS) {
    printf("%-34s", name);
    for (int wg_per_cu : {2, 4, 8}) {  // 256-thread workgroups: 2, 4, 8 waves per SIMD
        const int blocks = 256 * wg_per_cu, threads = 256;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<Op>), dim3(blocks), dim3(threads), 0, 0, dout, 12345u, iters);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<Op>), dim3(blocks), dim3(threads), 0, 0, dout, 12345u + rep, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD = waves/SIMD * ILP*UNROLL*ITERS*N ; cycles = ms * 2.4e6
        const double winstr = (double)wg_per_cu * ILP * UNROLL * iters * Op::N;
        printf("  %dw: %5.2f cyc/instr", wg_per_cu, best * 2.4e6 / winstr);
    }
    printf("\n");
    return 0;
}

int main() {
    uint32_t* dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    run<OpAdd>("v_add_u32", dout);
    run<OpAddLit>("v_add_u32 literal", dout);
    run<OpMin>("v_min_u32", dout);
    run<OpMulLo>("v_mul_lo_u32", dout);
    run<OpMulLoS>("v_mul_lo_u32 sgpr", dout);
    run<OpMulHi>("v_mul_hi_u32", dout);
    run<OpMulI24>("v_mul_u32_u24", dout);
    run<OpMadU24>("v_mad_u32_u24", dout);
    run<OpMad64Vcc>("v_mad_i64_i32 vcc", dout);
    run<OpMad64S>("v_mad_i64_i32 s[6:7]", dout);
    run<OpMadU64>("v_mad_u64_u32 vcc", dout);
    run<OpMad64Zero>("v_mad_i64_i32 zero addend", dout);
    run<OpSubCo>("v_sub_co_u32", dout);
    run<OpCndmask>("v_cndmask_b32", dout);
    run<OpPkAdd>("v_pk_add_u16", dout);
    run<OpAdd3>("v_add3_u32", dout);
    run<OpSNop>("v_add_u32 + s_nop 0 (per add)", dout);
    run<OpBBAdd>("mod add: add, add lit, min", dout);
    run<OpBBAddSel>("mod add: add, subrev_co, cndmask", dout);
    run<OpSmulV>("smul: mad64, mul_lo, mad64", dout);
    run<OpSmulS>("smul: sgpr carry, sgpr MU", dout);
    run<OpSbox>("sbox: add, 4 smul, add, min", dout);
    run<OpSmulLoHi>("smul: lo, hi, lo, hi, sub", dout);
    // sustained runs (tens of milliseconds per launch): does the clock hold?
    run<OpAdd>("v_add_u32, 40x longer", dout, ITERS * 40);
    run<OpSbox>("sbox, 40x longer", dout, ITERS * 40);
    run<OpSbox>("sbox, 200x longer", dout, ITERS * 200);
    return 0;
}
