// tools/ubench.hip -- VALU instruction-rate microbenchmarks for the 64-bit modular arithmetic on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Reports issue cycles per wave64 instruction per SIMD (assuming the measured clock), for the candidate
// building blocks of a Goldilocks multiply: v_mad_u64_u32, v_mul_lo/hi_u32, 24-bit multiplies, 64-bit adds,
// f64 FMA -- and for the library's gl_mul_loose / gl_add_loose / poseidon_mds as compiled.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../zkm_amd/csrc/poseidon_dev.h"

#define ITERS 2048
#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k_bench(uint64_t* out, uint64_t seed) {
    uint64_t x[CHAINS];
    uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = seed * (lane + 1) + i * 0x9E3779B97F4A7C15ULL;
    uint32_t m = (uint32_t)seed | 1u;
    uint64_t m64 = seed | 0x8000000000000001ULL;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) {  // v_mad_u64_u32: 32x32 + 64
                x[i] = (uint64_t)(uint32_t)x[i] * m + x[i];
            } else if (OP == 1) {  // v_mul_lo_u32
                uint32_t a = (uint32_t)x[i];
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(m));
                x[i] = a;
            } else if (OP == 2) {  // v_mul_hi_u32
                uint32_t a = (uint32_t)x[i];
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(m));
                x[i] = a;
            } else if (OP == 3) {  // v_mul_u32_u24
                uint32_t a = (uint32_t)x[i];
                asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a) : "v"(a), "v"(m));
                x[i] = a;
            } else if (OP == 4) {  // v_mad_u32_u24
                uint32_t a = (uint32_t)x[i];
                asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(a) : "v"(a), "v"(m));
                x[i] = a;
            } else if (OP == 5) {  // 64-bit add (v_lshl_add_u64 or add_co/addc pair)
                x[i] = x[i] + m64;
            } else if (OP == 6) {  // v_fma_f64
                double d = __longlong_as_double(x[i]);
                double e = __longlong_as_double(m64);
                asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(d) : "v"(d), "v"(e));
                x[i] = __double_as_longlong(d);
            } else if (OP == 7) {  // gl_mul_loose as compiled
                x[i] = gl_mul_loose(x[i], m64);
            } else if (OP == 8) {  // gl_add_loose as compiled
                x[i] = gl_add_loose(x[i], m64);
            } else if (OP == 9) {  // v_add_u32 (plain VALU reference)
                uint32_t a = (uint32_t)x[i];
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(m));
                x[i] = a;
            } else if (OP == 10) {  // v_mul_hi_u32_u24
                uint32_t a = (uint32_t)x[i];
                asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(a) : "v"(a), "v"(m));
                x[i] = a;
            } else if (OP == 11) {  // v_mul_f64
                double d = __longlong_as_double(x[i]);
                double e = __longlong_as_double(m64);
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(d), "v"(e));
                x[i] = __double_as_longlong(d);
            } else if (OP == 12) {  // gl_mul (canonical)
                x[i] = gl_mul(x[i], m64);
            } else if (OP == 13) {  // v_cvt_f64_u32 + back
                uint32_t a = (uint32_t)x[i];
                double d;
                asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d) : "v"(a));
                x[i] = __double_as_longlong(d);
            }
        }
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc ^= x[i];
    out[lane] = acc;
}

__global__ __launch_bounds__(256) void k_bench_perm(uint64_t* out, uint64_t seed, int reps) {
    uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed * (lane + 1) + i;
    for (int r = 0; r < reps; r++) poseidon_permute(s);
    out[lane] = s[0] ^ s[5];
}

__global__ __launch_bounds__(256) void k_bench_mds(uint64_t* out, uint64_t seed, int reps) {
    uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed * (lane + 1) + i;
    for (int r = 0; r < reps; r++) poseidon_mds(s);
    out[lane] = s[0] ^ s[5];
}

template <int OP>
static void run(const char* name, uint64_t* d_out, double clock_ghz, int nsimd) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 0x1234567ULL);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 0x1234567ULL);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double wave_ops = (double)blocks * threads / 64 * ITERS * CHAINS;
    double cyc = ms * 1e-3 * clock_ghz * 1e9 * nsimd / wave_ops;
    printf("%-28s %8.3f ms  %7.2f cycles/wave-op/SIMD  (%.2f Top/s lane-ops)\n", name, ms, cyc, wave_ops * 64 / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    double clock_ghz = p.clockRate / 1e6;
    int nsimd = p.multiProcessorCount * 4;
    printf("device %s  CUs %d  clock %.3f GHz\n", p.name, p.multiProcessorCount, clock_ghz);
    uint64_t* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * 8);
    run<9>("v_add_u32", d_out, clock_ghz, nsimd);
    run<0>("v_mad_u64_u32", d_out, clock_ghz, nsimd);
    run<1>("v_mul_lo_u32", d_out, clock_ghz, nsimd);
    run<2>("v_mul_hi_u32", d_out, clock_ghz, nsimd);
    run<3>("v_mul_u32_u24", d_out, clock_ghz, nsimd);
    run<10>("v_mul_hi_u32_u24", d_out, clock_ghz, nsimd);
    run<4>("v_mad_u32_u24", d_out, clock_ghz, nsimd);
    run<5>("add u64", d_out, clock_ghz, nsimd);
    run<6>("v_fma_f64", d_out, clock_ghz, nsimd);
    run<11>("v_mul_f64", d_out, clock_ghz, nsimd);
    run<13>("v_cvt_f64_u32", d_out, clock_ghz, nsimd);
    run<8>("gl_add_loose", d_out, clock_ghz, nsimd);
    run<7>("gl_mul_loose", d_out, clock_ghz, nsimd);
    run<12>("gl_mul (canonical)", d_out, clock_ghz, nsimd);
    {
        const int blocks = 256 * 8, threads = 256, reps = 16;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int which = 0; which < 2; which++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (which == 0) hipLaunchKernelGGL(k_bench_perm, dim3(blocks), dim3(threads), 0, 0, d_out, 77ULL, reps);
                else hipLaunchKernelGGL(k_bench_mds, dim3(blocks), dim3(threads), 0, 0, d_out, 77ULL, reps * 64);
                hipEventRecord(b);
                hipEventSynchronize(b);
            }
            float ms;
            hipEventElapsedTime(&ms, a, b);
            double n = (double)blocks * threads * reps * (which ? 64 : 1);
            double cyc = ms * 1e-3 * clock_ghz * 1e9 * nsimd / (n / 64);
            printf("%-28s %8.3f ms  %9.0f cycles/wave-call/SIMD  %.3f G calls/s\n", which ? "poseidon_mds" : "poseidon_permute", ms, cyc,
                   n / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
