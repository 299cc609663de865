// tools/ubench2.hip -- per-instruction VALU issue cost on gfx950, low loop overhead (32 ops per iteration, 16 chains).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 1024
#define CH 16
#define REP 2
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t seed) {
    uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t a[CH], b[CH];
    uint64_t x[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) { a[i] = (uint32_t)(seed * (lane + 1) + i); b[i] = a[i] ^ 0x9e3779b9u; x[i] = seed * (lane + 3) + i; }
    uint32_t m = (uint32_t)seed | 1u;
    uint64_t m64 = seed | 0x8000000000000001ULL;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++)
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                if (OP == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(a[i]), "v"(m) : "vcc");
                if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"(m64));
                if (OP == 5) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(x[i]), "v"(m64) : "vcc");
                if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == 8) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 10) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 11) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(m));
                if (OP == 12) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (OP == 14) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x[i]));
                if (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));
            }
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) acc ^= a[i] ^ x[i];
    out[lane] = acc;
}
template <int OP> static void run(const char* name, uint64_t* d, double ghz, int nsimd) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x1234567ULL); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x1234567ULL);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    double wave_ops = (double)blocks * threads / 64 * ITERS * CH * REP;
    printf("%-18s %8.3f ms  %6.2f cycles/wave-instr/SIMD @%.1fGHz\n", name, ms, ms * 1e-3 * ghz * 1e9 * nsimd / wave_ops, ghz);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double ghz = p.clockRate / 1e6; int nsimd = p.multiProcessorCount * 4;
    uint64_t* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_add_u32", d, ghz, nsimd); run<1>("v_mov_b32", d, ghz, nsimd); run<2>("v_cndmask_b32", d, ghz, nsimd);
    run<10>("v_xor_b32", d, ghz, nsimd); run<9>("v_add3_u32", d, ghz, nsimd); run<15>("v_alignbit_b32", d, ghz, nsimd);
    run<11>("v_perm_b32", d, ghz, nsimd); run<7>("v_add_co_u32", d, ghz, nsimd); run<8>("v_addc_co_u32", d, ghz, nsimd);
    run<13>("v_mad_u32_u24", d, ghz, nsimd); run<6>("v_mul_lo_u32", d, ghz, nsimd); run<12>("v_mul_hi_u32", d, ghz, nsimd);
    run<3>("v_mad_u64_u32", d, ghz, nsimd); run<4>("v_lshl_add_u64", d, ghz, nsimd); run<5>("v_cmp_lt_u64", d, ghz, nsimd);
    run<14>("v_lshlrev_b64", d, ghz, nsimd);
    return 0;
}
