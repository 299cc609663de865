// tools/ubench3.hip -- x^7 s-box throughput for different 64x64 modular-multiply formulations (development aid).
// 12 independent chains per lane, as in the Poseidon full rounds.  Prints ms and s-boxes/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../zkm_amd/csrc/gl_dev.h"
#define ITERS 256

__device__ __forceinline__ uint64_t mul_old(uint64_t a, uint64_t b) { return gl_reduce128(a * b, __umul64hi(a, b)); }
__device__ __forceinline__ uint64_t mul_new(uint64_t a, uint64_t b) { return gl_mul_loose(a, b); }
__device__ __forceinline__ uint64_t sqr_new(uint64_t a) { return gl_mul_loose(a, a); }
// kA: chained mads with zero-extended addends
__device__ __forceinline__ uint64_t mul_a(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t p00 = (uint64_t)al * bl;
    uint64_t t = (uint64_t)al * bh + (p00 >> 32);
    uint64_t u = (uint64_t)ah * bl + (uint32_t)t;
    uint64_t hi = (uint64_t)ah * bh + ((t >> 32) + (u >> 32));
    uint64_t lo = (u << 32) | (uint32_t)p00;
    return gl_reduce128(lo, hi);
}
__device__ __forceinline__ uint64_t red4(uint64_t lo, uint64_t hi) {
    uint64_t hi_hi = hi >> 32; uint32_t hi_lo = (uint32_t)hi;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;
    uint64_t r = (uint64_t)hi_lo * 0xFFFFFFFFu + t0;
    if (r < t0) r += GL_EPS;
    return r;
}
__device__ __forceinline__ uint64_t red5(uint64_t lo, uint64_t hi) {
    uint64_t hi_hi = hi >> 32; uint32_t hi_lo = (uint32_t)hi;
    uint64_t t0 = lo - hi_hi;
    t0 -= (lo < hi_hi) ? GL_EPS : 0;
    uint64_t r = (uint64_t)hi_lo * 0xFFFFFFFFu + t0;
    r += (r < t0) ? GL_EPS : 0;
    return r;
}
// borrow detected on the high word only: lo < hi_hi (< 2^32) iff lo_hi == 0 && lo_lo < hi_hi
__device__ __forceinline__ uint64_t red6(uint64_t lo, uint64_t hi) {
    uint32_t h1 = (uint32_t)(hi >> 32), h0 = (uint32_t)hi;
    uint64_t t0 = lo - h1;
    uint32_t m = (uint32_t)((int32_t)((uint32_t)(t0 >> 32) & ~(uint32_t)(lo >> 32)) >> 31);  // borrow <=> high word went 0 -> 0xFFFFFFFF
    t0 -= m;   // m = 0xFFFFFFFF = EPS on borrow
    uint64_t r = (uint64_t)h0 * 0xFFFFFFFFu + t0;
    r += (r < t0) ? GL_EPS : 0;
    return r;
}
// red6 with the wrap of h0 * EPS + t0 taken from the multiply-add's own carry-out (inline asm: mad, select, add)
__device__ __forceinline__ uint64_t red7(uint64_t lo, uint64_t hi) {
    uint32_t h1 = (uint32_t)(hi >> 32), h0 = (uint32_t)hi;
    uint64_t t0 = lo - h1;
    uint32_t m = (uint32_t)((int32_t)((uint32_t)(t0 >> 32) & ~(uint32_t)(lo >> 32)) >> 31);
    t0 -= m;
    uint64_t r, carry;
    uint32_t add;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, %1" : "=&v"(r), "=&s"(carry), "=v"(add) : "v"(h0), "v"(t0));
    return r + add;
}
// product with the middle terms summed in one 64-bit accumulator, its overflow taken from the carry-out
__device__ __forceinline__ uint64_t mul_c(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t p00 = (uint64_t)al * bl;
    uint64_t m1 = (uint64_t)al * bh + (p00 >> 32);
    uint64_t m2, c;
    uint32_t cw;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, 1, %1" : "=&v"(m2), "=&s"(c), "=v"(cw) : "v"(ah), "v"(bl), "v"(m1));
    uint64_t hi = (uint64_t)ah * bh + ((m2 >> 32) | ((uint64_t)cw << 32));
    uint64_t lo = (m2 << 32) | (uint32_t)p00;
    return red7(lo, hi);
}
template <int R> __device__ __forceinline__ uint64_t mul_r(uint64_t a, uint64_t b) {
    uint64_t lo, hi; gl_mul_wide(a, b, lo, hi);
    return R == 4 ? red4(lo, hi) : R == 5 ? red5(lo, hi) : R == 6 ? red6(lo, hi) : red7(lo, hi);
}
template <int V>
__device__ __forceinline__ uint64_t sbox(uint64_t x) {
    if (V == 8) { uint64_t x2 = mul_c(x, x), x4 = mul_c(x2, x2), x3 = mul_c(x, x2); return mul_c(x3, x4); }
    if (V >= 4) { uint64_t x2 = mul_r<V>(x, x), x4 = mul_r<V>(x2, x2), x3 = mul_r<V>(x, x2); return mul_r<V>(x3, x4); }
    if (V == 0) { uint64_t x2 = mul_old(x, x), x4 = mul_old(x2, x2), x3 = mul_old(x, x2); return mul_old(x3, x4); }
    if (V == 1) { uint64_t x2 = mul_new(x, x), x4 = mul_new(x2, x2), x3 = mul_new(x, x2); return mul_new(x3, x4); }
    if (V == 2) { uint64_t x2 = sqr_new(x), x4 = sqr_new(x2), x3 = mul_new(x, x2); return mul_new(x3, x4); }
    { uint64_t x2 = mul_a(x, x), x4 = mul_a(x2, x2), x3 = mul_a(x, x2); return mul_a(x3, x4); }
}
template <int V, bool FENCE>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t seed) {
    uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed * (lane + 3) + i;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            s[i] = sbox<V>(s[i]);
            if (FENCE && (i & 1)) __builtin_amdgcn_sched_barrier(0);
        }
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) acc ^= s[i];
    out[lane] = acc;
}
template <int V, bool FENCE> static void run(const char* name, uint64_t* d) {
    const int blocks = 256 * 16, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, FENCE>), dim3(blocks), dim3(threads), 0, 0, d, 0x1234567ULL); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL((k<V, FENCE>), dim3(blocks), dim3(threads), 0, 0, d, 0x1234567ULL);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    double n = (double)blocks * threads * ITERS * 12;
    uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.3f ms  %7.2f G sbox/s  (check %016llx)\n", name, ms, n / ms / 1e6, (unsigned long long)h);
}
int main() {
    uint64_t* d; hipMalloc(&d, 256 * 16 * 256 * 8);
    run<0, true>("old (a*b, umul64hi) fence", d); run<0, false>("old nofence", d);
    run<1, true>("new shared products fence", d); run<1, false>("new nofence", d);
    run<2, true>("new + squarings fence", d); run<2, false>("new + squarings nofence", d);
    run<3, true>("chained zext addends fence", d); run<3, false>("chained nofence", d);
    run<4, true>("chained + mad-fused reduce", d); run<5, true>("chained + select-EPS reduce", d); run<6, true>("chained + sign-mask borrow", d);
    run<7, true>("sign-mask + mad carry-out asm", d); run<7, false>("mad carry-out asm nofence", d);
    run<8, true>("carry-out product + reduce", d); run<8, false>("carry-out product nofence", d);
    return 0;
}
