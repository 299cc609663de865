// tools/ubench_tail.hip -- the carry tails of gl_dev.h (the EPS correction at the end of gl_mul_loose / gl_reduce128, gl_add_rr,
// gl_sub_rr) and poseidon_fold against host arithmetic: whole-chip launches, one wave per SIMD (instructions of a wave issue back to
// back: where missing wait states would show), products by wave-uniform constants, and the partial rounds of the Poseidon witness
// kernel (register pressure, AGPR spills).  Prints mismatch counts and dependent-chain rates.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -o tools/ubench_tail tools/ubench_tail.hip
// Written for the round-5 experiment recorded in profiles/r05_carry_tail_experiment.txt (not adopted); kept as the check any
// future change to those tails has to pass before it goes near the library.
#include "../zkm_amd/csrc/poseidon_dev.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_mul(const uint64_t* a, const uint64_t* b, uint64_t* o, int reps) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint64_t x = a[i], y = b[i];
    for (int r = 0; r < reps; r++) x = gl_mul_loose(x, y);
    o[i] = x;
}
__global__ void k_addsub(const uint64_t* a, const uint64_t* b, uint64_t* o, int reps) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint64_t x = a[i], y = b[i];
    for (int r = 0; r < reps; r++) { x = gl_add_rr(x, y); y = gl_sub_rr(y, x); }
    o[i] = x ^ (y * 3);
}
__global__ void k_fold(const uint64_t* a, const uint64_t* b, uint64_t* o) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    o[i] = poseidon_fold(a[i] >> 5, b[i] >> 5);
}
__constant__ uint64_t KTAB[12] = {25, 0x8000000000000001ULL, 0xFFFFFFFF00000000ULL, 3, 0x123456789ABCDEFULL, 0xFFFFFFFFULL,
                                  0x100000000ULL, 0xFEDCBA9876543210ULL, 7, 0xFFFFFFFEFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL, 1};
__global__ void k_mulconst(const uint64_t* a, uint64_t* o) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint64_t x = a[i];
    uint64_t d = gl_mul_loose(x, 25);
#pragma unroll
    for (int q = 1; q < 12; q++) d = gl_add_loose(d, gl_mul_loose(x + q, KTAB[q]));
    o[i] = d;
}
// the partial rounds of the Poseidon witness kernel (witness.hip k_poseidon_trace), 12 words in, 12 + 44 words out
__global__ __launch_bounds__(256) void k_partial(const uint64_t* in, uint64_t* out, size_t n, int stage) {
    size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canon(in[i * n + r]);
    uint64_t* o = out + r;
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_loose(s[i], PC::ZKM_POSEIDON_FAST_FIRST_RC[i]);
    uint64_t t[12];
    t[0] = s[0];
#pragma unroll
    for (int c = 1; c < 12; c++) {
        uint64_t acc = 0;
#pragma unroll
        for (int q = 1; q < 12; q++) acc = gl_add_loose(acc, gl_mul_loose(s[q], PC::ZKM_POSEIDON_FAST_INIT[q - 1][c - 1]));
        t[c] = acc;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = t[i];
    if (stage == 1) {
#pragma unroll
        for (int i = 0; i < 12; i++) o[(size_t)i * n] = gl_canon(s[i]);
        return;
    }
#pragma unroll 1
    for (int q = 0; q < 22; q++) {
        if (stage == 2 && q == 1) break;
        uint64_t x = s[0];
        gl_t x3 = gl_mul(gl_mul_loose(x, x), x);
        gl_t x7 = gl_mul(x, gl_mul_loose(x3, x3));
        o[(size_t)(12 + 2 * q) * n] = x3;
        o[(size_t)(12 + 2 * q + 1) * n] = x7;
        uint64_t s0 = gl_add_loose(x7, PC::ZKM_POSEIDON_FAST_RC[q]);
        uint64_t d = gl_mul_loose(s0, 25);
#pragma unroll
        for (int i = 1; i < 12; i++) d = gl_add_loose(d, gl_mul_loose(s[i], PC::ZKM_POSEIDON_FAST_W_HATS[q][i - 1]));
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = gl_add_loose(s[i], gl_mul_loose(s0, PC::ZKM_POSEIDON_FAST_VS[q][i - 1]));
        s[0] = d;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) o[(size_t)i * n] = gl_canon(s[i]);
}
static void h_partial(const uint64_t* in, uint64_t* out, int stage = 0) {   // out[0..11] state, out[12..55] x3/x7
    auto mm = [](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)(a % GL_P) * (b % GL_P)) % GL_P); };
    auto ad = [](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)(a % GL_P) + (b % GL_P)) % GL_P); };
    uint64_t s[12], t[12];
    for (int i = 0; i < 12; i++) s[i] = ad(in[i], pc_host::ZKM_POSEIDON_FAST_FIRST_RC[i]);
    t[0] = s[0];
    for (int c = 1; c < 12; c++) { uint64_t acc = 0; for (int q = 1; q < 12; q++) acc = ad(acc, mm(s[q], pc_host::ZKM_POSEIDON_FAST_INIT[q - 1][c - 1])); t[c] = acc; }
    for (int i = 0; i < 12; i++) s[i] = t[i];
    for (int q = 0; q < (stage == 1 ? 0 : stage == 2 ? 1 : 22); q++) {
        uint64_t x = s[0], x3 = mm(mm(x, x), x), x7 = mm(x, mm(x3, x3));
        out[12 + 2 * q] = x3; out[13 + 2 * q] = x7;
        uint64_t s0 = ad(x7, pc_host::ZKM_POSEIDON_FAST_RC[q]), d = mm(s0, 25);
        for (int i = 1; i < 12; i++) d = ad(d, mm(s[i], pc_host::ZKM_POSEIDON_FAST_W_HATS[q][i - 1]));
        for (int i = 1; i < 12; i++) s[i] = ad(s[i], mm(s0, pc_host::ZKM_POSEIDON_FAST_VS[q][i - 1]));
        s[0] = d;
    }
    for (int i = 0; i < 12; i++) out[i] = s[i];
}
static uint64_t h_add(uint64_t a, uint64_t b) { uint64_t s = a + b; if (s < a) { uint64_t t = s + GL_EPS; s = t < s ? t + GL_EPS : t; } return s; }
static uint64_t h_sub(uint64_t a, uint64_t b) { uint64_t d = a - b; if (a < b) { uint64_t t = d - GL_EPS; d = t > d ? t - GL_EPS : t; } return d; }
static uint64_t canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
static uint64_t h_mulmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)(a % GL_P) * (b % GL_P)) % GL_P); }

int main() {
    const size_t n = 1 << 22;
    std::vector<uint64_t> a(n), b(n), o(n);
    uint64_t s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < n; i++) { a[i] = rnd(); b[i] = rnd(); }
    const uint64_t edge[] = {0, 1, GL_P - 1, GL_P, GL_P + 1, ~0ULL, 0xFFFFFFFFULL, 0x100000000ULL, 0xFFFFFFFF00000000ULL, 0xFFFFFFFEFFFFFFFFULL};
    size_t e = 0;
    for (uint64_t x : edge) for (uint64_t y : edge) { a[e] = x; b[e] = y; e++; }
    uint64_t *da, *db, *dout;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, n * 8));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // products
    for (int reps : {1, 64}) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mul, dim3(n / 256), dim3(256), 0, 0, da, db, dout, reps); CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; i += (reps == 1 ? 1 : 97)) {
            uint64_t x = a[i] % GL_P; for (int r = 0; r < reps; r++) x = h_mulmod(x, b[i]);
            if (canon(o[i]) != x) bad++;
        }
        printf("gl_mul_loose x%d: %zu mismatches, %.3f ms (%.2f G products/s)\n", reps, bad, ms, n * (double)reps / ms / 1e6);
    }
    for (int reps : {1, 64}) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_addsub, dim3(n / 256), dim3(256), 0, 0, da, db, dout, reps); CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) {
            uint64_t x = a[i], y = b[i];
            for (int r = 0; r < reps; r++) { x = h_add(x, y); y = h_sub(y, x); }
            if (o[i] != (x ^ (y * 3))) bad++;     // the loose forms are deterministic word for word
        }
        printf("gl_add_rr/gl_sub_rr x%d: %zu mismatches, %.3f ms (%.2f G pairs/s)\n", reps, bad, ms, n * (double)reps / ms / 1e6);
    }
    {
        hipLaunchKernelGGL(k_fold, dim3(n / 256), dim3(256), 0, 0, da, db, dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) {
            unsigned __int128 v = (unsigned __int128)(a[i] >> 5) + ((unsigned __int128)(b[i] >> 5) << 32);
            if (canon(o[i]) != (uint64_t)(v % GL_P)) bad++;
        }
        printf("poseidon_fold: %zu mismatches\n", bad);
    }
    // one wave per SIMD: the instructions of a wave issue back to back, which is where missing wait states show
    {
        size_t bad = 0, bad2 = 0, bad3 = 0;
        const size_t m = 65536;
        for (size_t off = 0; off + m <= n; off += m * 8) {
            hipLaunchKernelGGL(k_mul, dim3(m / 64), dim3(64), 0, 0, da + off, db + off, dout + off, 16);
            hipLaunchKernelGGL(k_addsub, dim3(m / 64), dim3(64), 0, 0, da + off, db + off, dout + off + m, 16);
            hipLaunchKernelGGL(k_fold, dim3(m / 64), dim3(64), 0, 0, da + off, db + off, dout + off + 2 * m);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(o.data() + off, dout + off, 3 * m * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < m; i++) {
                uint64_t x = a[off + i] % GL_P; for (int r = 0; r < 16; r++) x = h_mulmod(x, b[off + i]);
                if (canon(o[off + i]) != x) bad++;
                uint64_t u = a[off + i], y = b[off + i];
                for (int r = 0; r < 16; r++) { u = h_add(u, y); y = h_sub(y, u); }
                if (o[off + m + i] != (u ^ (y * 3))) bad2++;
                unsigned __int128 v = (unsigned __int128)(a[off + i] >> 5) + ((unsigned __int128)(b[off + i] >> 5) << 32);
                if (canon(o[off + 2 * m + i]) != (uint64_t)(v % GL_P)) bad3++;
            }
        }
        printf("one wave per SIMD: products %zu, add/sub %zu, fold %zu mismatches\n", bad, bad2, bad3);
    }
    {
        const size_t m = 4096;
        for (int stage : {1, 2, 0}) {
        hipLaunchKernelGGL(k_partial, dim3(m / 256), dim3(256), 0, 0, da, dout, m, stage);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dout, 56 * m * 8, hipMemcpyDeviceToHost));
        size_t bad = 0; int first = -1; size_t percol[12] = {0};
        for (size_t r = 0; r < m; r++) {
            uint64_t in[12], ex[56];
            for (int i = 0; i < 12; i++) in[i] = a[i * m + r];
            h_partial(in, ex, stage);
            for (int j = 0; j < (stage == 0 ? 56 : 12); j++) if (o[j * m + r] != ex[j]) { bad++;  if (j < 12) percol[j]++; if (first < 0 || j < first) first = j; }
        }
        printf("witness partial rounds (stage %d): %zu mismatching cells (lowest column %d) per state word:", stage, bad, first);
        for (int j = 0; j < 12; j++) printf(" %zu", percol[j]);
        printf("\n");
        }
    }
    {
        hipLaunchKernelGGL(k_mulconst, dim3(n / 256), dim3(256), 0, 0, da, dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost));
        const uint64_t K[12] = {25, 0x8000000000000001ULL, 0xFFFFFFFF00000000ULL, 3, 0x123456789ABCDEFULL, 0xFFFFFFFFULL,
                                0x100000000ULL, 0xFEDCBA9876543210ULL, 7, 0xFFFFFFFEFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL, 1};
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) {
            unsigned __int128 acc = h_mulmod(a[i], 25);
            for (int q = 1; q < 12; q++) acc += h_mulmod(a[i] + q, K[q]);
            if (canon(o[i]) != (uint64_t)(acc % GL_P)) bad++;
        }
        printf("products by uniform constants: %zu mismatches\n", bad);
    }
    return 0;
}
