// tools/ubench_mfma_mds.hip -- can the 4x4x4 i8 MFMA (16 independent blocks of 4 lanes; B and D are lane-local) carry the dense
// 12x12 MDS layer of Poseidon?  (1) probes the operand layout, (2) checks an MFMA MDS layer against poseidon_mds_add on random
// loose states, (3) times R full rounds (12 s-boxes + MDS) with either layer.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Izkm_amd/csrc -Iinclude tools/ubench_mfma_mds.hip -o tools/ubench_mfma_mds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "poseidon_dev.h"

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_probe(const int* a, const int* b, int* out) {
    int l = threadIdx.x;
    v4i c = {0, 0, 0, 0};
    v4i d = __builtin_amdgcn_mfma_i32_4x4x4i8(a[l], b[l], c, 0, 0, 0);
    for (int v = 0; v < 4; v++) out[4 * l + v] = d[v];
}

// ---- MFMA MDS layer.  Lane-local view of v_mfma_i32_4x4x4_16b_i8 (block = 4 consecutive lanes):
//   D[v] (this lane) = C[v] + sum_k Abyte_k(lane 4*blk + v) * Bbyte_k(this lane)
// so with A = row (lane & 3) of a 4x4 block of the MDS matrix and B = four state bytes of this lane, D[v] is output row v for
// this lane's own state.  State bytes are unsigned: B carries byte ^ 0x80 (= byte - 128 as int8) and the accumulators start at
// 128 * rowsum.
struct mds_consts {
    int a[3][3];  // [output group][input group]: bytes M[4 ig + (lane & 3)][4 g + k]
};
__device__ __forceinline__ mds_consts mds_make_consts() {
    mds_consts c;
    const int row = threadIdx.x & 3;
    for (int ig = 0; ig < 3; ig++)
        for (int g = 0; g < 3; g++) {
            uint32_t w = 0;
            for (int k = 0; k < 4; k++) {
                // poseidon_m1(i, j) with a run-time row
                constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
                int i = 4 * ig + row, j = 4 * g + k;
                uint32_t m = C[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
                w |= m << (8 * k);
            }
            c.a[ig][g] = (int)w;
        }
    return c;
}

// 4x4 byte transpose: in[k] = dword of word k; out[b] = (in[0].b, in[1].b, in[2].b, in[3].b)
__device__ __forceinline__ void transpose4(const uint32_t in[4], uint32_t out[4]) {
    // v_perm_b32(hi, lo, sel): byte i of the result = byte sel_i of the 8-byte value {hi, lo} (0..3 = lo, 4..7 = hi)
    uint32_t t0 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400);  // (in0.b0, in1.b0, in0.b1, in1.b1)
    uint32_t t1 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602);  // (in0.b2, in1.b2, in0.b3, in1.b3)
    uint32_t t2 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400);
    uint32_t t3 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602);
    out[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100);  // (t0.lo16, t2.lo16)
    out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302);  // (t0.hi16, t2.hi16)
    out[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
    out[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
}

template <bool ADD>
__device__ __forceinline__ void poseidon_mds_add_mfma(uint64_t s[12], const uint64_t* add, const mds_consts& mc) {
    uint32_t B[3][8];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            lo[k] = (uint32_t)s[4 * g + k] ^ 0x80808080u;
            hi[k] = (uint32_t)(s[4 * g + k] >> 32) ^ 0x80808080u;
        }
        transpose4(lo, &B[g][0]);
        transpose4(hi, &B[g][4]);
    }
    int m16 = 65536, one = 1;
    asm("" : "+s"(m16));
    asm("" : "+s"(one));
#pragma unroll
    for (int ig = 0; ig < 3; ig++) {
        v4i acc[8];
        const v4i zero = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 8; b++) acc[b] = __builtin_amdgcn_mfma_i32_4x4x4i8(mc.a[ig][0], (int)B[0][b], zero, 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 8; b++) acc[b] = __builtin_amdgcn_mfma_i32_4x4x4i8(mc.a[ig][1], (int)B[1][b], acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 8; b++) acc[b] = __builtin_amdgcn_mfma_i32_4x4x4i8(mc.a[ig][2], (int)B[2][b], acc[b], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int r = 4 * ig + v;
            // signed digits D_b = sum_k m_k (byte_k - 128): value = sum_b D_b 2^(8b) + 128 rowsum 0x01010101 per half
            const int p01 = (acc[1][v] << 8) + acc[0][v], p23 = (acc[3][v] << 8) + acc[2][v];
            const int p45 = (acc[5][v] << 8) + acc[4][v], p67 = (acc[7][v] << 8) + acc[6][v];
            const uint64_t off = (uint64_t)(r == 0 ? 128 * 264 : 128 * 256) * 0x01010101ull;
            int64_t al = (int64_t)((ADD ? (uint64_t)(uint32_t)add[r] : 0) + off), ah = (int64_t)((ADD ? add[r] >> 32 : 0) + off);
            al += (int64_t)p23 * m16;
            al += (int64_t)p01 * one;
            ah += (int64_t)p67 * m16;
            ah += (int64_t)p45 * one;
            s[r] = poseidon_fold((uint64_t)al, (uint64_t)ah);
        }
    }
}

__global__ void k_check(const uint64_t* in, uint64_t* out_ref, uint64_t* out_mfma, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;  // (n is a multiple of 64: MFMA needs the whole wave)
    const mds_consts mc = mds_make_consts();
    uint64_t a[12], b[12];
    for (int k = 0; k < 12; k++) a[k] = b[k] = in[k * n + i];
    poseidon_mds_add<true>(a, &PC::ZKM_POSEIDON_RC[12]);
    poseidon_mds_add_mfma<true>(b, &PC::ZKM_POSEIDON_RC[12], mc);
    for (int k = 0; k < 12; k++) {
        out_ref[k * n + i] = gl_canon(a[k]);
        out_mfma[k * n + i] = gl_canon(b[k]);
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_rounds(uint64_t* data, size_t n, int rounds) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const mds_consts mc = mds_make_consts();
    uint64_t s[12];
    for (int k = 0; k < 12; k++) s[k] = data[k * n + i];
    if (MODE == 3) {  // MFMA layer, waves of a SIMD start out of phase (waves 4 apart share a SIMD)
        const int w = (threadIdx.x >> 6) >> 2;
        if (w & 1) __builtin_amdgcn_s_sleep(20);
    }
#pragma unroll 1
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = poseidon_sbox7(s[k]);
        if (MODE == 0) poseidon_mds_add<true>(s, &PC::ZKM_POSEIDON_RC[12 * (r & 7)]);
        else if (MODE == 1 || MODE == 3) poseidon_mds_add_mfma<true>(s, &PC::ZKM_POSEIDON_RC[12 * (r & 7)], mc);
        // MODE 2: s-boxes only
    }
    for (int k = 0; k < 12; k++) data[k * n + i] = s[k];
}

static uint64_t rnd64() { return ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand(); }

int main() {
    // ---- (1) layout probe
    {
        std::vector<int> a(64), b(64), out(256);
        for (int l = 0; l < 64; l++) { a[l] = (int)rnd64(); b[l] = (int)rnd64(); }
        int *da, *db, *dout;
        hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024);
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dout);
        hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
        int ok1 = 1, ok2 = 1;
        for (int l = 0; l < 64; l++)
            for (int v = 0; v < 4; v++) {
                int blk = l / 4, e1 = 0, e2 = 0;
                for (int k = 0; k < 4; k++) {
                    e1 += (int)(int8_t)(a[4 * blk + v] >> (8 * k)) * (int)(int8_t)(b[l] >> (8 * k));
                    e2 += (int)(int8_t)(a[l] >> (8 * k)) * (int)(int8_t)(b[4 * blk + v] >> (8 * k));
                }
                if (out[4 * l + v] != e1) ok1 = 0;
                if (out[4 * l + v] != e2) ok2 = 0;
            }
        printf("layout: D[v](lane) = sum_k A_k(lane 4 blk + v) B_k(lane): %s;  A/B swapped: %s\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no");
    }
    // ---- (2) MDS check
    const size_t n = 1 << 16;
    std::vector<uint64_t> h(12 * n);
    for (auto& x : h) x = rnd64();
    for (int k = 0; k < 12; k++) { h[k * n + 0] = ~0ull; h[k * n + 1] = 0; h[k * n + 2] = 0xFFFFFFFF00000000ull; h[k * n + 3] = 0x8080808080808080ull; }
    uint64_t *din, *d1, *d2;
    hipMalloc(&din, 96 * n); hipMalloc(&d1, 96 * n); hipMalloc(&d2, 96 * n);
    hipMemcpy(din, h.data(), 96 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, din, d1, d2, n);
    std::vector<uint64_t> r1(12 * n), r2(12 * n);
    hipMemcpy(r1.data(), d1, 96 * n, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, 96 * n, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < 12 * n; i++) bad += r1[i] != r2[i];
    printf("MDS layer via MFMA vs multiply-add form on %zu states: %zu mismatching words\n", n, bad);
    // ---- (3) timing
    const size_t N = 1 << 22;
    uint64_t* dd; hipMalloc(&dd, 96 * N); hipMemset(dd, 5, 96 * N);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 64;
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            if (mode == 1) hipLaunchKernelGGL(k_rounds<1>, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            if (mode == 2) hipLaunchKernelGGL(k_rounds<2>, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            if (mode == 3) hipLaunchKernelGGL(k_rounds<3>, dim3(N / 512), dim3(512), 0, 0, dd, N, R);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: %d rounds x %zu states: %.3f ms  = %.2f ns per round and wave-lane-group (%.1f G rounds/s)\n",
               mode == 0 ? "multiply-add MDS" : mode == 1 ? "MFMA MDS" : mode == 2 ? "s-boxes only" : "MFMA MDS, skewed waves", R, N, best, best * 1e6 / R / (N / 64), (double)R * N / best / 1e6);
    }
    return 0;
}
