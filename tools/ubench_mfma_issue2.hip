// VALU-bound mix with a minority of MFMAs: is the MFMA time hidden?  bursts vs fine interleave, dependent vs independent VALU work
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

// per iteration: NV multiply-adds (8 chains) and NM MFMAs (8 accumulators).  MODE 0: VALU only; 1: MFMA burst then VALU burst;
// 2: fine interleave (1 MFMA per NV / NM VALU); 3: MFMA only
template <int MODE, int NV, int NM>
__global__ __launch_bounds__(256) void k(int* out, int iters, int a, int b) {
    v4i acc[8];
    uint64_t m[8];
    for (int i = 0; i < 8; i++) { acc[i] = v4i{i, i, i, i}; m[i] = threadIdx.x + i; }
    uint32_t mul = (uint32_t)a | 1;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NM; i++) acc[i & 7] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, acc[i & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NV; i++) m[i & 7] = (uint64_t)(uint32_t)m[i & 7] * mul + m[i & 7];
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NM; i++) {
                acc[i & 7] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, acc[i & 7], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / NM; j++) m[(i * (NV / NM) + j) & 7] = (uint64_t)(uint32_t)m[(i * (NV / NM) + j) & 7] * mul + m[(i * (NV / NM) + j) & 7];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NV; i++) m[i & 7] = (uint64_t)(uint32_t)m[i & 7] * mul + m[i & 7];
        } else {
#pragma unroll
            for (int i = 0; i < NM; i++) acc[i & 7] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, acc[i & 7], 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (int)m[i] + (int)(m[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run(int* d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, 512, 72>), dim3(blocks), dim3(256), 0, 0, d, iters, 3, 5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    int* d; hipMalloc(&d, 256 * 4096 * 4);
    const int iters = 200;
    for (int wg = 1; wg <= 8; wg *= 2) {   // workgroups of 4 waves per CU: wg waves per SIMD
        int blocks = 256 * wg;
        float t0 = run<0>(d, blocks, iters), t3 = run<3>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters);
        printf("%d waves/SIMD: VALU only (512 mads) %.3f ms, MFMA only (72) %.3f ms, bursts %.3f ms, interleaved %.3f ms\n", wg, t0, t3, t1, t2);
    }
    return 0;
}
