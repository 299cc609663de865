// tools/ubench_mfma_issue.hip -- does v_mfma_i32_4x4x4_16b_i8 run beside VALU work (same wave / other waves of the SIMD)?
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench_mfma_issue.hip -o tools/ubench_mfma_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (8 independent accumulators); 1: VALU only (8 independent multiply-add chains); 2: both in every wave, interleaved;
// 3: even waves MFMA, odd waves VALU
template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int iters, int a, int b) {
    v4i acc[8];
    uint64_t m[8];
    for (int i = 0; i < 8; i++) { acc[i] = v4i{i, i, i, i}; m[i] = threadIdx.x + i; }
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 1);
    uint32_t mul = (uint32_t)a | 1;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) m[i] = (uint64_t)(uint32_t)m[i] * mul + m[i];
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (int)m[i] + (int)(m[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    int* d; hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    const char* names[4] = {"MFMA only (32 per iteration)", "VALU only (32 v_mad_u64_u32 per iteration)", "both, every wave", "even waves MFMA, odd waves VALU"};
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 3, 5);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 3, 5);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 3, 5);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 3, 5);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // per SIMD: 8 waves x iters x 32 instructions of each kind the mode runs
        double per_simd = 8.0 * iters * 32 * (mode == 3 ? 0.5 : 1.0);
        printf("%-45s %8.3f ms   %.2f cycles (2.4 GHz) per instruction of each kind on a SIMD\n", names[mode], best, best * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
