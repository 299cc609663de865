// Development aid: do a VALU-bound kernel (Poseidon permutations in registers) and an HBM-bound kernel (grid-stride copy) overlap
// when launched on two streams?  Prints the time of each alone and of both together.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkm_amd/csrc -I include tools/overlap_test.hip -o /tmp/overlap_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "poseidon_dev.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_hash(uint64_t* out, int reps) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[12];
    for (int i = 0; i < 12; i++) s[i] = j * 12 + i;
    for (int r = 0; r < reps; r++) poseidon_permute(s);
    out[j] = s[0];
}
__global__ __launch_bounds__(256) void k_copy(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main(int argc, char** argv) {
    int hash_lds = argc > 1 ? atoi(argv[1]) : 0;
    int copy_blocks = argc > 2 ? atoi(argv[2]) : 2048;
    hipStream_t a, b;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, least));
    size_t nh = 1 << 22, nc = (size_t)1 << 28;  // 4 GB copy buffers (16 B elements)
    uint64_t* dout; ulonglong2 *cin, *cout;
    CK(hipMalloc(&dout, nh * 8)); CK(hipMalloc(&cin, nc * 16)); CK(hipMalloc(&cout, nc * 16));
    CK(hipMemset(cin, 1, nc * 16));
    if (hash_lds) CK(hipFuncSetAttribute((const void*)k_hash, hipFuncAttributeMaxDynamicSharedMemorySize, hash_lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](bool h, bool c) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, a));
        CK(hipStreamWaitEvent(b, e0, 0));
        if (c) for (int k = 0; k < 6; k++) hipLaunchKernelGGL(k_copy, dim3(copy_blocks), dim3(256), 0, a, cin, cout, nc);
        if (h) hipLaunchKernelGGL(k_hash, dim3(nh / 256), dim3(256), hash_lds, b, dout, 33);
        CK(hipEventRecord(e1, b));
        CK(hipStreamWaitEvent(a, e1, 0));
        CK(hipEventRecord(e1, a));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms;
    };
    run(true, true);
    float th = run(true, false), tc = run(false, true), tb = run(true, true);
    printf("hash_lds=%d copy_blocks=%d: hash alone %.2f ms, copy alone %.2f ms (%.0f GB/s), both %.2f ms (sum %.2f)\n", hash_lds, copy_blocks, th, tc,
           6 * 2.0 * nc * 16 / tc / 1e6, tb, th + tc);
    return 0;
}
