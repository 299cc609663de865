// tools/ubench_pcie.hip -- host -> device: SDMA copies (hipMemcpyAsync from pinned memory) against a kernel that READS the pinned host
// memory itself (zero-copy), 2 GiB, for one and for four streams.   hipcc -O3 --offload-arch=gfx950 tools/ubench_pcie.hip -o tools/ubench_pcie
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)
__global__ void k_copy(const ulonglong2* __restrict__ src, ulonglong2* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    void *h, *d;
    CHECK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CHECK(hipMalloc(&d, bytes));
    memset(h, 1, bytes);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int nst : {1, 4}) {
        std::vector<hipStream_t> st(nst);
        for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a, 0));
                CHECK(hipStreamWaitEvent(st[0], a, 0));
                for (int k = 0; k < nst; k++) {
                    if (k) CHECK(hipStreamWaitEvent(st[k], a, 0));
                    const size_t off = bytes / nst * k, len = bytes / nst;
                    if (mode == 0) CHECK(hipMemcpyAsync((char*)d + off, (char*)h + off, len, hipMemcpyHostToDevice, st[k]));
                    else hipLaunchKernelGGL(k_copy, dim3(mode == 1 ? 256 : 2048), dim3(256), 0, st[k], (const ulonglong2*)((char*)h + off), (ulonglong2*)((char*)d + off), len / 16);
                }
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(b, 0));
                CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep) printf("%d stream(s), %s: %.1f GB/s\n", nst, mode == 0 ? "hipMemcpyAsync" : (mode == 1 ? "kernel read, 256 workgroups" : "kernel read, 2048 workgroups"), bytes / ms / 1e6);
            }
        }
    }
    return 0;
}
