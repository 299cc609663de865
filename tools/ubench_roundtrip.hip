// tools/ubench_roundtrip.hip -- what does a transcript round trip cost, and what would it cost without the download kernel?
//   (a) producer kernel -> one-workgroup copy kernel into pinned memory + flag (what zkm_ctx::download does)
//   (b) producer kernel writes its (small) result into pinned memory itself -> hipStreamWriteValue64(flag) -- no second kernel
//   (c) producer kernel writes result AND flag itself (single workgroup: legal only when one workgroup produces the result)
// hipcc -O3 --offload-arch=gfx950 tools/ubench_roundtrip.hip -o tools/ubench_roundtrip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)
__global__ void k_produce(uint64_t* out, uint64_t v) { if (threadIdx.x < 64) out[threadIdx.x] = v + threadIdx.x; }
__global__ void k_download(const uint64_t* src, uint64_t* dst, uint64_t* flag, uint64_t seq) {
    if (threadIdx.x < 64) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_produce_flag(uint64_t* out, uint64_t v, uint64_t* flag, uint64_t seq) {
    if (threadIdx.x < 64) out[threadIdx.x] = v + threadIdx.x;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
int main() {
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint64_t *d, *h, *flag;
    CHECK(hipMalloc(&d, 512));
    CHECK(hipHostMalloc((void**)&h, 1024, hipHostMallocCoherent));
    flag = h + 64; *flag = 0;
    const int N = 2000;
    for (int mode = 0; mode < 3; mode++) {
        uint64_t seq = (uint64_t)mode << 32;
        double t0 = 0; bool ok = true;
        for (int i = -100; i < N; i++) {
            if (i == 0) t0 = now();
            seq++;
            if (mode == 0) {
                hipLaunchKernelGGL(k_produce, dim3(1), dim3(256), 0, st, d, seq);
                hipLaunchKernelGGL(k_download, dim3(1), dim3(256), 0, st, d, h, flag, seq);
            } else if (mode == 1) {
                hipLaunchKernelGGL(k_produce, dim3(1), dim3(256), 0, st, h, seq);
                hipError_t e = hipStreamWriteValue64(st, flag, seq, 0);
                if (e != hipSuccess) { printf("hipStreamWriteValue64: %s\n", hipGetErrorString(e)); ok = false; break; }
            } else {
                hipLaunchKernelGGL(k_produce_flag, dim3(1), dim3(256), 0, st, h, seq, flag, seq);
            }
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
            if (h[5] != seq + 5) { printf("mode %d: stale data at iteration %d\n", mode, i); ok = false; break; }
        }
        if (ok) printf("mode %d: %.2f us per round trip\n", mode, (now() - t0) / N);
    }
    return 0;
}
