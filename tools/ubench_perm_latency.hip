// tools/ubench_perm_latency.hip -- latency of ONE Poseidon permutation in a chain of dependent ones, per form, at a given number of
// waves on the GPU (the regime of tree tops, leaves of short wide tables and small FRI layers: about one wave per SIMD).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Izkm_amd/csrc tools/ubench_perm_latency.hip -o tools/ubench_perm_latency && tools/ubench_perm_latency
// Every form is first checked against poseidon_permute (one lane per hash) on random states.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "poseidon_lat_dev.h"

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// hash h starts from state word i = mix(h * 12 + i + 1); `steps` permutations back to back; out = the first four words
template <int FORM>
__global__ __launch_bounds__(256) void k_chain(uint64_t* out, size_t nhash, int steps) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (FORM == 0) {                                   // one lane per hash
        if (t >= nhash) return;
        uint64_t s[12];
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = mix(t * 12 + i + 1);
        for (int k = 0; k < steps; k++) poseidon_permute_out(s, POSEIDON_OUT_ALL);
#pragma unroll
        for (int i = 0; i < 4; i++) out[4 * t + i] = s[i];
    } else if (FORM == 1) {                            // a quad per hash: lane q holds words q, q + 4, q + 8
        const size_t h = t >> 2;
        const unsigned q = t & 3;
        __shared__ __attribute__((aligned(16))) uint32_t qtab[ZKM_QUAD_TAB_WORDS];
        quad_tab_load(qtab);
        const poseidon_quad Q(threadIdx.x, qtab);
        uint64_t s[3];
#pragma unroll
        for (int a = 0; a < 3; a++) s[a] = mix(h * 12 + q + 4 * a + 1);
        for (int k = 0; k < steps; k++) {
            poseidon_permute_quad(s, Q);
        }
        if (h < nhash) out[4 * h + q] = s[0];
    } else {                                           // 16 lanes per hash
        const size_t h = t >> 4;
        const unsigned lane = threadIdx.x & 63, idx = lane & 15;
        uint64_t x = idx < 12 ? mix(h * 12 + idx + 1) : 0;
        for (int k = 0; k < steps; k++) x = poseidon_permute_wide(x, lane);
        if (h < nhash && idx < 4) out[4 * h + idx] = x;
    }
}

template <int FORM>
static float run(uint64_t* d_out, size_t nhash, int steps, int lanes_per_hash) {
    const size_t threads = nhash * lanes_per_hash;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_chain<FORM>, dim3((threads + 255) / 256), dim3(256), 0, 0, d_out, nhash, steps);   // warm-up
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_chain<FORM>, dim3((threads + 255) / 256), dim3(256), 0, 0, d_out, nhash, steps);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    const size_t NMAX = 1 << 18;
    uint64_t* d;
    CHECK(hipMalloc(&d, NMAX * 4 * 8));
    std::vector<uint64_t> ref(4 * 4096), got(4 * 4096);
    run<0>(d, 4096, 3, 1);
    CHECK(hipMemcpy(ref.data(), d, ref.size() * 8, hipMemcpyDeviceToHost));
    const char* names[3] = {"one lane per hash", "quad of lanes per hash", "16 lanes per hash"};
    for (int f = 1; f < 3; f++) {
        if (f == 1) run<1>(d, 4096, 3, 4);
        if (f == 2) run<3>(d, 4096, 3, 16);
        CHECK(hipMemcpy(got.data(), d, got.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); i++) bad += got[i] != ref[i];
        printf("check %-24s vs one-lane form on 4096 hashes x 3 permutations: %zu differing words\n", names[f], bad);
    }
    const int steps = 64;
    printf("\nus per permutation in a chain of %d (the launch's time / %d), by waves on the GPU (1024 SIMDs):\n", steps, steps);
    printf("%-10s %18s %24s %18s\n", "waves", names[0], names[1], names[2]);
    for (size_t waves : {256, 512, 1024, 2048, 4096}) {
        const float t0 = run<0>(d, waves * 64, steps, 1), t1 = run<1>(d, waves * 16, steps, 4), t3 = run<3>(d, waves * 4, steps, 16);
        printf("%-10zu %18.2f %24.2f %18.2f\n", waves, t0 * 1e3 / steps, t1 * 1e3 / steps, t3 * 1e3 / steps);
    }
    printf("(hashes per wave: 64 / 16 / 4)\n");
    return 0;
}
