// tools/ubench_strided.hip -- HBM rate of the NTT's strided-tile access pattern (no arithmetic): every workgroup reads a tile of
// 2^S rows x T contiguous elements (row stride 2^(L-S) elements, 8 B each) into registers and writes it back in place.
// Decides the pass plan of ntt.hip: which (S, T, bytes per lane) shapes reach the streaming rate.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_strided.hip -o tools/ubench_strided && tools/ubench_strided
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// NT threads; tile = 2^S x 2^LT elements; VEC elements (8 B each) per lane access; each thread moves E = tile / NT elements
template <int S, int LT, int VEC, int NT>
__global__ __launch_bounds__(NT) void k_tile(uint64_t* data, int L, size_t col_stride) {
    constexpr int T = 1 << LT, R = 1 << S, LPR = T / VEC, RPS = NT / LPR, STEPS = R / RPS;
    static_assert(LPR >= 1 && RPS >= 1 && STEPS >= 1, "shape");
    const int m = L - S;
    const size_t tiles_per_col = ((size_t)1 << m) >> LT;
    const size_t col = blockIdx.x / tiles_per_col, t = blockIdx.x % tiles_per_col;
    uint64_t* base = data + col * col_stride + t * T;
    const int lane_in_row = threadIdx.x % LPR, row0 = threadIdx.x / LPR;
    typedef uint64_t vec_t __attribute__((ext_vector_type(VEC)));
    vec_t v[STEPS];
#pragma unroll
    for (int k = 0; k < STEPS; k++) v[k] = *reinterpret_cast<const vec_t*>(base + ((size_t)(row0 + k * RPS) << m) + lane_in_row * VEC);
#pragma unroll
    for (int k = 0; k < STEPS; k++) v[k] += 1;
#pragma unroll
    for (int k = 0; k < STEPS; k++) *reinterpret_cast<vec_t*>(base + ((size_t)(row0 + k * RPS) << m) + lane_in_row * VEC) = v[k];
}

// contiguous tile of 2^S elements x 1 (the last pass): plain streaming copy in place
template <int VEC>
__global__ __launch_bounds__(256) void k_stream(uint64_t* data, size_t n) {
    typedef uint64_t vec_t __attribute__((ext_vector_type(VEC)));
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC * 4;
    vec_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const vec_t*>(data + i + k * VEC);
#pragma unroll
    for (int k = 0; k < 4; k++) *reinterpret_cast<vec_t*>(data + i + k * VEC) = v[k] + 1;
    (void)n;
}

template <int S, int LT, int VEC, int NT>
static void run(uint64_t* d, int L, int ncols) {
    size_t n = (size_t)1 << L;
    size_t blocks = ((n >> S) >> LT) * ncols;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_tile<S, LT, VEC, NT>), dim3(blocks), dim3(NT), 0, 0, d, L, n); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_tile<S, LT, VEC, NT>), dim3(blocks), dim3(NT), 0, 0, d, L, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double gb = 2.0 * n * ncols * 8 / 1e9;
    printf("L=%d rows=2^%-2d seg=%4d B  %2d B/lane  %4d thr  %2d el/thr : %7.3f ms  %7.1f GB/s (read+write)\n", L, S, (1 << LT) * 8, VEC * 8, NT,
           (1 << (S + LT)) / NT, ms, gb / (ms * 1e-3));
}

int main() {
    const int L = 22, ncols = 64;
    size_t n = (size_t)1 << L;
    uint64_t* d; hipMalloc(&d, n * ncols * 8); hipMemset(d, 1, n * ncols * 8);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        size_t blocks = n * ncols / (256 * 2 * 4);
        hipLaunchKernelGGL(k_stream<2>, dim3(blocks), dim3(256), 0, 0, d, n); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_stream<2>, dim3(blocks), dim3(256), 0, 0, d, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("streaming copy in place, 16 B/lane: %7.3f ms  %7.1f GB/s (read+write)\n", ms, 2.0 * n * ncols * 8 / 1e9 / (ms * 1e-3));
    }
    // today's shapes: 2048-element tiles, 8 B per lane, 256 threads
    run<7, 4, 1, 256>(d, L, ncols);   // S=7, 128 B segments
    run<7, 5, 1, 512>(d, L, ncols);   // S=7, 256 B segments (r01 plan)
    run<8, 4, 1, 512>(d, L, ncols);   // S=8, 128 B
    // 16 B per lane
    run<7, 5, 2, 256>(d, L, ncols);
    run<8, 5, 2, 512>(d, L, ncols);
    run<8, 6, 2, 512>(d, L, ncols);
    // large tiles for a two-pass plan: S = 9..11
    run<9, 5, 1, 1024>(d, L, ncols);  // 16K elements, 256 B segments, 16 el/thread
    run<9, 5, 2, 512>(d, L, ncols);   // 32 el/thread
    run<9, 5, 2, 1024>(d, L, ncols);
    run<9, 4, 2, 512>(d, L, ncols);   // 8K elements, 128 B segments
    run<9, 4, 2, 256>(d, L, ncols);
    run<9, 6, 2, 1024>(d, L, ncols);  // 32K elements, 512 B segments, 32 el/thread
    run<10, 4, 2, 512>(d, L, ncols);  // 16K elements, 128 B segments
    run<10, 4, 2, 1024>(d, L, ncols);
    run<10, 5, 2, 1024>(d, L, ncols); // 32K elements
    run<11, 3, 1, 1024>(d, L, ncols); // 16K elements, 64 B segments
    run<11, 4, 2, 1024>(d, L, ncols); // 32K elements, 128 B
    run<8, 5, 2, 256>(d, L, ncols);   // 8K elements, 256 B, 32 el/thr
    run<8, 5, 1, 1024>(d, L, ncols);
    return 0;
}
