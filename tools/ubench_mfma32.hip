// tools/ubench_mfma32.hip -- the dense 12x12 MDS layer of Poseidon on v_mfma_i32_32x32x32_i8 with a block-diagonal A operand:
// one instruction per byte position multiplies the states of all 64 lanes by the matrix (lane l holds rows (reg&3)+8(reg>>2)+4(l>>5)
// of column l&31 of D and the k-block l>>5 of B: with A = blockdiag(M, M) in that numbering every lane gets M x ITS OWN state).
// (1) layout probe, (2) bit-exactness against poseidon_mds_add, (3) timing of R full rounds (12 s-boxes + layer).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Izkm_amd/csrc -Iinclude tools/ubench_mfma32.hip -o tools/ubench_mfma32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "poseidon_dev.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_probe(const v4i* a, const v4i* b, int* out) {
    int l = threadIdx.x;
    v16i c = {0};
    v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], c, 0, 0, 0);
    for (int v = 0; v < 16; v++) out[16 * l + v] = d[v];
}

// A operand of this lane: lane l = (i = l & 31, h = l >> 5) holds row i of A for the k-block h.  Row i is owned by half (i >> 2) & 1 and
// is register reg = (i & 3) + 4 (i >> 3) of that half's D tuple; registers 0..11 are the state words.
__device__ __forceinline__ v4i mds_a_operand() {
    const int l = threadIdx.x & 63, i = l & 31, h = l >> 5;
    const int reg = (i & 3) + 4 * (i >> 3);
    const bool mine = (((i >> 2) & 1) == h) && reg < 12;
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    v4i a = {0, 0, 0, 0};
    for (int g = 0; g < 3; g++) {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) {
            const int j = 4 * g + k;
            uint32_t m = 0;
            for (int t = 0; t < 12; t++) m = ((j - reg + 12) % 12 == t) ? C[t] : m;   // C[(j - reg) mod 12] with a run-time reg
            if (reg == 0 && j == 0) m += 8;
            w |= m << (8 * k);
        }
        a[g] = mine ? (int)w : 0;
    }
    return a;
}

// the additive constants of a layer as the accumulators want them: half of the round constant + 128 rowsum 0x01010101 (that start makes
// each half the plain unsigned byte sum although the MFMA sees byte - 128); index 30 = "no constant"
namespace pc_cx {
#define ZKM_CONST static constexpr
#define ZKM_CONSTEXPR static constexpr
#include "poseidon_constants.inc"
#undef ZKM_CONST
#undef ZKM_CONSTEXPR
}
struct mdsc_t { uint64_t v[31][12][2]; };
constexpr mdsc_t poseidon_make_mdsc(int rowsum0 = 264) {
    mdsc_t t{};
    for (int r = 0; r < 31; r++)
        for (int w = 0; w < 12; w++) {
            const uint64_t c = r < 30 ? pc_cx::ZKM_POSEIDON_RC[r * 12 + w] : 0;
            const uint64_t off = (uint64_t)(w == 0 ? 128 * rowsum0 : 128 * 256) * 0x01010101ull;
            t.v[r][w][0] = (c & 0xFFFFFFFFull) + off;
            t.v[r][w][1] = (c >> 32) + off;
        }
    return t;
}
static __device__ __constant__ const mdsc_t ZKM_POSEIDON_MDSC = poseidon_make_mdsc();
static __device__ __constant__ const mdsc_t ZKM_POSEIDON_MDSC_CIRC = poseidon_make_mdsc(256);   // (the 8 s[0] of row 0 added outside the MFMA)

__device__ __forceinline__ void transpose4(const uint32_t in[4], uint32_t out[4]) {
    uint32_t t0 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400);  // (in0.b0, in1.b0, in0.b1, in1.b1)
    uint32_t t1 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602);  // (in0.b2, in1.b2, in0.b3, in1.b3)
    uint32_t t2 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400);
    uint32_t t3 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602);
    out[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100);
    out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302);
    out[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
    out[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
}

// out[r] = sum_i M[r][i] s[i] + add[r]: bytes b of the twelve words (minus 128: int8) are the B operand of MFMA b; D_b[r] is a signed
// digit of weight 2^(8b); the two halves are summed by multiply-adds from 128 rowsum 0x01010101 + the constant's half (that start
// makes each half the plain unsigned byte sum), then folded as in poseidon_mds_add.
__device__ __forceinline__ void poseidon_mds_add_mfma32(uint64_t s[12], const uint64_t (*cc)[2], const v4i A) {
    uint32_t T[3][8];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            lo[k] = (uint32_t)s[4 * g + k];
            hi[k] = (uint32_t)(s[4 * g + k] >> 32);
        }
        transpose4(lo, &T[g][0]);
        transpose4(hi, &T[g][4]);
    }
    int m8 = 1 << 8, m16 = 1 << 16, m24 = 1 << 24;
    asm("" : "+s"(m8));
    asm("" : "+s"(m16));
    asm("" : "+s"(m24));
    const v16i zero = {0};
    int64_t al[12], ah[12];
    // high halves first (byte positions 4..7), then the low ones; a word is folded as soon as its last digit has arrived.  An
    // accumulator starts in the multiply-add of its first digit, from the scalar pair of its constant: nothing is live before.
#pragma unroll
    for (int bb = 0; bb < 8; bb++) {
        const int b = bb ^ 4;   // 4, 5, 6, 7, 0, 1, 2, 3
        v4i B = {(int)(T[0][b] ^ 0x80808080u), (int)(T[1][b] ^ 0x80808080u), (int)(T[2][b] ^ 0x80808080u), 0};
        v16i D = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, zero, 0, 0, 0);
        if ((b & 3) == 0) asm volatile("s_nop 15" : "+v"(D));   // the first consumers are asm statements: nothing pads the MFMA's latency for them
        const int mult = (b & 3) == 1 ? m8 : (b & 3) == 2 ? m16 : m24;
#pragma unroll
        for (int r = 0; r < 12; r++) {
            if ((b & 3) == 0) {
                int64_t acc;
                uint64_t dummy;
                asm("v_mad_i64_i32 %0, %1, %2, 1, %3" : "=&v"(acc), "=&s"(dummy) : "v"(D[r]), "s"(cc[r][b >> 2]));
                if (b == 4) ah[r] = acc; else al[r] = acc;
            } else if (b > 4) ah[r] += (int64_t)D[r] * mult;
            else al[r] += (int64_t)D[r] * mult;
            if (b == 3) s[r] = poseidon_fold((uint64_t)al[r], (uint64_t)ah[r]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one accumulator tuple live at a time
    }
}

// The same layer WITHOUT the byte transposition: B = the halves of four words as they are (k slot = (word, byte)), A row (word, byte b0)
// picks byte b0 of each input word: one MFMA yields the four digits of four output words from four input words, three chained
// MFMAs (one per input group, C = the running tuple) finish them.  The matrix is circulant apart from the 8 on (0, 0): the A
// operand only depends on (input group - output group) mod 3, the 8 s[0] is added by two multiply-adds.
struct mds_raw_a { v4i d[3]; };
__device__ __forceinline__ mds_raw_a mds_raw_operands() {
    const int l = threadIdx.x & 63, i = l & 31, h = l >> 5;
    const int reg = (i & 3) + 4 * (i >> 3);          // register of this half's D tuple that row i is (16 per half, all used)
    const bool mine = ((i >> 2) & 1) == h;
    const int wl = reg >> 2, b0 = reg & 3;           // D[4 wl + b0] = digit b0 of output word wl of the group
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    mds_raw_a a;
    for (int d = 0; d < 3; d++)
        for (int il = 0; il < 4; il++) {             // B register il = input word il of the group; byte b of it = k slot (il, b)
            const int idx = (4 * d + il - wl + 12) % 12;
            uint32_t m = 0;
            for (int t = 0; t < 12; t++) m = idx == t ? C[t] : m;
            a.d[d][il] = mine ? (int)(m << (8 * b0)) : 0;
        }
    return a;
}
__device__ __forceinline__ void poseidon_mds_add_mfma_raw(uint64_t s[12], const uint64_t (*cc)[2], const mds_raw_a& A) {
    int m8 = 1 << 8, m16 = 1 << 16, m24 = 1 << 24, eight = 8;
    asm("" : "+s"(m8));
    asm("" : "+s"(m16));
    asm("" : "+s"(m24));
    asm("" : "+s"(eight));
    v4i X[2][3];   // [half][input group]
#pragma unroll
    for (int g = 0; g < 3; g++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            X[0][g][k] = (int)((uint32_t)s[4 * g + k] ^ 0x80808080u);
            X[1][g][k] = (int)((uint32_t)(s[4 * g + k] >> 32) ^ 0x80808080u);
        }
    const uint32_t s0l = (uint32_t)s[0], s0h = (uint32_t)(s[0] >> 32);
    const v16i zero = {0};
#pragma unroll
    for (int og = 0; og < 3; og++) {
        int64_t ah[4], al[4];
#pragma unroll
        for (int hh = 1; hh >= 0; hh--) {
            v16i D = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.d[(0 - og + 3) % 3], X[hh][0], zero, 0, 0, 0);
            D = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.d[(1 - og + 3) % 3], X[hh][1], D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.d[(2 - og + 3) % 3], X[hh][2], D, 0, 0, 0);
            asm volatile("s_nop 15" : "+v"(D));
#pragma unroll
            for (int w = 0; w < 4; w++) {
                int64_t acc;
                uint64_t dummy;
                asm("v_mad_i64_i32 %0, %1, %2, 1, %3" : "=&v"(acc), "=&s"(dummy) : "v"(D[4 * w]), "s"(cc[4 * og + w][hh]));
                acc += (int64_t)D[4 * w + 1] * m8;
                acc += (int64_t)D[4 * w + 2] * m16;
                acc += (int64_t)D[4 * w + 3] * m24;
                if (og == 0 && w == 0) acc += (int64_t)(uint64_t)((uint64_t)(hh ? s0h : s0l) * (uint32_t)eight);
                if (hh) ah[w] = acc; else al[w] = acc;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int w = 0; w < 4; w++) s[4 * og + w] = poseidon_fold((uint64_t)al[w], (uint64_t)ah[w]);
    }
}

__global__ void k_check(const uint64_t* in, uint64_t* out_ref, uint64_t* out_mfma, size_t n, int raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const v4i A = mds_a_operand();
    const mds_raw_a AR = mds_raw_operands();
    uint64_t a[12], b[12];
    for (int k = 0; k < 12; k++) a[k] = b[k] = in[k * n + i];
    if (raw) {
        poseidon_mds_add<true>(a, &PC::ZKM_POSEIDON_RC[24]);
        poseidon_mds_add_mfma_raw(b, ZKM_POSEIDON_MDSC_CIRC.v[2], AR);
        for (int k = 0; k < 12; k++) {
            out_ref[k * n + i] = gl_canon(a[k]);
            out_mfma[k * n + i] = gl_canon(b[k]);
        }
        return;
    }
    poseidon_mds_add<true>(a, &PC::ZKM_POSEIDON_RC[12]);
    poseidon_mds_add_mfma32(b, ZKM_POSEIDON_MDSC.v[1], A);
    for (int k = 0; k < 12; k++) {
        out_ref[k * n + i] = gl_canon(a[k]);
        out_mfma[k * n + i] = gl_canon(b[k]);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rounds(uint64_t* data, size_t n, int rounds) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const v4i A = mds_a_operand();
    const mds_raw_a AR = mds_raw_operands();
    uint64_t s[12];
    for (int k = 0; k < 12; k++) s[k] = data[k * n + i];
#pragma unroll 1
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = poseidon_sbox7(s[k]);
        if (MODE == 4) poseidon_mds_add_mfma_raw(s, ZKM_POSEIDON_MDSC_CIRC.v[r & 7], AR);
        if (MODE == 0) poseidon_mds_add<true>(s, &PC::ZKM_POSEIDON_RC[12 * (r & 7)]);
        else if (MODE == 1) poseidon_mds_add_mfma32(s, ZKM_POSEIDON_MDSC.v[r & 7], A);
        // MODE 2: s-boxes only
    }
    for (int k = 0; k < 12; k++) data[k * n + i] = s[k];
}

// the multiply-add layer at the MFMA kernel's occupancy (four waves per SIMD): what the registers cost on their own
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rounds_occ4(uint64_t* data, size_t n, int rounds) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[12];
    for (int k = 0; k < 12; k++) s[k] = data[k * n + i];
#pragma unroll 1
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = poseidon_sbox7(s[k]);
        poseidon_mds_add<true>(s, &PC::ZKM_POSEIDON_RC[12 * (r & 7)]);
    }
    for (int k = 0; k < 12; k++) data[k * n + i] = s[k];
}

static uint64_t rnd64() { return ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    // ---- (1) layout probe: D[i][n] = sum over (half, byte) of A(lane i + 32 half, byte) B(lane n + 32 half, byte)?
    {
        std::vector<int> a(256), b(256), out(1024);
        for (int l = 0; l < 256; l++) { a[l] = (int)rnd64(); b[l] = (int)rnd64(); }
        int *da, *db, *dout;
        CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dout, 4096));
        CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, (const v4i*)da, (const v4i*)db, dout);
        CK(hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost));
        auto byte_of = [](const std::vector<int>& v, int lane, int j) { return (int)(int8_t)(v[4 * lane + j / 4] >> (8 * (j % 4))); };
        int ok = 1;
        for (int l = 0; l < 64; l++)
            for (int reg = 0; reg < 16; reg++) {
                const int n = l & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
                int e = 0;
                for (int h = 0; h < 2; h++)
                    for (int j = 0; j < 16; j++) e += byte_of(a, i + 32 * h, j) * byte_of(b, n + 32 * h, j);
                if (out[16 * l + reg] != e) ok = 0;
            }
        printf("layout: D[(reg&3)+8(reg>>2)+4(l>>5)][l&31] = sum_{h,j} A(lane i+32h, byte j) B(lane n+32h, byte j): %s\n", ok ? "YES" : "no");
    }
    // ---- (2) MDS check
    const size_t n = 1 << 16;
    std::vector<uint64_t> h(12 * n);
    for (auto& x : h) x = rnd64();
    for (int k = 0; k < 12; k++) { h[k * n + 0] = ~0ull; h[k * n + 1] = 0; h[k * n + 2] = 0xFFFFFFFF00000000ull; h[k * n + 3] = 0x8080808080808080ull; h[k * n + 4] = 0x7F7F7F7F7F7F7F7Full; }
    uint64_t *din, *d1, *d2;
    CK(hipMalloc(&din, 96 * n)); CK(hipMalloc(&d1, 96 * n)); CK(hipMalloc(&d2, 96 * n));
    CK(hipMemcpy(din, h.data(), 96 * n, hipMemcpyHostToDevice));
    std::vector<uint64_t> r1(12 * n), r2(12 * n);
    for (int raw = 0; raw < 2; raw++) {
        hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, din, d1, d2, n, raw);
        CK(hipMemcpy(r1.data(), d1, 96 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), d2, 96 * n, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < 12 * n; i++) bad += r1[i] != r2[i];
        printf("MDS layer via 32x32x32 MFMA (%s) vs multiply-add form on %zu states: %zu mismatching words\n", raw ? "untransposed B" : "byte planes", n, bad);
    }
    // ---- (3) timing
    const size_t N = 1 << 22;
    uint64_t* dd; CK(hipMalloc(&dd, 96 * N)); CK(hipMemset(dd, 5, 96 * N));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 64;
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            if (mode == 1) hipLaunchKernelGGL(k_rounds<1>, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            if (mode == 2) hipLaunchKernelGGL(k_rounds<2>, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            if (mode == 4) hipLaunchKernelGGL(k_rounds<4>, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            if (mode == 3) hipLaunchKernelGGL(k_rounds_occ4, dim3(N / 256), dim3(256), 0, 0, dd, N, R);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s: %d rounds x %zu states: %.3f ms (%.1f G rounds/s)\n",
               mode == 0 ? "multiply-add MDS" : mode == 1 ? "MFMA 32x32x32 MDS" : mode == 2 ? "s-boxes only" : mode == 3 ? "multiply-add MDS at 4 waves/SIMD" : "MFMA MDS, untransposed B", R, N, best, (double)R * N / best / 1e6);
    }
    return 0;
}
