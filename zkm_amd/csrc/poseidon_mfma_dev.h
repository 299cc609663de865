// poseidon_mfma_dev.h -- the dense 12x12 MDS layer of the FULL rounds on the matrix core (device only, leaf hashing).
//
// v_mfma_i32_32x32x32_i8 computes D[i][n] = sum_k A[i][k] B[k][n] with lane l holding column n = l & 31 of B for the k-block l >> 5
// (16 bytes) and, of D, column n = l & 31, rows (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), reg = 0..15.  With A = blockdiag(M, M) written in
// THAT row / k numbering, every lane receives M times the 16 bytes it supplied itself: one instruction multiplies one byte position
// of the twelve state words of all 64 lanes by the matrix -- no cross-lane traffic, the permutation stays one hash per lane.
// (Layout probed and the layer checked bit for bit against poseidon_mds_add by tools/ubench_mfma32.hip.)
//
//   per layer:  48 v_perm (4x4 byte transposes: byte b of words 4g..4g+3 -> one register) + 24 v_xor (byte - 128: the operands are int8)
//               8 MFMAs (one per byte position; the matrix pipe, not the vector ALU)
//               per row 5 shift-adds / adds on 32-bit partial sums, 2 multiply-adds and a 6-instruction fold
//   instead of 288 multiply-adds + the fold (poseidon_mds_add).  Round 5's form (96 multiply-adds into two 64-bit accumulators per row, all 24
//   byte planes up front) needed 126 registers = four waves per SIMD; round 6's keeps 32-bit partial sums, transposes a half at a time and
//   parks the high-half sums in LDS while the low half is worked on: 93 registers, five waves (profiles/r06_mfma_lean.txt).
//
// MFMA ignores EXEC: a kernel that uses this keeps all lanes of its waves alive (no early return for the tail; clamp the index instead).
// The fused partial-round layers (M^3: three int8 digit planes) and the 4-row last layer stay on the vector ALU (poseidon_dev.h).
#pragma once
#include "poseidon_dev.h"

// ZKM_MFMA_PARK: the high-half sums of the layer wait in LDS (24 words per lane, stride ZKM_MFMA_PARK_STRIDE = the workgroup size) while the
// low half is worked on: the kernel then fits the 96 registers of five waves per SIMD (0: everything in registers, 110 of them, four waves)
#ifndef ZKM_MFMA_PARK
#define ZKM_MFMA_PARK 1
#endif
#define ZKM_MFMA_PARK_STRIDE 256
typedef int zkm_v4i __attribute__((ext_vector_type(4)));
typedef int zkm_v16i __attribute__((ext_vector_type(16)));
#if !defined(__HIP_DEVICE_COMPILE__)
// host pass of a .hip file: the names exist (kernels are parsed for the host too), nothing runs
struct poseidon_mds_mfma {
    zkm_v4i A;
    uint32_t* park = nullptr;
    GL_HD void layer(uint64_t s[12], int next) const { poseidon_mds_valu{}.layer(s, next); }
};
GL_HD zkm_v4i poseidon_mfma_operand() { return zkm_v4i{0, 0, 0, 0}; }
#else

namespace pc_cx {   // the round constants once more, as constant expressions
#define ZKM_CONST static constexpr
#define ZKM_CONSTEXPR static constexpr
#include "poseidon_constants.inc"
#undef ZKM_CONST
#undef ZKM_CONSTEXPR
}  // namespace pc_cx
// A operand of this lane: lane l = (i = l & 31, h = l >> 5) holds row i of A for the k-block h.  Row i belongs to half (i >> 2) & 1 and
// is register reg = (i & 3) + 4 (i >> 3) of that half's D tuple; registers 0..11 are the state words, 12..15 unused.
__device__ __forceinline__ zkm_v4i poseidon_mfma_operand() {
    const int l = threadIdx.x & 63, i = l & 31, h = l >> 5;
    const int reg = (i & 3) + 4 * (i >> 3);
    const bool mine = (((i >> 2) & 1) == h) && reg < 12;
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    zkm_v4i a = {0, 0, 0, 0};
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

// in[k] = one 32-bit half of word k; out[b] = (in[0].byte b, in[1].byte b, in[2].byte b, in[3].byte b)
__device__ __forceinline__ void poseidon_transpose4(const uint32_t in[4], uint32_t out[4]) {
    // v_perm_b32(hi, lo, sel): byte i of the result = byte sel_i of the 8-byte value {hi, lo} (0..3 = lo, 4..7 = hi)
    const uint32_t t0 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400);  // (in0.b0, in1.b0, in0.b1, in1.b1)
    const uint32_t t1 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602);  // (in0.b2, in1.b2, in0.b3, in1.b3)
    const uint32_t t2 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400);
    const uint32_t t3 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602);
    out[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100);
    out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302);
    out[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
    out[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
}

// s <- M s + (constants of round `next`; 30 = none).  In: loose words.  Out: loose words.
// ONE 64-bit sum T and ONE 32-bit sum Y per row, built from 32-bit pieces:
//     M s + c  ==  T + 2^48 Y,      T = t0 + X + 2^16 Z + 2^32 U,   X = D0 + 2^8 D1,  Z = D2 + 2^8 D3,  U = D4 + 2^8 D5,   Y = y0 + D6 + 2^8 D7
// with D_b the matrix core's result for byte position b (a signed 18-bit number per row).  t0 / y0 hold the round constant in three pieces
// (low word, bits 32..47 at weight 2^32, bits 48..63 in Y) and the 128 rowsum offsets of their byte positions, so every partial sum is a
// plain non-negative number: T < 2^57, Y < 2^27.  The fold: 2^48 Y = 2^48 (Y mod 2^16) + 2^64 (Y >> 16) == 2^48 (Y mod 2^16) + EPS (Y >> 16).
struct poseidon_mdsc2_t { uint64_t t0[31][12]; uint32_t y0[31][12]; };
constexpr poseidon_mdsc2_t poseidon_make_mdsc2() {
    poseidon_mdsc2_t t{};
    for (int r = 0; r < 31; r++)
        for (int w = 0; w < 12; w++) {
            const uint64_t c = r < 30 ? pc_cx::ZKM_POSEIDON_RC[r * 12 + w] : 0;
            const uint64_t off = (uint64_t)(w == 0 ? 128 * 264 : 128 * 256);   // 128 x rowsum: 256, + 8 on row 0
            t.t0[r][w] = (c & 0xFFFFFFFFull) + off * 0x01010101ull + ((((c >> 32) & 0xFFFFull) + off * 0x0101ull) << 32);
            t.y0[r][w] = (uint32_t)((c >> 48) + off * 0x0101ull);
        }
    return t;
}
static __device__ __constant__ const poseidon_mdsc2_t ZKM_POSEIDON_MDSC2 = poseidon_make_mdsc2();

// T + 2^48 Y -> loose, for T < 2^57, Y < 2^27 (carry of the high-word add as a VALUE, like poseidon_fold)
__device__ __forceinline__ uint64_t poseidon_fold_ty(uint64_t T, uint32_t Y) {
    const uint64_t t = (uint64_t)(Y >> 16) * 0xFFFFFFFFu + T;             // < 2^57 + 2^43: cannot wrap
    uint32_t rhi, wrap;
    uint64_t carry;
    asm("v_add_co_u32_e64 %0, %1, %3, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, %1"
        : "=&v"(rhi), "=&s"(carry), "=v"(wrap)
        : "v"((uint32_t)(t >> 32)), "v"(Y << 16));
    // a carry means rhi wrapped below 2^26: adding EPS cannot wrap again
    return (((uint64_t)rhi << 32) | (uint32_t)t) + wrap;
}

// (park: 24 words of LDS per lane, stride ZKM_MFMA_PARK_STRIDE -- the high-half sums U and Y wait there while the low half is worked on, so
// that the layer fits the 96 registers of five waves per SIMD; LDS instructions are not vector-ALU instructions)
template <bool PARK>
__device__ __forceinline__ void poseidon_mds_add_mfma(uint64_t s[12], int next, const zkm_v4i A, uint32_t* park) {
    const uint64_t* t0 = ZKM_POSEIDON_MDSC2.t0[next];
    const uint32_t* y0 = ZKM_POSEIDON_MDSC2.y0[next];
    const zkm_v16i zero = {0};
    int m16 = 1 << 16;
    POSEIDON_OPAQUE(m16);
    // every partial sum is a 32-bit register until a row is finished: U = D4 + 2^8 D5 and Y = y0 + D6 + 2^8 D7 from the high halves first
    // (the low halves of the state wait in their registers), then X = D0 + 2^8 D1 and Z = D2 + 2^8 D3, and row by row
    // T = t0 + X + 2^16 Z + 2^32 U, fold(T, Y)
    uint32_t U[12], Y[12];
    {
        uint32_t P[3][4];
#pragma unroll
        for (int g = 0; g < 3; g++) {
            uint32_t hi[4];
#pragma unroll
            for (int k = 0; k < 4; k++) hi[k] = (uint32_t)(s[4 * g + k] >> 32);
            poseidon_transpose4(hi, P[g]);
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const zkm_v4i B = {(int)(P[0][b] ^ 0x80808080u), (int)(P[1][b] ^ 0x80808080u), (int)(P[2][b] ^ 0x80808080u), 0};
            const zkm_v16i D = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 12; r++) {
                if (b == 0) U[r] = (uint32_t)D[r];
                else if (b == 1) U[r] += (uint32_t)D[r] << 8;
                else if (b == 2) Y[r] = y0[r] + (uint32_t)D[r];
                else Y[r] += (uint32_t)D[r] << 8;
            }
            POSEIDON_SCHED_FENCE();   // one result tuple live at a time
        }
    }
    if constexpr (PARK) {
#pragma unroll
        for (int r = 0; r < 12; r++) {
            park[r * ZKM_MFMA_PARK_STRIDE] = U[r];
            park[(12 + r) * ZKM_MFMA_PARK_STRIDE] = Y[r];
        }
        POSEIDON_SCHED_FENCE();
    }
    {
        uint32_t P[3][4];
#pragma unroll
        for (int g = 0; g < 3; g++) {
            uint32_t lo[4];
#pragma unroll
            for (int k = 0; k < 4; k++) lo[k] = (uint32_t)s[4 * g + k];
            poseidon_transpose4(lo, P[g]);
        }
        uint32_t X[12], Z[12];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const zkm_v4i B = {(int)(P[0][b] ^ 0x80808080u), (int)(P[1][b] ^ 0x80808080u), (int)(P[2][b] ^ 0x80808080u), 0};
            const zkm_v16i D = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 12; r++) {
                if (b == 0) X[r] = (uint32_t)D[r];
                else if (b == 1) X[r] += (uint32_t)D[r] << 8;
                else if (b == 2) Z[r] = (uint32_t)D[r];
                else {
                    Z[r] += (uint32_t)D[r] << 8;
                    // (t0 + X as ONE multiply-add with the inline constant 1 and the scalar pair of t0: a VALU instruction of this ISA reads
                    // one scalar operand, so with the multiplier in an SGPR as well the compiler copies t0 into a register pair first)
                    int64_t acc;
                    uint64_t unused;
                    asm("v_mad_i64_i32 %0, %1, %2, 1, %3" : "=&v"(acc), "=&s"(unused) : "v"(X[r]), "s"(t0[r]));
                    acc += (int64_t)(int32_t)Z[r] * m16;
                    const uint32_t u = PARK ? park[r * ZKM_MFMA_PARK_STRIDE] : U[r], y = PARK ? park[(12 + r) * ZKM_MFMA_PARK_STRIDE] : Y[r];
                    const uint64_t T = ((uint64_t)((uint32_t)((uint64_t)acc >> 32) + u) << 32) | (uint32_t)acc;
                    s[r] = poseidon_fold_ty(T, y);
                    if ((r & 1) == 1) POSEIDON_SCHED_FENCE();   // (two rows' temporaries at a time)
                }
            }
            POSEIDON_SCHED_FENCE();
        }
    }
}

struct poseidon_mds_mfma {
    zkm_v4i A;
    uint32_t* park = nullptr;
    __device__ __forceinline__ void layer(uint64_t s[12], int next) const {
        poseidon_mds_add_mfma<(ZKM_MFMA_PARK != 0)>(s, next, A, park);
    }
};
#endif
