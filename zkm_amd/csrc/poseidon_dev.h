// poseidon_dev.h -- Goldilocks Poseidon permutation (width 12, 4 + 22 + 4 rounds, x^7), device + host.
//
// Same function as the reference's in-tree permutation prover/src/poseidon/poseidon_stark.rs:51-95
// (constant_layer :164-169, sbox_monomial :239-251, mds_layer :310-345, mds_partial_layer_init :392-404,
// mds_partial_layer_fast :463-487), which is also plonky2 PoseidonHash's permutation (Merkle hasher,
// Challenger).  One permutation per lane: the 12-word state lives in 24 VGPRs, round constants are
// wave-uniform and come from the scalar (constant) cache.  No MFMA: this is 64-bit modular integer work.
//
// The state is kept "loose" (any uint64 representing its residue) between rounds and canonicalised once
// on exit -- see gl_dev.h for the overflow arguments of every loose primitive.
#pragma once
#include "gl_dev.h"

namespace pc_host {
#undef ZKM_CONST
#define ZKM_CONST static const
#include "poseidon_constants.inc"
}  // namespace pc_host
#if defined(__HIPCC__)
namespace pc_dev {
#undef ZKM_CONST
#define ZKM_CONST static __device__ __constant__ const
#include "poseidon_constants.inc"
}  // namespace pc_dev
#endif
#undef ZKM_CONST

#if defined(__HIP_DEVICE_COMPILE__)
#define PC pc_dev
#else
#define PC pc_host
#endif

// x^7 on a loose value -> loose
GL_HD uint64_t poseidon_sbox7(uint64_t x) {
    uint64_t x2 = gl_mul_loose(x, x);
    uint64_t x4 = gl_mul_loose(x2, x2);
    uint64_t x3 = gl_mul_loose(x, x2);
    return gl_mul_loose(x3, x4);
}

// Circulant MDS (first row CIRC = {17,15,41,16,2,28,13,13,39,18,34,20}, plus 8 on the (0,0) entry).
// Each output is sum_i c_i * s[(i+r)%12] with c_i < 2^6: split every state word into 32-bit halves,
// accumulate the halves in two 64-bit sums (each < 2^32 * 264 < 2^41), recombine to a 73-bit value and
// reduce once.  Inputs loose, outputs loose.
GL_HD void poseidon_mds(uint64_t s[12]) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)lo[(i + r) % 12] * C[i];
            ah += (uint64_t)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) {
            al += (uint64_t)lo[0] * 8u;
            ah += (uint64_t)hi[0] * 8u;
        }
        // value = al + ah * 2^32  (< 2^74): low 64 bits and the carry-out word
        uint64_t low = al + (ah << 32);
        uint64_t high = (ah >> 32) + (low < al ? 1 : 0);
        s[r] = gl_reduce128(low, high);
    }
}

GL_HD void poseidon_full_round(uint64_t s[12], int round) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_loose(s[i], PC::ZKM_POSEIDON_RC[round * 12 + i]);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = poseidon_sbox7(s[i]);
    poseidon_mds(s);
}

GL_HD void poseidon_partial_rounds(uint64_t s[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_loose(s[i], PC::ZKM_POSEIDON_FAST_FIRST_RC[i]);
    {
        uint64_t t[12];
        t[0] = s[0];
#pragma unroll
        for (int c = 1; c < 12; c++) {
            uint64_t acc = 0;
#pragma unroll
            for (int r = 1; r < 12; r++) acc = gl_add_loose(acc, gl_mul_loose(s[r], PC::ZKM_POSEIDON_FAST_INIT[r - 1][c - 1]));
            t[c] = acc;
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        uint64_t s0 = gl_add_loose(poseidon_sbox7(s[0]), PC::ZKM_POSEIDON_FAST_RC[r]);
        uint64_t d = gl_mul_loose(s0, 25);
#pragma unroll
        for (int i = 1; i < 12; i++) d = gl_add_loose(d, gl_mul_loose(s[i], PC::ZKM_POSEIDON_FAST_W_HATS[r][i - 1]));
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = gl_add_loose(s[i], gl_mul_loose(s0, PC::ZKM_POSEIDON_FAST_VS[r][i - 1]));
        s[0] = d;
    }
}

// In: any uint64 words (loose).  Out: canonical.
GL_HD void poseidon_permute(uint64_t s[12]) {
#pragma unroll 1
    for (int r = 0; r < 4; r++) poseidon_full_round(s, r);
    poseidon_partial_rounds(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) poseidon_full_round(s, 26 + r);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}
