// poseidon_lat_dev.h -- the LATENCY forms of the Poseidon permutation: one hash across 16 lanes / across a quad of lanes.
//
// hash.hip uses them where a launch is a chain of dependent permutations with too few hashes to fill the machine (tree tops, leaves of
// short tables, small FRI layers); tools/ubench_perm_latency.hip times them alone.  Same function as poseidon_permute (poseidon_dev.h).
#pragma once
#include "poseidon_dev.h"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// ---- one permutation across 12 lanes of a 16-lane row (the smallest tree levels and matrices) ----
// The top levels of every tree hold too few nodes to fill the machine, so a launch costs one permutation's LATENCY: ~40 us for the
// one-lane-per-hash form on a wave alone on its SIMD (12k instructions), ~24 us for the four-lane form below (5.7k), ~13 us here
// (3.2k).  A hash owns a 16-lane row of the wave (lanes 0..11 = the twelve state words): every round is constant add, x^7 (lane 0
// only in the partial rounds), and the circulant MDS with the twelve rotated neighbours fetched by ds_bpermute -- the textbook rounds
// (poseidon_stark.rs:65-95, 164-169, 239-251, 310-345).  800 wave instructions per hash (four-lane form 360, one lane 190): for
// launches of <= 1024 hashes (zkm_ctx::wide_max_hashes), where the chain is all there is.  Bit-exact with poseidon_permute (the
// fused partial rounds there are an algebraic regrouping).
__device__ __forceinline__ uint64_t poseidon_permute_wide(uint64_t x, unsigned lane) {
    const unsigned idx = lane & 15, base = lane & ~15u;
    const bool active = idx < 12;
    int src[12];
#pragma unroll
    for (int i = 1; i < 12; i++) src[i] = (int)(base + (idx + i) % 12);
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t diag = idx == 0 ? 8u : 0u;
    const gl_t* rcp = PC::ZKM_POSEIDON_RC + (active ? idx : 0);
    x = gl_add_loose(x, rcp[0]);
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        const uint64_t y = poseidon_sbox7(x);
        x = (full || idx == 0) ? y : x;
        const uint64_t k = r + 1 < 30 ? rcp[(r + 1) * 12] : 0;
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        uint64_t al = (uint64_t)(uint32_t)k + (uint64_t)lo * (C[0] + diag), ah = (k >> 32) + (uint64_t)hi * (C[0] + diag);
#pragma unroll
        for (int i = 1; i < 12; i++) {
            al += (uint64_t)(uint32_t)__shfl((int)lo, src[i]) * C[i];
            ah += (uint64_t)(uint32_t)__shfl((int)hi, src[i]) * C[i];
        }
        x = poseidon_fold(al, ah);
    }
    return gl_canon(x);
}


// ---- one permutation across FOUR lanes (short matrices, small tree levels) ----
// Where a launch holds too few hashes to fill the machine it costs a permutation's LATENCY (~27 us for the one-lane-per-hash form on a
// wave that has its SIMD to itself: 12k dependent-ish instructions).  Here a hash owns a quad of lanes; lane q holds the state words
// q, q + 4, q + 8.  Every round is the textbook one (poseidon_stark.rs:65-95, 164-169, 239-251, 310-345): constant add (folded into
// the accumulators of the previous linear layer), x^7 on the lane's three words (word 0 only in the partial rounds), and the circulant
// MDS out[r] = sum_j C[(j - r) mod 12] s[j] with the nine foreign words fetched by DPP quad permutes (v_mov_b32 dpp: no LDS, no
// ds_bpermute).  For output word r = q + 4a and the input word in slot b of lane (q + k) mod 4 the coefficient is
// C[k + 4 ((b - a + 2 [q + k >= 4]) mod 3)]: it depends on (k, (b - a) mod 3) and the lane -- twelve per-lane multipliers set up once.
// ~5.3k instructions per permutation and wave (16 hashes): the latency of the 16-lane form this replaces (rounds 1-2; 5.7k, four
// hashes per wave, ds_bpermute) at a QUARTER of its issue slots -- 330 wave instructions per hash against 1430, 190 for the
// one-lane form.  Bit-exact with poseidon_permute (the fused partial rounds there are an algebraic regrouping).
struct poseidon_quad {
    uint32_t coef[4][3];
    uint32_t diag;    // the +8 on the (0, 0) entry: lane 0, output slot 0, own slot 0
    unsigned q;
    __device__ __forceinline__ explicit poseidon_quad(unsigned lane) : q(lane & 3) {
        constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool wrap = q + k >= 4;
#pragma unroll
            for (int t = 0; t < 3; t++) coef[k][t] = wrap ? C[k + 4 * ((t + 2) % 3)] : C[k + 4 * t];
        }
        diag = q == 0 ? 8u : 0u;
    }
};
template <int CTRL>
__device__ __forceinline__ uint32_t quad_fetch(uint32_t v) {   // lane q reads lane (q + k) mod 4 of its quad: quad_perm [k, k+1, k+2, k+3]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// s[a] = state word q + 4a (any uint64 representatives); out: canonical.  Every lane of the wave must call this (DPP reads neighbours).
__device__ __forceinline__ void poseidon_permute_quad(uint64_t (&s)[3], const poseidon_quad& Q) {
    const gl_t* rcp = PC::ZKM_POSEIDON_RC + Q.q;
#pragma unroll
    for (int a = 0; a < 3; a++) s[a] = gl_add_loose(s[a], rcp[4 * a]);
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;                 // (uniform)
        {
            const uint64_t y = poseidon_sbox7(s[0]);
            s[0] = (full || Q.q == 0) ? y : s[0];
        }
        if (full) {
            s[1] = poseidon_sbox7(s[1]);
            s[2] = poseidon_sbox7(s[2]);
        }
        uint32_t lo[4][3], hi[4][3];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            lo[0][b] = (uint32_t)s[b];
            hi[0][b] = (uint32_t)(s[b] >> 32);
            lo[1][b] = quad_fetch<0x39>(lo[0][b]); hi[1][b] = quad_fetch<0x39>(hi[0][b]);   // quad_perm [1, 2, 3, 0]
            lo[2][b] = quad_fetch<0x4E>(lo[0][b]); hi[2][b] = quad_fetch<0x4E>(hi[0][b]);   // quad_perm [2, 3, 0, 1]
            lo[3][b] = quad_fetch<0x93>(lo[0][b]); hi[3][b] = quad_fetch<0x93>(hi[0][b]);   // quad_perm [3, 0, 1, 2]
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const uint64_t kc = r + 1 < 30 ? rcp[(r + 1) * 12 + 4 * a] : 0;   // the next round's constant rides in the accumulators
            uint64_t al = (uint32_t)kc, ah = kc >> 32;
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const uint32_t m = Q.coef[k][(b - a + 3) % 3];
                    al += (uint64_t)lo[k][b] * m;
                    ah += (uint64_t)hi[k][b] * m;
                }
            if (a == 0) {
                al += (uint64_t)lo[0][0] * Q.diag;
                ah += (uint64_t)hi[0][0] * Q.diag;
            }
            s[a] = poseidon_fold(al, ah);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) s[a] = gl_canon(s[a]);
}

#endif
