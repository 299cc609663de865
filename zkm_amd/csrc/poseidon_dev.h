// poseidon_dev.h -- Goldilocks Poseidon permutation (width 12, 4 + 22 + 4 rounds, x^7), device + host.
//
// Same function as the reference's in-tree permutation prover/src/poseidon/poseidon_stark.rs:51-95
// (constant_layer :164-169, sbox_monomial :239-251, mds_layer :310-345), which is also plonky2
// PoseidonHash's permutation (Merkle hasher, Challenger).  One permutation per lane: the 12-word state
// lives in 24 VGPRs, round constants are wave-uniform and come from the scalar (constant) cache.
// No MFMA: this is 64-bit modular integer work.
//
// Formulation.  On gfx950 a 32x32->64 multiply-add (v_mad_u64_u32) issues at nearly the rate of an add
// (profiles/r01_ubench_valu_rates.txt), so the cost of a round is its instruction count, and a full
// 64x64 modular multiply (4 mads + ~15 carry/reduction instructions) is ~25x the price of a multiply by
// a 6-bit MDS entry.  The reference's "fast partial round" form (sparse matrices with 64-bit entries,
// poseidon_stark.rs:463-487) therefore LOSES here: every partial round is done in the textbook form
// -- s-box on lane 0, dense circulant MDS with 6-bit entries -- and the next round's constants are
// folded into the MDS accumulators before the single reduction.  Algebraically identical, so outputs are
// bit-exact (pinned by the plonky2 test vectors and by naive == fast in the oracle).
//
// Partial rounds, fused three at a time.  Between two lane-0 s-boxes the state only goes through linear maps, and the
// integer powers of the MDS matrix stay small: the entries of M^3 are < 2^21 and its row sums < 2^25, so a row of M^3
// applied to 32-bit halves still fits the same pair of 64-bit accumulators (24 multiply-adds) as a row of M.  With
// delta_k = sbox(a_k[0]) - a_k[0] the three rounds
//     a1 = M x + K1,   a2 = M (a1 + e0 delta1) + K2,   a3 = M (a2 + e0 delta2) + K3
// become  a1[0] = M[0,:] x + c1,   a2[0] = M^2[0,:] x + M[0,0] delta1 + c2,
//         a3 = M^3 x + M^2[:,0] delta1 + M[:,0] delta2 + c3          (c1, c2, c3 precomputed from the round constants)
// i.e. two single-row products and ONE dense product instead of three dense products: 23 linear layers (round 3's
// MDS and the 22 partial rounds) cost 8 dense products.  Tables: tools/gen_poseidon_constants.py derive_fused (which
// also asserts the no-overflow bound).  Algebraically identical to the reference's rounds, so outputs are bit-exact.
//
// The state is kept "loose" (any uint64 representing its residue) between rounds and canonicalised once
// on exit -- see gl_dev.h for the overflow arguments of every loose primitive.
#pragma once
#include "gl_dev.h"

#undef ZKM_CONSTEXPR
#define ZKM_CONSTEXPR static constexpr
namespace pc_host {
#undef ZKM_CONST
#define ZKM_CONST static const
#include "poseidon_constants.inc"
ZKM_CONST uint64_t ZKM_POSEIDON_ZERO12[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // "constants" after the last round
}  // namespace pc_host
#if defined(__HIPCC__)
namespace pc_dev {
#undef ZKM_CONST
#define ZKM_CONST static __device__ __constant__ const
#include "poseidon_constants.inc"
ZKM_CONST uint64_t ZKM_POSEIDON_ZERO12[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
}  // namespace pc_dev
#endif
#undef ZKM_CONST
#undef ZKM_CONSTEXPR

#if defined(__HIP_DEVICE_COMPILE__)
#define PC pc_dev
#define POSEIDON_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// region markers for tools/isa_histogram.py (comments in the assembly; no instructions): the permutation's control flow has parts
// that run 8, 7 and 1 times per permutation, which loop trip counts alone cannot express
#define POSEIDON_REGION(name) asm volatile("; ZKM_REGION " name)
#else
#define PC pc_host
#define POSEIDON_SCHED_FENCE() ((void)0)
#define POSEIDON_REGION(name) ((void)0)
#endif

// Hide a small multiplier from the optimiser (it stays in an SGPR).  Left to itself the compiler turns the MDS entries that are
// powers of two (2, 8, 16) into v_lshl_add_u64, which needs the 32-bit word zero-extended into a register pair first (a v_mov per
// term) and issues no faster than the v_mad_u64_u32 that takes the word directly.
#if defined(__HIP_DEVICE_COMPILE__)
#define POSEIDON_OPAQUE(c) asm("" : "+s"(c))
// A table pointer the optimiser cannot see through and cannot move: the constants behind it are (re)loaded with scalar loads where
// they are used.  Without it the loop-invariant ones (first-round constants, the last fused group's) are hoisted out of the sponge
// loop as ~100 SGPRs, spilled to VGPR lanes and read back with v_readlane -- 150 vector instructions per permutation for values the
// scalar cache delivers for free.
#define POSEIDON_OPAQUE_PTR(p) asm volatile("" : "+s"(p))
#else
#define POSEIDON_OPAQUE(c) ((void)0)
#define POSEIDON_OPAQUE_PTR(p) ((void)0)
#endif

// x^7 on a loose value -> loose
GL_HD uint64_t poseidon_sbox7(uint64_t x) {
    uint64_t x2 = gl_mul_loose(x, x);
    uint64_t x4 = gl_mul_loose(x2, x2);
    uint64_t x3 = gl_mul_loose(x, x2);
    return gl_mul_loose(x3, x4);
}

// al + 2^32 ah -> loose, for al, ah < 2^59.  No compares: the carries are produced as VALUES.
//   2^32 ah = 2^64 ah_hi + 2^32 ah_lo == EPS ah_hi + 2^32 ah_lo;   s1 = al + EPS ah_hi < 2^60 (one multiply-add);
//   hs = s1_hi + ah_lo is a 33-bit number whose top bit c again weighs 2^64 == EPS;  result = (hs_lo32 : s1_lo) + EPS c,
//   which cannot wrap (c = 1 implies hs_lo32 < 2^28).
// Additive constants are folded in by starting the accumulators at (const_lo, const_hi).
GL_HD uint64_t poseidon_fold(uint64_t al, uint64_t ah) {
#if defined(__HIP_DEVICE_COMPILE__)
    // t = al + EPS ah_hi < 2^60; adding 2^32 ah_lo only touches the high word, and its carry-out (an SGPR pair) selects the EPS
    // correction: multiply-add, add, select, add.  t < 2^60 means the corrected sum cannot wrap again.
    const uint64_t t = (uint64_t)(uint32_t)(ah >> 32) * 0xFFFFFFFFu + al;
    uint32_t rhi, wrap;
    uint64_t carry;
    asm("v_add_co_u32_e64 %0, %1, %3, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, %1"
        : "=&v"(rhi), "=&s"(carry), "=v"(wrap)
        : "v"((uint32_t)(t >> 32)), "v"((uint32_t)ah));
    return (((uint64_t)rhi << 32) | (uint32_t)t) + wrap;
#endif
    uint64_t s1 = (uint64_t)(uint32_t)(ah >> 32) * 0xFFFFFFFFu + al;
    uint64_t hs = (uint64_t)(uint32_t)ah + (s1 >> 32);
    uint64_t base = (hs << 32) | (uint32_t)s1;
    return (uint64_t)(uint32_t)(hs >> 32) * 0xFFFFFFFFu + base;
}

// Circulant MDS (first row CIRC = {17,15,41,16,2,28,13,13,39,18,34,20}, plus 8 on the (0,0) entry):
//   out[r] = sum_i CIRC[i] * s[(i+r)%12] + [r==0] 8 s[0] + add[r]
// Each state word is split into 32-bit halves, the halves are accumulated in two 64-bit sums that start at the halves of
// the additive constant (next round's constant) and stay < 2^32 * 265 < 2^41, and the pair is folded once (poseidon_fold).
// Inputs loose, outputs loose.
template <bool ADD, int ROW0 = 0, int ROW1 = 12>
GL_HD void poseidon_mds_add(uint64_t s[12], const uint64_t* add) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
    uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20}, diag = 8;
#pragma unroll
    for (int i = 0; i < 12; i++) POSEIDON_OPAQUE(C[i]);
    POSEIDON_OPAQUE(diag);
#pragma unroll
    for (int r = ROW0; r < ROW1; r++) {   // (rows outside [ROW0, ROW1) keep their input word: callers treat them as dead)
        uint64_t al = ADD ? (uint64_t)(uint32_t)add[r] : 0, ah = ADD ? add[r] >> 32 : 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)lo[(i + r) % 12] * C[i];
            ah += (uint64_t)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) {
            al += (uint64_t)lo[0] * diag;
            ah += (uint64_t)hi[0] * diag;
        }
        s[r] = poseidon_fold(al, ah);
        if ((r & 3) == 3) POSEIDON_SCHED_FENCE();
    }
}
GL_HD void poseidon_mds(uint64_t s[12]) { poseidon_mds_add<false>(s, nullptr); }

// sbox(a) - a mod p, loose -> loose
GL_HD uint64_t poseidon_sbox_delta(uint64_t a) { return gl_sub_rr(poseidon_sbox7(a), a); }
GL_HD constexpr uint32_t poseidon_m1(int i, int j) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return C[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
}

// T linear layers (T = 3 or 2) with the T - 1 lane-0 s-boxes between them; c3 = constants after the last layer.
template <int T>
GL_HD void poseidon_partial_group(uint64_t s[12], uint64_t c1, uint64_t c2, const uint64_t* c3) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
    uint64_t al = (uint32_t)c1, ah = c1 >> 32;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        al += (uint64_t)lo[j] * poseidon_m1(0, j);
        ah += (uint64_t)hi[j] * poseidon_m1(0, j);
    }
    const uint64_t d1 = poseidon_sbox_delta(poseidon_fold(al, ah));
    const uint32_t d1l = (uint32_t)d1, d1h = (uint32_t)(d1 >> 32);
    uint32_t d2l = 0, d2h = 0;
    POSEIDON_SCHED_FENCE();
    if (T == 3) {
        al = (uint64_t)d1l * poseidon_m1(0, 0) + (uint32_t)c2;
        ah = (uint64_t)d1h * poseidon_m1(0, 0) + (c2 >> 32);
#pragma unroll
        for (int j = 0; j < 12; j++) {
            al += (uint64_t)lo[j] * PC::ZKM_POSEIDON_M2[0][j];
            ah += (uint64_t)hi[j] * PC::ZKM_POSEIDON_M2[0][j];
        }
        const uint64_t d2 = poseidon_sbox_delta(poseidon_fold(al, ah));
        d2l = (uint32_t)d2;
        d2h = (uint32_t)(d2 >> 32);
        POSEIDON_SCHED_FENCE();
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        if (T == 3) {
            al = (uint64_t)d1l * PC::ZKM_POSEIDON_M2[i][0] + (uint64_t)d2l * poseidon_m1(i, 0) + (uint32_t)c3[i];
            ah = (uint64_t)d1h * PC::ZKM_POSEIDON_M2[i][0] + (uint64_t)d2h * poseidon_m1(i, 0) + (c3[i] >> 32);
#pragma unroll
            for (int j = 0; j < 12; j++) {
                al += (uint64_t)lo[j] * PC::ZKM_POSEIDON_M3[i][j];
                ah += (uint64_t)hi[j] * PC::ZKM_POSEIDON_M3[i][j];
            }
        } else {
            al = (uint64_t)d1l * poseidon_m1(i, 0) + (uint32_t)c3[i];
            ah = (uint64_t)d1h * poseidon_m1(i, 0) + (c3[i] >> 32);
#pragma unroll
            for (int j = 0; j < 12; j++) {
                al += (uint64_t)lo[j] * PC::ZKM_POSEIDON_M2[i][j];
                ah += (uint64_t)hi[j] * PC::ZKM_POSEIDON_M2[i][j];
            }
        }
        s[i] = poseidon_fold(al, ah);
        if ((i & 1) == 1) POSEIDON_SCHED_FENCE();
    }
}

// One loop over the eight full rounds with the partial-round section hanging off round 3: every piece of round code exists
// once, so the hot loop of the leaf kernel (33 absorb steps per row) is ~30 KB of instructions instead of ~57 KB (the second
// block of full rounds, two inline s-box layers and the last MDS used to be separate copies) and stays inside the 64 KB
// instruction cache two CUs share.
//
// What the caller reads of the result.  A sponge in overwrite mode (plonky2 hash_n_to_m_no_pad: the next input chunk REPLACES the rate
// words) only carries the four capacity words from one permutation to the next, and a digest is the first four words: the last MDS
// layer then needs 4 of its 12 rows (8 x (24 multiply-adds + fold) less) and only what leaves the sponge is canonicalised.
//   ALL: twelve canonical words.   CAPACITY: words 8..11, loose (they go straight into the next permutation); words 0..7 are garbage.
//   DIGEST: words 0..3, canonical; words 4..11 are garbage.
enum { POSEIDON_OUT_ALL = 0, POSEIDON_OUT_CAPACITY = 1, POSEIDON_OUT_DIGEST = 2 };

// The MDS layer of a full round: s <- M s + constants of round `next` (30 = none).  poseidon_mfma_dev.h has the matrix-core form.
struct poseidon_mds_valu {
    GL_HD void layer(uint64_t s[12], int next) const {
        poseidon_mds_add<true>(s, next < 30 ? &PC::ZKM_POSEIDON_RC[next * 12] : PC::ZKM_POSEIDON_ZERO12);
    }
};

// In: any uint64 words (loose).  `out` must be wave-uniform on the device (it selects code, not lanes).
template <class MDS>
GL_HD void poseidon_permute_out_t(uint64_t s[12], int out, const MDS& mds) {
    POSEIDON_REGION("entry");
    const uint64_t* rc0 = PC::ZKM_POSEIDON_RC;
    POSEIDON_OPAQUE_PTR(rc0);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_lc(s[i], rc0[i]);   // loose + CANONICAL round constant (tests/test_oracle_primitives.py
                                                                   // checks every table entry < p): one correction, not two (gl_dev.h)
#pragma unroll 1
    for (int r = 0; r < 8; r++) {
        POSEIDON_REGION("full_sbox");
#pragma unroll
        for (int i = 0; i < 12; i++) {
            s[i] = poseidon_sbox7(s[i]);
            if ((i & 1) == 1) POSEIDON_SCHED_FENCE();
        }
        if (r == 3) {
            POSEIDON_REGION("partial_head");
            // round 3's MDS opens the first fused group; the groups cover the MDS of rounds 3 .. 25 and the s-boxes of rounds 4 .. 25
#pragma unroll 1
            for (int g = 0; g < 7; g++) {  // MDS of rounds 3g+3 .. 3g+5 and the s-boxes of rounds 3g+4, 3g+5; then round 3g+6's
                POSEIDON_REGION("partial_group");
                poseidon_partial_group<3>(s, PC::ZKM_POSEIDON_FUSED_C1[g], PC::ZKM_POSEIDON_FUSED_C2[g], PC::ZKM_POSEIDON_FUSED_C3[g]);
                s[0] = poseidon_sbox7(s[0]);
            }
            POSEIDON_REGION("partial_tail");
            const uint64_t* fc1 = PC::ZKM_POSEIDON_FUSED_C1;
            const uint64_t* fc3 = PC::ZKM_POSEIDON_FUSED_C3[7];
            POSEIDON_OPAQUE_PTR(fc1);
            POSEIDON_OPAQUE_PTR(fc3);
            poseidon_partial_group<2>(s, fc1[7], 0, fc3);  // MDS of rounds 24, 25
        } else if (r == 7 && out != POSEIDON_OUT_ALL) {
            POSEIDON_REGION("last_mds_rows");
            if (out == POSEIDON_OUT_CAPACITY) poseidon_mds_add<false, 8, 12>(s, nullptr);
            else poseidon_mds_add<false, 0, 4>(s, nullptr);
        } else {
            POSEIDON_REGION("full_mds");
            // full round r < 3 is round r, r > 3 is round 22 + r; the constants added are those of the NEXT round (none after the last)
            const int next = (r < 3 ? r : 22 + r) + 1;
            mds.layer(s, next);
        }
    }
    POSEIDON_REGION("exit");
    if (out == POSEIDON_OUT_ALL) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
    } else if (out == POSEIDON_OUT_DIGEST) {
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = gl_canon(s[i]);
    }
}

GL_HD void poseidon_permute_out(uint64_t s[12], int out) { poseidon_permute_out_t(s, out, poseidon_mds_valu{}); }

// In: any uint64 words (loose).  Out: canonical.
GL_HD void poseidon_permute(uint64_t s[12]) { poseidon_permute_out(s, POSEIDON_OUT_ALL); }
