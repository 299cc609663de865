// poseidon_lat_dev.h -- the LATENCY forms of the Poseidon permutation: one hash across 16 lanes / across a quad of lanes.
//
// hash.hip uses them where a launch is a chain of dependent permutations with too few hashes to fill the machine (tree tops, leaves of
// short tables, small FRI layers); tools/ubench_perm_latency.hip times them alone.  Same function as poseidon_permute (poseidon_dev.h).
#pragma once
#include "poseidon_dev.h"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// ---- one permutation across 12 lanes of a 16-lane row (the smallest tree levels and matrices) ----
// The top levels of every tree hold too few nodes to fill the machine, so a launch costs one permutation's LATENCY: ~39 us for the
// one-lane-per-hash form on a wave alone on its SIMD (12k instructions), ~16 us for the four-lane form below, ~12 us here.  A hash
// owns a 16-lane row of the wave (lanes 0..11 = the twelve state words); a textbook round is constant add, x^7 (lane 0 only in the
// partial rounds) and the circulant MDS with the eleven rotated neighbours fetched by ds_bpermute (poseidon_stark.rs:65-95, 164-169,
// 239-251, 310-345).  Four hashes per wave: for launches of <= 1024 hashes (zkm_ctx::wide_max_hashes), where the chain is all there
// is.  Bit-exact with poseidon_permute.
// Partial rounds fused three at a time as in the other two forms (poseidon_dev.h poseidon_partial_group): every lane fetches the twelve
// words once (24 ds_bpermute instead of 66 for three textbook rounds), computes delta1 and delta2 for itself with the uniform rows 0 of
// M and M^2, and its own output word with row idx of M^3 -- twelve per-lane multipliers + M^2[idx][0], M[idx][0], read from a constant
// table on entry (four 16-byte loads, consumed four rounds later).  Rounds 24 and 25 stay textbook.
struct wide_tab_t { uint32_t v[16][16]; };
constexpr wide_tab_t make_wide_tab() {
    wide_tab_t t{};
    for (int i = 0; i < 12; i++) {
        for (int j = 0; j < 12; j++) t.v[i][j] = pc_host::ZKM_POSEIDON_M3[i][j];
        t.v[i][12] = pc_host::ZKM_POSEIDON_M2[i][0];
        t.v[i][13] = poseidon_m1(i, 0);
    }
    return t;
}
static __device__ __constant__ const wide_tab_t ZKM_WIDE_TAB = make_wide_tab();
__device__ __forceinline__ uint64_t poseidon_permute_wide(uint64_t x, unsigned lane) {
    const unsigned idx = lane & 15, base = lane & ~15u;
    const bool active = idx < 12;
    int src[12];
#pragma unroll
    for (int i = 1; i < 12; i++) src[i] = (int)(base + (idx + i) % 12);
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t diag = idx == 0 ? 8u : 0u;
    const gl_t* rcp = PC::ZKM_POSEIDON_RC + (active ? idx : 0);
    uint32_t m3[16];
    {
        const uint4* tp = reinterpret_cast<const uint4*>(&ZKM_WIDE_TAB.v[active ? idx : 0][0]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = tp[i];
            m3[4 * i] = v.x; m3[4 * i + 1] = v.y; m3[4 * i + 2] = v.z; m3[4 * i + 3] = v.w;
        }
    }
    auto textbook_linear = [&](uint64_t v, uint64_t k) {      // M v + k for this lane's word
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        uint64_t al = (uint64_t)(uint32_t)k + (uint64_t)lo * (C[0] + diag), ah = (k >> 32) + (uint64_t)hi * (C[0] + diag);
#pragma unroll
        for (int i = 1; i < 12; i++) {
            al += (uint64_t)(uint32_t)__shfl((int)lo, src[i]) * C[i];
            ah += (uint64_t)(uint32_t)__shfl((int)hi, src[i]) * C[i];
        }
        return poseidon_fold(al, ah);
    };
    x = gl_add_loose(x, rcp[0]);
#pragma unroll 1
    for (int r = 0; r < 8; r++) {                               // the eight full rounds; the partial rounds hang off round 3
        x = poseidon_sbox7(x);
        if (r == 3) {
#pragma unroll 1
            for (int g = 0; g < 7; g++) {                       // linear layers of rounds 3g+3 .. 3g+5, s-boxes of rounds 3g+4 .. 3g+6
                const uint64_t c1 = PC::ZKM_POSEIDON_FUSED_C1[g], c2 = PC::ZKM_POSEIDON_FUSED_C2[g];
                const uint64_t c3 = PC::ZKM_POSEIDON_FUSED_C3[g][active ? idx : 0];
                uint32_t lo[12], hi[12];
                {
                    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
#pragma unroll
                    for (int j = 0; j < 12; j++) {
                        lo[j] = (uint32_t)__shfl((int)xl, (int)(base + j));
                        hi[j] = (uint32_t)__shfl((int)xh, (int)(base + j));
                    }
                }
                uint64_t al = (uint32_t)c1, ah = c1 >> 32;
#pragma unroll
                for (int j = 0; j < 12; j++) {
                    al += (uint64_t)lo[j] * poseidon_m1(0, j);
                    ah += (uint64_t)hi[j] * poseidon_m1(0, j);
                }
                const uint64_t d1 = poseidon_sbox_delta(poseidon_fold(al, ah));
                const uint32_t d1l = (uint32_t)d1, d1h = (uint32_t)(d1 >> 32);
                al = (uint64_t)d1l * poseidon_m1(0, 0) + (uint32_t)c2;
                ah = (uint64_t)d1h * poseidon_m1(0, 0) + (c2 >> 32);
#pragma unroll
                for (int j = 0; j < 12; j++) {
                    al += (uint64_t)lo[j] * PC::ZKM_POSEIDON_M2[0][j];
                    ah += (uint64_t)hi[j] * PC::ZKM_POSEIDON_M2[0][j];
                }
                const uint64_t d2 = poseidon_sbox_delta(poseidon_fold(al, ah));
                const uint32_t d2l = (uint32_t)d2, d2h = (uint32_t)(d2 >> 32);
                al = (uint64_t)d1l * m3[12] + (uint64_t)d2l * m3[13] + (uint32_t)c3;
                ah = (uint64_t)d1h * m3[12] + (uint64_t)d2h * m3[13] + (c3 >> 32);
#pragma unroll
                for (int j = 0; j < 12; j++) {
                    al += (uint64_t)lo[j] * m3[j];
                    ah += (uint64_t)hi[j] * m3[j];
                }
                x = poseidon_fold(al, ah);
                const uint64_t y = poseidon_sbox7(x);
                x = idx == 0 ? y : x;
            }
            x = textbook_linear(x, rcp[25 * 12]);               // rounds 24 and 25, textbook
            const uint64_t y = poseidon_sbox7(x);
            x = idx == 0 ? y : x;
            x = textbook_linear(x, rcp[26 * 12]);
        } else {
            const int next = (r < 3 ? r : 22 + r) + 1;          // full round r < 3 is round r, r > 3 is round 22 + r
            x = textbook_linear(x, next < 30 ? rcp[next * 12] : 0);
        }
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
// The 22 partial rounds are FUSED three at a time as in the one-lane form (poseidon_dev.h, poseidon_partial_group): between two word-0
// s-boxes the state only passes linear maps, so with delta_k = sbox(a_k[0]) - a_k[0]
//     a1[0] = M[0,:] x + c1,   a2[0] = M^2[0,:] x + M[0,0] delta1 + c2,   a3 = M^3 x + M^2[:,0] delta1 + M[:,0] delta2 + c3
// -- two single-row products, which every lane of the quad computes for itself from the twelve words it has fetched anyway (no
// reduction across lanes, no broadcast), and ONE dense product for its three rows, instead of three dense products: ~350 instead of
// 495 instructions per three rounds (24 -> 15 us per permutation on a wave that has its SIMD to itself).  The entries of the M^2 / M^3
// rows in a lane's word order are 66 per-lane multipliers: kept in registers they cost the kernels their co-residency (200 VGPRs:
// eight contexts side by side lost 12 %), so they live in a 1.3 KB LDS table -- one image per lane position q, filled from constant
// memory at kernel start (quad_tab_load) and read with ds_read_b128 where a product needs them.  Rounds 24 and 25 stay textbook.
#define ZKM_QUAD_TAB_STRIDE 84          // dwords per lane position: 84 mod 32 = 20 -> the four images sit in different LDS banks
#define ZKM_QUAD_TAB_WORDS (4 * ZKM_QUAD_TAB_STRIDE)
struct quad_tab_t { uint32_t v[4][ZKM_QUAD_TAB_STRIDE]; };
// image of lane position Q: [0,12) row 0 of M, [12,24) row 0 of M^2 -- entry 3 k + b = multiplier of the word in slot b of lane
// (Q + k) mod 4, i.e. word ((Q + k) mod 4) + 4 b; [24,60) rows Q + 4 a of M^3, entry 24 + 12 a + 3 k + b; [60,63) M^2[Q + 4 a][0];
// [64,67) M[Q + 4 a][0]
constexpr quad_tab_t make_quad_tab() {
    quad_tab_t t{};
    for (int Q = 0; Q < 4; Q++)
        for (int k = 0; k < 4; k++)
            for (int b = 0; b < 3; b++) {
                const int j = ((Q + k) & 3) + 4 * b;
                t.v[Q][3 * k + b] = poseidon_m1(0, j);
                t.v[Q][12 + 3 * k + b] = pc_host::ZKM_POSEIDON_M2[0][j];
                for (int a = 0; a < 3; a++) t.v[Q][24 + 12 * a + 3 * k + b] = pc_host::ZKM_POSEIDON_M3[Q + 4 * a][j];
            }
    for (int Q = 0; Q < 4; Q++)
        for (int a = 0; a < 3; a++) {
            t.v[Q][60 + a] = pc_host::ZKM_POSEIDON_M2[Q + 4 * a][0];
            t.v[Q][64 + a] = poseidon_m1(Q + 4 * a, 0);
        }
    return t;
}
static __device__ __constant__ const quad_tab_t ZKM_QUAD_TAB = make_quad_tab();
// every thread of the workgroup calls this once, before the first poseidon_quad is built (ends with a workgroup barrier)
__device__ __forceinline__ void quad_tab_load(uint32_t* lds_tab /* [ZKM_QUAD_TAB_WORDS], 16-byte aligned */) {
    const uint32_t* src = &ZKM_QUAD_TAB.v[0][0];
    for (unsigned i = threadIdx.x; i < ZKM_QUAD_TAB_WORDS; i += blockDim.x) lds_tab[i] = src[i];
    __syncthreads();
}
struct poseidon_quad {
    uint32_t coef[4][3];
    uint32_t diag;    // the +8 on the (0, 0) entry: lane 0, output slot 0, own slot 0
    unsigned q;
    const uint32_t* t;   // this lane position's image of the table (LDS)
    __device__ __forceinline__ poseidon_quad(unsigned lane, const uint32_t* lds_tab) : q(lane & 3), t(lds_tab + (lane & 3) * ZKM_QUAD_TAB_STRIDE) {
        constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool wrap = q + k >= 4;
#pragma unroll
            for (int t2 = 0; t2 < 3; t2++) coef[k][t2] = wrap ? C[k + 4 * ((t2 + 2) % 3)] : C[k + 4 * t2];
        }
        diag = q == 0 ? 8u : 0u;
    }
    // twelve multipliers starting at dword `off` of the image (off a multiple of 4): three 16-byte LDS reads
    __device__ __forceinline__ void row(unsigned off, uint32_t (&m)[12]) const {
        const uint4* p = reinterpret_cast<const uint4*>(t + off);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint4 v = p[i];
            m[4 * i] = v.x; m[4 * i + 1] = v.y; m[4 * i + 2] = v.z; m[4 * i + 3] = v.w;
        }
    }
};
template <int CTRL>
__device__ __forceinline__ uint32_t quad_fetch(uint32_t v) {   // lane q reads lane (q + k) mod 4 of its quad: quad_perm [k, k+1, k+2, k+3]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// the twelve words of the hash as 32-bit halves: [k][b] = slot b of lane (q + k) mod 4
__device__ __forceinline__ void quad_gather(const uint64_t (&s)[3], uint32_t (&lo)[4][3], uint32_t (&hi)[4][3]) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        lo[0][b] = (uint32_t)s[b];
        hi[0][b] = (uint32_t)(s[b] >> 32);
        lo[1][b] = quad_fetch<0x39>(lo[0][b]); hi[1][b] = quad_fetch<0x39>(hi[0][b]);   // quad_perm [1, 2, 3, 0]
        lo[2][b] = quad_fetch<0x4E>(lo[0][b]); hi[2][b] = quad_fetch<0x4E>(hi[0][b]);   // quad_perm [2, 3, 0, 1]
        lo[3][b] = quad_fetch<0x93>(lo[0][b]); hi[3][b] = quad_fetch<0x93>(hi[0][b]);   // quad_perm [3, 0, 1, 2]
    }
}
// one textbook linear layer: s <- M s + kc (kc: this lane's pointer into a table of twelve constants, words q, q + 4, q + 8; or null)
__device__ __forceinline__ void quad_mds(uint64_t (&s)[3], const poseidon_quad& Q, const gl_t* kcp) {
    uint32_t lo[4][3], hi[4][3];
    quad_gather(s, lo, hi);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const uint64_t kc = kcp ? kcp[4 * a] : 0;           // the next round's constant rides in the accumulators
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
// three linear layers with the two word-0 s-boxes between them (see above); c3p: this lane's pointer into the group's twelve constants
__device__ __forceinline__ void quad_group3(uint64_t (&s)[3], const poseidon_quad& Q, uint64_t c1, uint64_t c2, const gl_t* c3p) {
    uint32_t lo[4][3], hi[4][3], m[12];
    quad_gather(s, lo, hi);
    Q.row(0, m);
    uint64_t al = (uint32_t)c1, ah = c1 >> 32;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            al += (uint64_t)lo[k][b] * m[3 * k + b];
            ah += (uint64_t)hi[k][b] * m[3 * k + b];
        }
    const uint64_t d1 = poseidon_sbox_delta(poseidon_fold(al, ah));
    const uint32_t d1l = (uint32_t)d1, d1h = (uint32_t)(d1 >> 32);
    Q.row(12, m);
    al = (uint64_t)d1l * poseidon_m1(0, 0) + (uint32_t)c2;
    ah = (uint64_t)d1h * poseidon_m1(0, 0) + (c2 >> 32);
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            al += (uint64_t)lo[k][b] * m[3 * k + b];
            ah += (uint64_t)hi[k][b] * m[3 * k + b];
        }
    const uint64_t d2 = poseidon_sbox_delta(poseidon_fold(al, ah));
    const uint32_t d2l = (uint32_t)d2, d2h = (uint32_t)(d2 >> 32);
    const uint4 c2m = *reinterpret_cast<const uint4*>(Q.t + 60), c1m = *reinterpret_cast<const uint4*>(Q.t + 64);
    const uint32_t m2c0[3] = {c2m.x, c2m.y, c2m.z}, m1c0[3] = {c1m.x, c1m.y, c1m.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        Q.row(24 + 12 * a, m);
        const uint64_t kc = c3p[4 * a];
        al = (uint64_t)d1l * m2c0[a] + (uint64_t)d2l * m1c0[a] + (uint32_t)kc;
        ah = (uint64_t)d1h * m2c0[a] + (uint64_t)d2h * m1c0[a] + (kc >> 32);
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                al += (uint64_t)lo[k][b] * m[3 * k + b];
                ah += (uint64_t)hi[k][b] * m[3 * k + b];
            }
        s[a] = poseidon_fold(al, ah);
    }
}
// s[a] = state word q + 4a (any uint64 representatives); out: canonical.  Every lane of the wave must call this (DPP reads neighbours).
__device__ __forceinline__ void poseidon_permute_quad(uint64_t (&s)[3], const poseidon_quad& Q) {
    const gl_t* rcp = PC::ZKM_POSEIDON_RC + Q.q;
#pragma unroll
    for (int a = 0; a < 3; a++) s[a] = gl_add_loose(s[a], rcp[4 * a]);
#pragma unroll 1
    for (int r = 0; r < 8; r++) {                               // the eight full rounds; the partial rounds hang off round 3
#pragma unroll
        for (int a = 0; a < 3; a++) s[a] = poseidon_sbox7(s[a]);
        if (r == 3) {
#pragma unroll 1
            for (int g = 0; g < 7; g++) {                       // linear layers of rounds 3g+3 .. 3g+5, s-boxes of rounds 3g+4 .. 3g+6
                quad_group3(s, Q, PC::ZKM_POSEIDON_FUSED_C1[g], PC::ZKM_POSEIDON_FUSED_C2[g], PC::ZKM_POSEIDON_FUSED_C3[g] + Q.q);
                const uint64_t y = poseidon_sbox7(s[0]);
                s[0] = Q.q == 0 ? y : s[0];
            }
            quad_mds(s, Q, rcp + 25 * 12);                      // rounds 24 and 25, textbook
            const uint64_t y = poseidon_sbox7(s[0]);
            s[0] = Q.q == 0 ? y : s[0];
            quad_mds(s, Q, rcp + 26 * 12);
        } else {
            const int next = (r < 3 ? r : 22 + r) + 1;          // full round r < 3 is round r, r > 3 is round 22 + r
            quad_mds(s, Q, next < 30 ? rcp + next * 12 : nullptr);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) s[a] = gl_canon(s[a]);
}

#endif
