// host_poseidon.hip -- the Poseidon permutation of the HOST-side Challenger (plonky2 Challenger<F, PoseidonHash>).
//
// A twelve-table segment pushes ~25k field elements through the transcript (the openings of every column of every table at two
// points, 36 caps, the FRI layers): ~3000 permutations per segment on one host thread, with the GPU waiting for the challenge that
// comes out.  The device formulation (poseidon_dev.h: dense fused partial rounds, tuned for v_mad_u64_u32) is the wrong shape for a
// CPU: here the partial rounds use the equivalent sparse factorisation (one dot product and one rank-1 update per round; the
// FAST_* tables of poseidon_constants.inc, generated from the public parameters), all sums of products are accumulated in 128 bits
// and reduced once, and the full rounds' circulant MDS runs on 32-bit halves in loops the compiler vectorises (AVX2 clone selected
// at load time).  Same function as poseidon_permute -- tests/test_abi.py compares both with each other and with the oracle.
#include <stdint.h>
#include <string.h>

#include "poseidon_dev.h"

#if !defined(__HIP_DEVICE_COMPILE__)
namespace {
using u128 = unsigned __int128;
constexpr uint64_t EPS = 0xFFFFFFFFull;                  // 2^64 mod p
constexpr uint64_t TWO128 = 0xFFFFFFFE00000001ull;       // 2^128 mod p = p - 2^32

// 128 bits -> a 64-bit representative (not necessarily canonical): 2^64 == EPS, 2^96 == -1
inline uint64_t red128(u128 x) {
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t h0 = hi & EPS, h1 = hi >> 32;
    // branch-free: the carry of t0 + t1 is a coin flip on field data, a mispredicted branch costs more than the whole reduction
    uint64_t t0 = lo - h1;
    if (lo < h1) t0 -= EPS;                               // (after the borrow t0 > 2^64 - 2^32: no second one)
    const uint64_t t1 = h0 * EPS;
    uint64_t r = t0 + t1;
    if (r < t1) r += EPS;                                 // (after the carry r < 2^64 - 2^33: no second one)
    return r;
}
inline uint64_t mulr(uint64_t a, uint64_t b) { return red128((u128)a * b); }
// representative + canonical constant
inline uint64_t addc(uint64_t a, uint64_t k) {
    uint64_t s = a + k;
    s += (0 - (uint64_t)(s < a)) & EPS;                   // a + k < 2^64 + p: after the wrap s < p, s + EPS < 2^64
    return s;
}
inline uint64_t sbox7(uint64_t x) {
    const uint64_t x2 = mulr(x, x), x4 = mulr(x2, x2), x3 = mulr(x2, x);
    return mulr(x3, x4);
}
// sum of up to 12 products of 64-bit words: 128-bit accumulator + overflow count, reduced once
struct acc192 {
    u128 a = 0;
    unsigned over = 0;
    inline void mac(uint64_t x, uint64_t y) {
        const u128 p = (u128)x * y;
        a += p;
        over += a < p;
    }
    inline uint64_t reduce() const {
        uint64_t r = red128(a);
        if (over) r = red128((u128)r + (u128)over * TWO128);   // r < 2^64, over <= 12: fits
        return r;
    }
};

// circulant MDS (+ 8 on the diagonal entry (0, 0)) on 32-bit halves: out_r = sum_j C[(j - r) mod 12] s_j.  REXT[12 + r - j] is that
// coefficient, contiguous in r: the inner loops are plain multiply-adds of 32-bit values into 64-bit lanes.
struct rext_table {
    alignas(64) uint64_t v[24];
    rext_table() {                                        // REXT[k] = C[(12 - k) mod 12], twice: from the one parameter table
        for (int k = 0; k < 24; k++) v[k] = pc_host::ZKM_POSEIDON_MDS_CIRC[(24 - k) % 12];
    }
};
const rext_table REXT_TABLE;
const uint64_t* const REXT = REXT_TABLE.v;
__attribute__((target_clones("avx2", "default"))) void mds_layer(uint64_t s[12]) {
    alignas(64) uint64_t al[12], ah[12];
    for (int r = 0; r < 12; r++) { al[r] = 0; ah[r] = 0; }
    for (int j = 0; j < 12; j++) {
        const uint64_t lo = (uint32_t)s[j], hi = s[j] >> 32;
        const uint64_t* c = REXT + 12 - j;
        for (int r = 0; r < 12; r++) {
            al[r] += c[r] * lo;
            ah[r] += c[r] * hi;
        }
    }
    al[0] += pc_host::ZKM_POSEIDON_MDS_DIAG[0] * (uint64_t)(uint32_t)s[0];   // (the only non-zero diagonal entry: 8)
    ah[0] += pc_host::ZKM_POSEIDON_MDS_DIAG[0] * (s[0] >> 32);
    for (int r = 0; r < 12; r++) s[r] = red128((u128)al[r] + ((u128)ah[r] << 32));       // both < 2^42
}

void full_round(uint64_t s[12], const uint64_t* rc) {
    for (int i = 0; i < 12; i++) s[i] = sbox7(addc(s[i], rc[i]));
    mds_layer(s);
}
}  // namespace

void zkm_host_poseidon_permute(uint64_t s[12]) {
    namespace K = pc_host;
    for (int r = 0; r < 4; r++) full_round(s, K::ZKM_POSEIDON_RC + 12 * r);
    // partial rounds, sparse form: constants moved in front, the first dense layer pulled out (11 x 11 on words 1 .. 11)
    for (int i = 0; i < 12; i++) s[i] = addc(s[i], K::ZKM_POSEIDON_FAST_FIRST_RC[i]);
    {
        uint64_t t[12];
        t[0] = s[0];
        for (int c = 1; c < 12; c++) {
            acc192 a;
            for (int r = 1; r < 12; r++) a.mac(s[r], K::ZKM_POSEIDON_FAST_INIT[r - 1][c - 1]);
            t[c] = a.reduce();
        }
        memcpy(s, t, sizeof t);
    }
    const uint64_t m00 = K::ZKM_POSEIDON_MDS_CIRC[0] + K::ZKM_POSEIDON_MDS_DIAG[0];
    for (int r = 0; r < 22; r++) {
        const uint64_t s0 = addc(sbox7(s[0]), K::ZKM_POSEIDON_FAST_RC[r]);
        acc192 d;
        d.mac(s0, m00);
        for (int i = 1; i < 12; i++) d.mac(s[i], K::ZKM_POSEIDON_FAST_W_HATS[r][i - 1]);
        for (int i = 1; i < 12; i++) s[i] = red128((u128)s0 * K::ZKM_POSEIDON_FAST_VS[r][i - 1] + s[i]);   // < 2^128
        s[0] = d.reduce();
    }
    for (int r = 0; r < 4; r++) full_round(s, K::ZKM_POSEIDON_RC + 12 * (26 + r));
    for (int i = 0; i < 12; i++) s[i] = s[i] >= GL_P ? s[i] - GL_P : s[i];
}

// the device formulation compiled for the host: the cross-check of the function above (tests only)
void zkm_host_poseidon_permute_reference(uint64_t st[12]) { poseidon_permute(st); }
#endif
