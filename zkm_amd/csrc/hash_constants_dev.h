// hash_constants_dev.h -- Keccak-f[1600] and SHA-256 constants and round functions shared by the hashing, witness and
// constraint kernels (device only).  Keccak: FIPS 202 §3.2-3.4 (the reference takes keccakf from tiny-keccak 2.0.2 and lists
// the same constants in keccak/constants.rs); SHA-256 K: FIPS 180-4 §4.2.2 (sha_compress_sponge/constants.rs:1-10).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

static __device__ constexpr uint64_t KECCAK_RC_DEV[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

static __device__ constexpr uint32_t SHA256_K_DEV[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// rotl64 by a compile-time-constant amount: two v_alignbit_b32 on the 32-bit halves (the generic (v << r) | (v >> (64 - r)) compiles to
// two 64-bit shifts and an OR -- ~700 rotations per Keccak-f made that a quarter of the permutation's issue time)
__device__ __forceinline__ uint64_t rotl64(uint64_t v, unsigned r) {
    r &= 63;
    if (r == 0) return v;
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (r == 32) return ((uint64_t)lo << 32) | hi;
    // alignbit(a, b, s) = low 32 bits of ({a, b} >> s), 0 < s < 32
    const uint32_t a = r < 32 ? hi : lo, b = r < 32 ? lo : hi;   // rotate by r mod 32 after swapping the halves for r > 32
    const unsigned s = 32 - (r & 31);
    const uint32_t nh = __builtin_amdgcn_alignbit(a, b, s), nl = __builtin_amdgcn_alignbit(b, a, s);
    return ((uint64_t)nh << 32) | nl;
}

// Three-input bit functions in one instruction (v_bitop3_b32, new on gfx950; the truth table is the function applied to the column
// patterns 0xF0, 0xCC, 0xAA of the three operands).  The compiler forms it for 32-bit chi but not for 64-bit values or for xor chains,
// so the 64-bit helpers split the halves by hand: a Keccak-f round is 180 instructions instead of 260.
__device__ __forceinline__ uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t chi_64(uint64_t a, uint64_t b, uint64_t c) {   // a ^ (~b & c): 0xF0 ^ (~0xCC & 0xAA) = 0xD2
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xD2);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xD2);
    return ((uint64_t)hi << 32) | lo;
}

// One Keccak-f round on a[x + 5y] (theta, rho, pi, chi, then iota with `rc`; pass rc = 0 to stop before iota).  Theta's
// D[x] = C[x - 1] ^ rotl(C[x + 1], 1) is never formed: it enters each lane as the second and third operand of one xor3.
__device__ __forceinline__ void keccak_round_dev(uint64_t (&a)[25], uint64_t rc) {
    constexpr unsigned RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    uint64_t cx[5], cr[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) cx[x] = xor3_64(xor3_64(a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; x++) cr[x] = rotl64(cx[x], 1);
#pragma unroll
    for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) {
            unsigned r = RHO[x + 5 * y];
            uint64_t v = xor3_64(a[x + 5 * y], cx[(x + 4) % 5], cr[(x + 1) % 5]);
            b[y + 5 * ((2 * x + 3 * y) % 5)] = r ? rotl64(v, r) : v;
        }
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) a[x + 5 * y] = chi_64(b[x + 5 * y], b[(x + 1) % 5 + 5 * y], b[(x + 2) % 5 + 5 * y]);
    a[0] ^= rc;
}

__device__ __forceinline__ void keccakf_dev(uint64_t (&a)[25]) {
#pragma unroll
    for (int round = 0; round < 24; round++) keccak_round_dev(a, KECCAK_RC_DEV[round]);
}
