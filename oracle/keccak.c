/* oracle/keccak.c -- Keccak-f[1600] and Keccak-256 (pad 0x01 .. 0x80, rate 136).
 *
 * TEST INFRASTRUCTURE ONLY (CPU oracle / CPU baseline).
 *
 * The reference calls tiny-keccak 2.0.2 `keccakf` (crates.io dependency, not in the tree) through
 * /root/reference/prover/src/cpu/kernel/keccak_util.rs:6-31; this file restates the published
 * Keccak-f[1600] permutation (FIPS 202 §3.2-3.3: theta, rho, pi, chi, iota; 24 rounds).  Pinned by the
 * in-tree known-answer test keccak_util.rs:39-59 (tests/golden/keccakf_kat.json) and by the standard
 * Keccak-256 digests.  Sponge padding as keccak_sponge_stark.rs:334-341.
 */
#include <string.h>
#include "zkm_oracle.h"

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

/* rho rotation offsets indexed [x + 5*y] */
static const unsigned RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

static inline uint64_t rotl(uint64_t v, unsigned r) { return r ? (v << r) | (v >> (64 - r)) : v; }

void zko_keccakf(uint64_t a[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        /* rho + pi: B[y, 2x+3y] = rot(A[x,y]) */
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}

void zko_keccakf_batch(uint64_t* states, size_t k) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < k; i++) zko_keccakf(states + 25 * i);
}

void zko_keccak256(const uint8_t* msg, size_t len, uint8_t out[32]) {
    uint64_t st[25];
    uint8_t block[136];
    memset(st, 0, sizeof st);
    size_t off = 0;
    for (;;) {
        size_t k = len - off;
        int last = k < 136;
        if (last) {
            memset(block, 0, 136);
            memcpy(block, msg + off, k);
            block[k] ^= 0x01;
            block[135] ^= 0x80;
        } else {
            memcpy(block, msg + off, 136);
        }
        for (int i = 0; i < 17; i++) {
            uint64_t w = 0;
            for (int j = 0; j < 8; j++) w |= (uint64_t)block[8 * i + j] << (8 * j);
            st[i] ^= w;
        }
        zko_keccakf(st);
        if (last) break;
        off += 136;
    }
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(st[i / 8] >> (8 * (i % 8)));
}
