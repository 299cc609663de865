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
#include "hash_constants.h"


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
        a[0] ^= ZKO_KECCAK_RC[round];
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

/* ---- KeccakSpongeStark witness rows: restates keccak_sponge_stark.rs:253-438 (generate_rows_for_op,
 * generate_full_input_row, generate_final_row, generate_common_fields) with the column order of columns.rs:19-70. ---- */
enum { KS_FULL = 0, KS_CONTEXT = 1, KS_SEGMENT = 2, KS_VIRT = 3, KS_TIMESTAMP = 37, KS_LEN = 38, KS_ABSORBED = 39, KS_FINAL_LEN = 40,
       KS_ORIG_RATE = 176, KS_ORIG_CAP = 210, KS_BLOCK = 226, KS_XORED = 362, KS_PARTIAL = 396, KS_DIGEST = 438 };

size_t zko_keccak_sponge_trace(const uint8_t* inputs, const uint64_t* off, const uint64_t* meta, size_t nops, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n, rows = 0;
    for (size_t i = 0; i < nops; i++) {
        if (off[i + 1] <= off[i]) return 0;
        rows += (off[i + 1] - off[i]) / 136 + 1;
    }
    if (rows > n) return 0;
    memset(out, 0, sizeof(uint64_t) * ZKO_KECCAK_SPONGE_COLS * n);
    size_t row = 0;
    for (size_t op = 0; op < nops; op++) {
        const uint8_t* msg = inputs + off[op];
        size_t len = off[op + 1] - off[op], nwords = (len + 3) / 4;
        uint64_t st[25];
        memset(st, 0, sizeof st);
        size_t absorbed = 0;
        for (;;) {
            size_t rem = len - absorbed;
            int full = rem >= 136;
            uint8_t block[136];
            memset(block, 0, 136);
            memcpy(block, msg + absorbed, full ? 136 : rem);
#define CELL(c) out[(size_t)(c) * n + row]
            if (full) {
                CELL(KS_FULL) = 1;
            } else {
                if (rem == 135) block[135] = 0x81;
                else { block[rem] = 1; block[135] = 0x80; }
                CELL(KS_FINAL_LEN + rem) = 1;
            }
            CELL(KS_CONTEXT) = meta[4 * op];
            CELL(KS_SEGMENT) = meta[4 * op + 1];
            for (size_t i = 0; i < 34; i++) { size_t w = absorbed / 4 + i; CELL(KS_VIRT + i) = w < nwords ? meta[4 * op + 2] + w : 0; }
            CELL(KS_TIMESTAMP) = meta[4 * op + 3];
            CELL(KS_LEN) = len;
            CELL(KS_ABSORBED) = absorbed;
            for (int i = 0; i < 34; i++) CELL(KS_ORIG_RATE + i) = (uint32_t)(st[i / 2] >> (32 * (i & 1)));
            for (int i = 0; i < 16; i++) CELL(KS_ORIG_CAP + i) = (uint32_t)(st[(34 + i) / 2] >> (32 * ((34 + i) & 1)));
            for (int i = 0; i < 136; i++) CELL(KS_BLOCK + i) = block[i];
            for (int i = 0; i < 17; i++) {
                uint64_t w = 0;
                for (int j = 0; j < 8; j++) w |= (uint64_t)block[8 * i + j] << (8 * j);
                st[i] ^= w;
            }
            for (int i = 0; i < 34; i++) CELL(KS_XORED + i) = (uint32_t)(st[i / 2] >> (32 * (i & 1)));
            zko_keccakf(st);
            for (int i = 0; i < 42; i++) CELL(KS_PARTIAL + i) = (uint32_t)(st[(8 + i) / 2] >> (32 * ((8 + i) & 1)));
            for (int i = 0; i < 32; i++) CELL(KS_DIGEST + i) = (uint8_t)(st[i / 8] >> (8 * (i & 7)));
#undef CELL
            row++;
            if (!full) break;
            absorbed += 136;
        }
    }
    return rows;
}

/* ---- KeccakStark witness rows: restates keccak/keccak_stark.rs:62-226 (generate_trace_rows_for_perm,
 * copy_output_to_input, generate_trace_row_for_round) with the register map of keccak/columns.rs.  24 rows per
 * permutation; A(x, y) = input[y*5 + x]. ---- */

size_t zko_keccak_trace(const uint64_t* inputs, const uint64_t* timestamps, size_t nperms, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    if (nperms * 24 > n) return 0;
    memset(out, 0, sizeof(uint64_t) * ZKO_KECCAK_COLS * n);
    for (size_t p = 0; p < nperms; p++) {
        uint64_t A[5][5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) A[x][y] = inputs[25 * p + y * 5 + x];
        for (int round = 0; round < 24; round++) {
            size_t row = 24 * p + round;
#define CELL(c) out[(size_t)(c) * n + row]
            CELL(round) = 1;
            CELL(24) = timestamps[p];
            uint64_t C[5], Cp[5], Ap[5][5], App[5][5];
            for (int x = 0; x < 5; x++) {
                C[x] = A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4];
                for (int y = 0; y < 5; y++) {
                    CELL(25 + (x * 5 + y) * 2) = (uint32_t)A[x][y];
                    CELL(25 + (x * 5 + y) * 2 + 1) = A[x][y] >> 32;
                }
            }
            for (int x = 0; x < 5; x++) Cp[x] = C[x] ^ C[(x + 4) % 5] ^ rotl(C[(x + 1) % 5], 1);
            for (int x = 0; x < 5; x++)
                for (int z = 0; z < 64; z++) {
                    CELL(75 + x * 64 + z) = (C[x] >> z) & 1;
                    CELL(395 + x * 64 + z) = (Cp[x] >> z) & 1;
                }
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) {
                    Ap[x][y] = A[x][y] ^ C[x] ^ Cp[x];
                    for (int z = 0; z < 64; z++) CELL(715 + x * 320 + y * 64 + z) = (Ap[x][y] >> z) & 1;
                }
            /* B[x, y] = ROT(A'[(x + 3y) % 5, x], r); A''[x, y] = B[x, y] ^ (~B[x+1, y] & B[x+2, y]) */
            uint64_t B[5][5];
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) { int a = (x + 3 * y) % 5; B[x][y] = rotl(Ap[a][x], ZKO_KECCAK_R[a][x]); }
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) {
                    App[x][y] = B[x][y] ^ (~B[(x + 1) % 5][y] & B[(x + 2) % 5][y]);
                    CELL(2315 + x * 10 + y * 2) = (uint32_t)App[x][y];
                    CELL(2315 + x * 10 + y * 2 + 1) = App[x][y] >> 32;
                }
            for (int z = 0; z < 64; z++) CELL(2365 + z) = (App[0][0] >> z) & 1;
            uint64_t appp = App[0][0] ^ ZKO_KECCAK_RC[round];
            CELL(2429) = (uint32_t)appp;
            CELL(2430) = appp >> 32;
#undef CELL
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) A[x][y] = App[x][y];
            A[0][0] = appp;
        }
    }
    return nperms * 24;
}

/* ---- SHA-256 message-schedule tables ---- */
static inline uint32_t rotr32(uint32_t x, unsigned r) { return (x >> r) | (x << (32 - r)); }
static void put_le4(uint64_t* out, size_t n, size_t row, int col, uint32_t v) {
    for (int j = 0; j < 4; j++) out[(size_t)(col + j) * n + row] = (v >> (8 * j)) & 0xFF;
}
static void put_rot(uint64_t* out, size_t n, size_t row, int col, uint32_t in, unsigned r, int is_shift) {
    /* RotateRightOp / ShiftRightOp::generate_trace (rotate_right.rs:15-27, shift_right.rs:15-27), shr_carry :108-117 */
    uint32_t shift = in >> r, carry = (uint32_t)(((uint64_t)in << (32 - r)) & 0xFFFFFFFFu) >> (32 - r);
    put_le4(out, n, row, col, is_shift ? shift : rotr32(in, r));
    out[(size_t)(col + 4) * n + row] = shift;
    out[(size_t)(col + 5) * n + row] = carry;
}
/* ShaExtendStark::generate_trace (sha_extend/sha_extend_stark.rs:121-236): inputs = k x 16 bytes (w[i-15], w[i-2], w[i-16], w[i-7],
 * little-endian bytes of each word), one timestamp per row; padding rows zero. */
void zko_sha_extend_trace(const uint8_t* inputs, const uint64_t* timestamps, size_t k, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    memset(out, 0, sizeof(uint64_t) * 78 * n);
    for (size_t r = 0; r < k && r < n; r++) {
        uint32_t w[4];
        for (int q = 0; q < 4; q++) { memcpy(&w[q], inputs + 16 * r + 4 * q, 4); put_le4(out, n, r, 8 + 4 * q, w[q]); }
        uint32_t w15 = w[0], w2 = w[1], w16 = w[2], w7 = w[3];
        put_rot(out, n, r, 40, w15, 7, 0);
        put_rot(out, n, r, 46, w15, 18, 0);
        put_rot(out, n, r, 70, w15, 3, 1);
        uint32_t s0i = rotr32(w15, 7) ^ rotr32(w15, 18), s0 = s0i ^ (w15 >> 3);
        put_le4(out, n, r, 24, s0i);
        put_le4(out, n, r, 28, s0);
        put_rot(out, n, r, 52, w2, 17, 0);
        put_rot(out, n, r, 58, w2, 19, 0);
        put_rot(out, n, r, 64, w2, 10, 1);
        uint32_t s1i = rotr32(w2, 17) ^ rotr32(w2, 19), s1 = s1i ^ (w2 >> 10);
        put_le4(out, n, r, 32, s1i);
        put_le4(out, n, r, 36, s1);
        uint64_t wide = (uint64_t)s1 + w7 + s0 + w16;
        put_le4(out, n, r, 0, (uint32_t)wide);
        out[(size_t)(4 + (wide >> 32)) * n + r] = 1;
        out[(size_t)76 * n + r] = timestamps[r];
        out[(size_t)77 * n + r] = 1;
    }
}
/* ShaExtendSpongeStark::generate_trace (sha_extend_sponge_stark.rs:131-215) for k complete message schedules: w16 = k x 16
 * words w[0..15]; meta = k x 4 {context, segment, address of w[0], timestamp of round 0}; word j lives at address + 4 j; round i
 * (row 48 e + i) reads w[i+1], w[i+14], w[i], w[i+9], writes w[i+16] and is stamped timestamp + 20 i (2 * NUM_CHANNELS). */
size_t zko_sha_extend_sponge_trace(const uint32_t* w16, const uint64_t* meta, size_t k, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    if (48 * k > n) return 0;
    memset(out, 0, sizeof(uint64_t) * 76 * n);
    for (size_t e = 0; e < k; e++) {
        uint32_t w[64];
        memcpy(w, w16 + 16 * e, 64);
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = s1 + w[i - 16] + s0 + w[i - 7];
        }
        for (int rd = 0; rd < 48; rd++) {
            size_t row = 48 * e + rd;
            int i = rd + 16;
            out[(size_t)rd * n + row] = 1;
            const int src[4] = {i - 15, i - 2, i - 16, i - 7};
            for (int q = 0; q < 4; q++) {
                put_le4(out, n, row, 48 + 4 * q, w[src[q]]);
                out[(size_t)(68 + q) * n + row] = meta[4 * e + 2] + 4 * (uint64_t)src[q];
            }
            put_le4(out, n, row, 64, w[i]);
            out[(size_t)72 * n + row] = meta[4 * e + 2] + 4 * (uint64_t)i;
            out[(size_t)73 * n + row] = meta[4 * e];
            out[(size_t)74 * n + row] = meta[4 * e + 1];
            out[(size_t)75 * n + row] = meta[4 * e + 3] + 20 * (uint64_t)rd;
        }
    }
    return 48 * k;
}

/* ---- SHA-256 compression tables ---- */
static void put_wadd(uint64_t* out, size_t n, size_t row, int col, uint64_t wide) {
    put_le4(out, n, row, col, (uint32_t)wide);
    out[(size_t)(col + 4 + (wide >> 32)) * n + row] = 1;
}
/* One SHA-256 round on (a..h) with message word w and constant kc; returns temp1, temp2 through pointers */
static void sha_round(uint32_t s[8], uint32_t w, uint32_t kc) {
    uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    uint32_t s1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + s1 + ch + kc + w;
    uint32_t s0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), maj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = s0 + maj;
    s[7] = g; s[6] = f; s[5] = e; s[4] = d + t1; s[3] = c; s[2] = b; s[1] = a; s[0] = t1 + t2;
}
/* ShaCompressStark::generate_trace (sha_compress/sha_compress_stark.rs:227-400) for k compressions as the witness generator emits
 * them (witness/util.rs:605-690): 65 rows each -- rounds 0..63 on the running state with w_i, K_i, and a 65th row holding the final
 * state with w_i = k_i = 0.  hx = k x 8 words, w = k x 64 words, meta = k x 8 {context, segment, hx address, timestamp,
 * w address, w segment, w context, unused}. */
size_t zko_sha_compress_trace(const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t k, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    if (65 * k > n) return 0;
    memset(out, 0, sizeof(uint64_t) * 224 * n);
    for (size_t e_ = 0; e_ < k; e_++) {
        uint32_t s[8];
        memcpy(s, hx + 8 * e_, 32);
        for (int rd = 0; rd < 65; rd++) {
            size_t row = 65 * e_ + rd;
            uint32_t wi = rd < 64 ? w[64 * e_ + rd] : 0, ki = rd < 64 ? ZKO_SHA256_K[rd] : 0;
            uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
            for (int q = 0; q < 8; q++) put_le4(out, n, row, 4 * q, s[q]);
            put_le4(out, n, row, 32, ~e);
            put_le4(out, n, row, 36, wi);
            put_le4(out, n, row, 40, ki);
            uint32_t s1i = rotr32(e, 6) ^ rotr32(e, 11), s1 = s1i ^ rotr32(e, 25);
            uint32_t eaf = e & f, eng = ~e & g, ch = eaf ^ eng;
            put_le4(out, n, row, 44, s1i); put_le4(out, n, row, 48, s1); put_le4(out, n, row, 52, eaf);
            put_le4(out, n, row, 56, eng); put_le4(out, n, row, 60, ch);
            uint32_t s0i = rotr32(a, 2) ^ rotr32(a, 13), s0 = s0i ^ rotr32(a, 22);
            uint32_t ab = a & b, ac = a & c, bc = b & c, maji = ab ^ ac, maj = maji ^ bc;
            put_le4(out, n, row, 64, s0i); put_le4(out, n, row, 68, s0); put_le4(out, n, row, 72, ab); put_le4(out, n, row, 76, ac);
            put_le4(out, n, row, 80, bc); put_le4(out, n, row, 84, maji); put_le4(out, n, row, 88, maj);
            put_rot(out, n, row, 92, e, 6, 0); put_rot(out, n, row, 98, e, 11, 0); put_rot(out, n, row, 104, e, 25, 0);
            put_rot(out, n, row, 110, a, 2, 0); put_rot(out, n, row, 116, a, 13, 0); put_rot(out, n, row, 122, a, 22, 0);
            uint64_t t1w = (uint64_t)h + s1 + ch + ki + wi;
            uint32_t t1 = (uint32_t)t1w;
            uint64_t t2w = (uint64_t)s0 + maj;
            uint32_t t2 = (uint32_t)t2w;
            put_wadd(out, n, row, 150, t1w);
            put_wadd(out, n, row, 128, t2w);
            put_wadd(out, n, row, 134, (uint64_t)d + t1);
            put_wadd(out, n, row, 140, (uint64_t)t1 + t2);
            out[(size_t)146 * n + row] = meta[8 * e_ + 3];
            out[(size_t)147 * n + row] = meta[8 * e_ + 5];
            out[(size_t)148 * n + row] = meta[8 * e_ + 6];
            out[(size_t)149 * n + row] = meta[8 * e_ + 4] + 4 * (uint64_t)rd;
            out[(size_t)(159 + rd) * n + row] = 1;
            if (rd < 64) sha_round(s, wi, ki);
        }
    }
    return 65 * k;
}
/* ShaCompressSpongeStark::generate_trace (sha_compress_sponge/sha_compress_sponge_stark.rs:118-230): one row per compression */
void zko_sha_compress_sponge_trace(const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t k, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    memset(out, 0, sizeof(uint64_t) * 127 * n);
    for (size_t r = 0; r < k && r < n; r++) {
        uint32_t s[8];
        memcpy(s, hx + 8 * r, 32);
        for (int i = 0; i < 64; i++) sha_round(s, w[64 * r + i], ZKO_SHA256_K[i]);
        for (int q = 0; q < 8; q++) {
            put_le4(out, n, r, 4 * q, hx[8 * r + q]);
            put_le4(out, n, r, 32 + 4 * q, s[q]);
            put_wadd(out, n, r, 64 + 6 * q, (uint64_t)hx[8 * r + q] + s[q]);
            out[(size_t)(112 + q) * n + r] = meta[8 * r + 2] + 4 * (uint64_t)q;
        }
        out[(size_t)120 * n + r] = meta[8 * r + 4];
        out[(size_t)121 * n + r] = meta[8 * r + 3];
        out[(size_t)122 * n + r] = meta[8 * r];
        out[(size_t)123 * n + r] = meta[8 * r + 1];
        out[(size_t)124 * n + r] = meta[8 * r + 5];
        out[(size_t)125 * n + r] = meta[8 * r + 6];
        out[(size_t)126 * n + r] = 1;
    }
}
