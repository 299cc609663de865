/* oracle/poseidon.c -- Goldilocks Poseidon permutation and the plonky2 hashing modes built on it.
 *
 * TEST INFRASTRUCTURE ONLY (CPU oracle / CPU baseline).
 *
 * Permutation: restates the in-tree copy at /root/reference/prover/src/poseidon/poseidon_stark.rs
 *   poseidon_with_witness :51-64, full_rounds :65-77, partial_rounds :79-95, constant_layer :164-169,
 *   sbox_monomial :239-251, mds_layer :310-331 / mds_row_shf :333-345, mds_partial_layer_init :392-404,
 *   mds_partial_layer_fast :463-487, parameter tables constants.rs:11-870.
 * Pinned by the two upstream plonky2 test vectors in tests/test_oracle_poseidon.py,
 * and by naive-form == fast-form equality.
 *
 * Hash modes (hash_no_pad, hash_or_noop, two_to_one): plonky2 0.1.4 `hashing.rs` / `poseidon.rs`
 * (un-vendored) -- PARITY UNPINNED beyond the permutation itself: overwrite-mode
 * sponge, rate 8, output state[0..4]; <=4 elements are copied, not hashed.
 */
#include "zkm_oracle.h"
#include "gl.h"
#include "poseidon_constants.inc"

#define W 12
#define HALF_FULL 4
#define N_PARTIAL 22

static inline gl_t sbox7(gl_t x) {
    gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

static void mds_layer(gl_t s[W]) {
    gl_t out[W];
    for (int r = 0; r < W; r++) {
        u128_t acc = 0;
        for (int i = 0; i < W; i++) acc += (u128_t)s[(i + r) % W] * ZKM_POSEIDON_MDS_CIRC[i];
        acc += (u128_t)s[r] * ZKM_POSEIDON_MDS_DIAG[r];
        out[r] = gl_reduce128(acc);
    }
    for (int r = 0; r < W; r++) s[r] = out[r];
}

static void full_round(gl_t s[W], int round_ctr) {
    for (int i = 0; i < W; i++) s[i] = gl_add(s[i], gl_canon(ZKM_POSEIDON_RC[i + W * round_ctr]));
    for (int i = 0; i < W; i++) s[i] = sbox7(s[i]);
    mds_layer(s);
}

/* Textbook form: every partial round = constants, sbox on lane 0, dense MDS. */
void zko_poseidon_permute_naive(uint64_t s[W]) {
    int rc = 0;
    for (int r = 0; r < HALF_FULL; r++) full_round(s, rc++);
    for (int r = 0; r < N_PARTIAL; r++, rc++) {
        for (int i = 0; i < W; i++) s[i] = gl_add(s[i], gl_canon(ZKM_POSEIDON_RC[i + W * rc]));
        s[0] = sbox7(s[0]);
        mds_layer(s);
    }
    for (int r = 0; r < HALF_FULL; r++) full_round(s, rc++);
}

static void partial_rounds_fast(gl_t s[W]) {
    for (int i = 0; i < W; i++) s[i] = gl_add(s[i], ZKM_POSEIDON_FAST_FIRST_RC[i]);
    /* dense pre-matrix on lanes 1..11: result[c] = sum_r s[r] * INIT[r-1][c-1] */
    gl_t t[W];
    t[0] = s[0];
    for (int c = 1; c < W; c++) {
        gl_t acc = 0;
        for (int r = 1; r < W; r++) acc = gl_add(acc, gl_mul(s[r], ZKM_POSEIDON_FAST_INIT[r - 1][c - 1]));
        t[c] = acc;
    }
    for (int i = 0; i < W; i++) s[i] = t[i];
    for (int r = 0; r < N_PARTIAL; r++) {
        s[0] = sbox7(s[0]);
        s[0] = gl_add(s[0], ZKM_POSEIDON_FAST_RC[r]);
        gl_t d = gl_mul(s[0], ZKM_POSEIDON_MDS_CIRC[0] + ZKM_POSEIDON_MDS_DIAG[0]);
        for (int i = 1; i < W; i++) d = gl_add(d, gl_mul(s[i], ZKM_POSEIDON_FAST_W_HATS[r][i - 1]));
        for (int i = 1; i < W; i++) s[i] = gl_add(s[i], gl_mul(s[0], ZKM_POSEIDON_FAST_VS[r][i - 1]));
        s[0] = d;
    }
}

void zko_poseidon_permute(uint64_t s[W]) {
    int rc = 0;
    for (int r = 0; r < HALF_FULL; r++) full_round(s, rc++);
    partial_rounds_fast(s);
    rc += N_PARTIAL;
    for (int r = 0; r < HALF_FULL; r++) full_round(s, rc++);
}

/* The STARK witness row (262 columns) for one permutation; column map = poseidon/columns.rs:3-54;
 * poseidon_with_witness poseidon_stark.rs:51-64 + generate_trace_rows_for_perm :129-147. */
void zko_poseidon_witness_row(const uint64_t in[W], uint64_t timestamp, int filter, uint64_t row[ZKO_POSEIDON_COLS]) {
    gl_t s[W];
    for (int i = 0; i < ZKO_POSEIDON_COLS; i++) row[i] = 0;
    for (int i = 0; i < W; i++) s[i] = in[i];
    int rc = 0;
    for (int half = 0; half < 2; half++) {
        int base = half == 0 ? 26 : 166;
        for (int r = 0; r < HALF_FULL; r++, rc++) {
            for (int i = 0; i < W; i++) s[i] = gl_add(s[i], gl_canon(ZKM_POSEIDON_RC[i + W * rc]));
            for (int i = 0; i < W; i++) {
                gl_t x3 = gl_mul(gl_sqr(s[i]), s[i]);
                gl_t x7 = gl_mul(s[i], gl_sqr(x3));
                row[base + 24 * r + 2 * i] = x3;
                row[base + 24 * r + 2 * i + 1] = x7;
                s[i] = x7;
            }
            mds_layer(s);
        }
        if (half == 0) {
            for (int i = 0; i < W; i++) s[i] = gl_add(s[i], ZKM_POSEIDON_FAST_FIRST_RC[i]);
            gl_t t[W];
            t[0] = s[0];
            for (int c = 1; c < W; c++) {
                gl_t acc = 0;
                for (int r = 1; r < W; r++) acc = gl_add(acc, gl_mul(s[r], ZKM_POSEIDON_FAST_INIT[r - 1][c - 1]));
                t[c] = acc;
            }
            for (int i = 0; i < W; i++) s[i] = t[i];
            for (int r = 0; r < N_PARTIAL; r++) {
                gl_t x3 = gl_mul(gl_sqr(s[0]), s[0]);
                gl_t x7 = gl_mul(s[0], gl_sqr(x3));
                row[122 + 2 * r] = x3;
                row[122 + 2 * r + 1] = x7;
                s[0] = gl_add(x7, ZKM_POSEIDON_FAST_RC[r]);
                gl_t d = gl_mul(s[0], ZKM_POSEIDON_MDS_CIRC[0] + ZKM_POSEIDON_MDS_DIAG[0]);
                for (int i = 1; i < W; i++) d = gl_add(d, gl_mul(s[i], ZKM_POSEIDON_FAST_W_HATS[r][i - 1]));
                for (int i = 1; i < W; i++) s[i] = gl_add(s[i], gl_mul(s[0], ZKM_POSEIDON_FAST_VS[r][i - 1]));
                s[0] = d;
            }
            rc += N_PARTIAL;
        }
    }
    row[0] = filter ? 1 : 0;
    for (int i = 0; i < W; i++) { row[1 + i] = in[i]; row[13 + i] = s[i]; }
    row[25] = timestamp;
}

void zko_poseidon_hash_no_pad(const uint64_t* in, size_t len, uint64_t out[4]) {
    gl_t st[W] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; i++) st[i] = in[off + i];
        zko_poseidon_permute(st);
    }
    for (int i = 0; i < 4; i++) out[i] = st[i];
}

void zko_poseidon_hash_or_noop(const uint64_t* in, size_t len, uint64_t out[4]) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? in[i] : 0;
    } else {
        zko_poseidon_hash_no_pad(in, len, out);
    }
}

void zko_poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    gl_t st[W] = {0};
    for (int i = 0; i < 4; i++) { st[i] = l[i]; st[4 + i] = r[i]; }
    zko_poseidon_permute(st);
    for (int i = 0; i < 4; i++) out[i] = st[i];
}

void zko_poseidon_permute_batch(uint64_t* states, size_t k) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < k; i++) zko_poseidon_permute(states + 12 * i);
}
