/* oracle/commit.c -- NTT / coset LDE / Merkle-cap commitment == plonky2 PolynomialBatch.
 *
 * TEST INFRASTRUCTURE ONLY (CPU oracle / CPU baseline).
 *
 * Reference call sites: PolynomialBatch::from_values prover/src/prover.rs:154-163, 514-521;
 * from_coeffs :579-586; get_lde_values_packed :687, 723-748; merkle_tree.cap :180, 524, 588.
 * The callee is plonky2 0.1.4 fri/oracle.rs + hash/merkle_tree.rs (un-vendored): restated from
 * SURVEY.md App. A.3-A.6 -- ifft each column, zero-pad x2^rate_bits, coset_fft with shift g,
 * transpose to rows, reverse_index_bits rows, Merkle tree of hash_or_noop(leaf) / two_to_one, cap =
 * nodes at depth cap_height.  PARITY UNPINNED for the byte values of caps (no reference vectors exist).
 *
 * Storage mirrors the HIP product: LDE kept column-major with rows already in bit-reversed order
 * (lde[c*N + j] = natural evaluation bitrev(j)), so "leaf j" is row j across columns.
 */
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include "zkm_oracle.h"
#include "gl.h"

/* Threads of the LONG row-/column-parallel regions (leaf hashing, per-column LDE of a big batch): these scale to every core of the
 * host, unlike the many short regions whose fork/join cost caps the default (oracle_py.py _default_threads).  0 = the default. */
static int g_wide_threads = 0;
void zko_set_wide_threads(int n) { g_wide_threads = n > 0 ? n : 0; }
static int wide_threads(size_t work_words) {
    return (g_wide_threads > 0 && work_words >= ((size_t)1 << 24)) ? g_wide_threads : omp_get_max_threads();
}

uint64_t zko_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t zko_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t zko_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }
uint64_t zko_gl_root_of_unity(unsigned k) { return gl_root_of_unity(k); }
void zko_gl2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    gl2_t r = gl2_mul(gl2_make(a[0], a[1]), gl2_make(b[0], b[1]));
    out[0] = r.c[0]; out[1] = r.c[1];
}
void zko_gl2_inv(const uint64_t a[2], uint64_t out[2]) {
    gl2_t r = gl2_inv(gl2_make(a[0], a[1]));
    out[0] = r.c[0]; out[1] = r.c[1];
}

/* ---------------- NTT ---------------- */
static gl_t* make_twiddles(unsigned log_n, gl_t root) {
    size_t half = log_n ? (size_t)1 << (log_n - 1) : 1;
    gl_t* tw = (gl_t*)malloc(sizeof(gl_t) * half);
    tw[0] = 1;
    for (size_t i = 1; i < half; i++) tw[i] = gl_mul(tw[i - 1], root);
    return tw;
}

/* natural-order in, natural-order out; tw[i] = root^i, root a primitive 2^log_n-th root */
void zko_ntt_core(gl_t* a, unsigned log_n, const gl_t* tw) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, log_n);
        if (i < j) { gl_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, h = m >> 1, step = n >> s;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < h; j++) {
                gl_t u = a[k + j], v = gl_mul(a[k + j + h], tw[j * step]);
                a[k + j] = gl_add(u, v);
                a[k + j + h] = gl_sub(u, v);
            }
    }
}

/* plonky2 conventions (App. A.3): forward with shift: a_i *= shift^i then NTT;
 * inverse with shift: iNTT then a_i *= shift^-i.  shift == 0 or 1 means no coset. */
void zko_ntt(uint64_t* cols, size_t ncols, unsigned log_n, int inverse, uint64_t coset_shift) {
    size_t n = (size_t)1 << log_n;
    gl_t root = gl_root_of_unity(log_n);
    if (inverse) root = gl_inv(root);
    gl_t* tw = make_twiddles(log_n, root);
    gl_t ninv = gl_inv((gl_t)(n % GL_P));
    int coset = coset_shift > 1;
    gl_t sh = coset ? (inverse ? gl_inv(coset_shift) : coset_shift) : 1;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < ncols; c++) {
        gl_t* a = cols + c * n;
        if (!inverse && coset) {
            gl_t p = 1;
            for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], p); p = gl_mul(p, sh); }
        }
        zko_ntt_core(a, log_n, tw);
        if (inverse) {
            gl_t p = ninv;
            for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], p); if (coset) p = gl_mul(p, sh); }
        }
    }
    free(tw);
}

/* ---------------- Merkle ---------------- */
typedef struct {
    unsigned log_leaves, cap_height;
    uint64_t* nodes;   /* all levels concatenated; level l has (leaves >> l) digests of 4 words */
    size_t off[40];
} merkle_t;

static void merkle_alloc(merkle_t* m, unsigned log_leaves, unsigned cap_height) {
    m->log_leaves = log_leaves;
    m->cap_height = cap_height;
    size_t total = 0;
    for (unsigned l = 0; l + cap_height <= log_leaves; l++) {
        m->off[l] = total;
        total += ((size_t)1 << (log_leaves - l)) * 4;
    }
    m->nodes = (uint64_t*)malloc(sizeof(uint64_t) * total);
}

/* leaf digests must already be at level 0 */
static void merkle_build_inner(merkle_t* m) {
    unsigned top = m->log_leaves - m->cap_height;
    for (unsigned l = 1; l <= top; l++) {
        size_t cnt = (size_t)1 << (m->log_leaves - l);
        const uint64_t* ch = m->nodes + m->off[l - 1];
        uint64_t* pa = m->nodes + m->off[l];
#pragma omp parallel for schedule(static) num_threads(wide_threads(cnt * 64))
        for (size_t i = 0; i < cnt; i++) zko_poseidon_two_to_one(ch + 8 * i, ch + 8 * i + 4, pa + 4 * i);
    }
}

/* row-major leaves [nleaves][leaf_len] */
void* zko_merkle_from_rows(const uint64_t* rows, unsigned log_leaves, size_t leaf_len, unsigned cap_height) {
    merkle_t* m = (merkle_t*)malloc(sizeof(merkle_t));
    merkle_alloc(m, log_leaves, cap_height);
    size_t cnt = (size_t)1 << log_leaves;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < cnt; i++) zko_poseidon_hash_or_noop(rows + i * leaf_len, leaf_len, m->nodes + 4 * i);
    merkle_build_inner(m);
    return m;
}
void zko_merkle_free(void* mv) {
    merkle_t* m = (merkle_t*)mv;
    free(m->nodes);
    free(m);
}
void zko_merkle_cap(const void* mv, uint64_t* out) {
    const merkle_t* m = (const merkle_t*)mv;
    unsigned top = m->log_leaves - m->cap_height;
    memcpy(out, m->nodes + m->off[top], sizeof(uint64_t) * 4 * ((size_t)1 << m->cap_height));
}
void zko_merkle_path(const void* mv, size_t leaf, uint64_t* siblings) {
    const merkle_t* m = (const merkle_t*)mv;
    unsigned top = m->log_leaves - m->cap_height;
    for (unsigned l = 0; l < top; l++) memcpy(siblings + 4 * l, m->nodes + m->off[l] + 4 * ((leaf >> l) ^ 1), 32);
}
void zko_merkle_layer(const void* mv, unsigned level, uint64_t* out) {
    const merkle_t* m = (const merkle_t*)mv;
    memcpy(out, m->nodes + m->off[level], sizeof(uint64_t) * 4 * ((size_t)1 << (m->log_leaves - level)));
}

/* ---------------- PolynomialBatch ---------------- */
struct zko_batch {
    size_t ncols;
    unsigned log_n, rate_bits, cap_height;
    gl_t* coeffs; /* ncols x n, natural order */
    gl_t* lde;    /* ncols x N, row index bit-reversed */
    merkle_t* tree;
};

static zko_batch* batch_finish(zko_batch* b) {
    size_t n = (size_t)1 << b->log_n, N = n << b->rate_bits;
    unsigned log_N = b->log_n + b->rate_bits;
    b->lde = (gl_t*)malloc(sizeof(gl_t) * b->ncols * N);
    gl_t* tw = make_twiddles(log_N, gl_root_of_unity(log_N));
#pragma omp parallel num_threads(wide_threads(b->ncols * N))
    {
        gl_t* tmp = (gl_t*)malloc(sizeof(gl_t) * N);
#pragma omp for schedule(dynamic)
        for (size_t c = 0; c < b->ncols; c++) {
            gl_t p = 1;
            for (size_t i = 0; i < n; i++) { tmp[i] = gl_mul(b->coeffs[c * n + i], p); p = gl_mul(p, GL_GENERATOR); }
            memset(tmp + n, 0, sizeof(gl_t) * (N - n));
            zko_ntt_core(tmp, log_N, tw);
            gl_t* dst = b->lde + c * N;
            for (size_t j = 0; j < N; j++) dst[j] = tmp[bitrev(j, log_N)];
        }
        free(tmp);
    }
    free(tw);
    /* leaves: row j across columns */
    merkle_t* m = (merkle_t*)malloc(sizeof(merkle_t));
    merkle_alloc(m, log_N, b->cap_height);
    /* (rows are gathered 64 at a time: a row read column by column touches one cache line per column for 8 bytes of it, and on a
     * many-core host the gather, not the hashing, was what the row-parallel loop waited for -- profiles/cpu_oracle_full_size.json) */
    enum { LEAF_TILE = 64 };
#pragma omp parallel num_threads(wide_threads(b->ncols * N))
    {
        gl_t* tile = (gl_t*)malloc(sizeof(gl_t) * b->ncols * LEAF_TILE);
#pragma omp for schedule(static)
        for (size_t j0 = 0; j0 < N; j0 += LEAF_TILE) {
            const size_t t = N - j0 < LEAF_TILE ? N - j0 : LEAF_TILE;
            for (size_t c = 0; c < b->ncols; c++)
                for (size_t r = 0; r < t; r++) tile[r * b->ncols + c] = b->lde[c * N + j0 + r];
            for (size_t r = 0; r < t; r++) zko_poseidon_hash_or_noop(tile + r * b->ncols, b->ncols, m->nodes + 4 * (j0 + r));
        }
        free(tile);
    }
    merkle_build_inner(m);
    b->tree = m;
    return b;
}

zko_batch* zko_batch_from_coeffs(const uint64_t* coeffs, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height) {
    zko_batch* b = (zko_batch*)calloc(1, sizeof(zko_batch));
    size_t n = (size_t)1 << log_n;
    b->ncols = ncols; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->coeffs = (gl_t*)malloc(sizeof(gl_t) * ncols * n);
    memcpy(b->coeffs, coeffs, sizeof(gl_t) * ncols * n);
    return batch_finish(b);
}

zko_batch* zko_batch_from_values(const uint64_t* values, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height) {
    zko_batch* b = (zko_batch*)calloc(1, sizeof(zko_batch));
    size_t n = (size_t)1 << log_n;
    b->ncols = ncols; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->coeffs = (gl_t*)malloc(sizeof(gl_t) * ncols * n);
    memcpy(b->coeffs, values, sizeof(gl_t) * ncols * n);
    zko_ntt(b->coeffs, ncols, log_n, 1, 0);
    return batch_finish(b);
}

void zko_batch_free(zko_batch* b) {
    if (!b) return;
    free(b->coeffs);
    free(b->lde);
    if (b->tree) zko_merkle_free(b->tree);
    free(b);
}
void zko_batch_cap(const zko_batch* b, uint64_t* out) { zko_merkle_cap(b->tree, out); }
void zko_batch_coeffs(const zko_batch* b, uint64_t* out) {
    memcpy(out, b->coeffs, sizeof(gl_t) * b->ncols * ((size_t)1 << b->log_n));
}
void zko_batch_leaf(const zko_batch* b, size_t leaf, uint64_t* out) {
    size_t N = (size_t)1 << (b->log_n + b->rate_bits);
    for (size_t c = 0; c < b->ncols; c++) out[c] = b->lde[c * N + leaf];
}
void zko_batch_lde_row(const zko_batch* b, size_t natural_index, uint64_t* out) {
    zko_batch_leaf(b, bitrev(natural_index, b->log_n + b->rate_bits), out);
}
void zko_batch_merkle_path(const zko_batch* b, size_t leaf, uint64_t* siblings) { zko_merkle_path(b->tree, leaf, siblings); }
void zko_batch_digest_layer(const zko_batch* b, unsigned level, uint64_t* out) { zko_merkle_layer(b->tree, level, out); }

/* internal accessors for prover.c */
const gl_t* zko_batch_coeffs_ptr(const zko_batch* b) { return b->coeffs; }
const gl_t* zko_batch_lde_ptr(const zko_batch* b) { return b->lde; }
size_t zko_batch_ncols(const zko_batch* b) { return b->ncols; }
unsigned zko_batch_log_n(const zko_batch* b) { return b->log_n; }
