/* oracle/prover.c -- CPU restatement of prove_single_table and of its native verifier.
 *
 * TEST INFRASTRUCTURE ONLY (CPU oracle / CPU baseline).
 *
 * Follows, in this order of authority (all under /root/reference/prover/src):
 *   prover.rs:441-641   prove_single_table (stage order, transcript order)
 *   prover.rs:645-789   compute_quotient_polys (step / next_step, Z_H, Lagrange first/last, chunking :560-575)
 *   stark.rs:91-148     fri_instance (three batches: zeta, g*zeta, 1)
 *   proof.rs:299-367    StarkOpeningSet::new / to_fri_openings
 *   get_challenges.rs:190-233, verifier.rs:178-292, :344-354   verifier side
 *   config.rs:17-33     standard_fast_config
 * and SURVEY.md App. A.7-A.10 for the plonky2 0.1.4 internals it calls (Challenger, prove_openings,
 * fri_proof, proof of work, FRI verifier).  Those internals are RECALLED: "parity unpinned" at the byte
 * level; validity is pinned by prove -> verify below.  PoW takes the SMALLEST witness (App. A.9).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "zkm_oracle.h"
#include "hash_constants.h"
#include "arith_constants.inc"
#include "gl.h"
#include "poseidon_constants.inc"

/* from commit.c */
void zko_ntt_core(gl_t* a, unsigned log_n, const gl_t* tw);
void* zko_merkle_from_rows(const uint64_t* rows, unsigned log_leaves, size_t leaf_len, unsigned cap_height);
void zko_merkle_free(void* m);
void zko_merkle_cap(const void* m, uint64_t* out);
void zko_merkle_path(const void* m, size_t leaf, uint64_t* siblings);
const gl_t* zko_batch_coeffs_ptr(const zko_batch* b);
const gl_t* zko_batch_lde_ptr(const zko_batch* b);
size_t zko_batch_ncols(const zko_batch* b);
unsigned zko_batch_log_n(const zko_batch* b);

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------ constraint templates */
#define T gl_t
#define TNAME(x) b_##x
#define T_ADD(a, b) gl_add(a, b)
#define T_SUB(a, b) gl_sub(a, b)
#define T_MUL(a, b) gl_mul(a, b)
#define T_MULB(a, s) gl_mul(a, s)
#define T_FROMB(s) ((gl_t)(s))
#include "constraints_tmpl.h"
#undef T
#undef TNAME
#undef T_ADD
#undef T_SUB
#undef T_MUL
#undef T_MULB
#undef T_FROMB

#define T gl2_t
#define TNAME(x) e_##x
#define T_ADD(a, b) gl2_add(a, b)
#define T_SUB(a, b) gl2_sub(a, b)
#define T_MUL(a, b) gl2_mul(a, b)
#define T_MULB(a, s) gl2_scalar_mul(a, s)
#define T_FROMB(s) gl2_from_base(s)
#include "constraints_tmpl.h"
#undef T
#undef TNAME
#undef T_ADD
#undef T_SUB
#undef T_MUL
#undef T_MULB
#undef T_FROMB

/* ------------------------------------------------------------------ challenger (App. A.7) */
void zko_challenger_init(zko_challenger* c) { memset(c, 0, sizeof *c); }

static void duplex(zko_challenger* c) {
    for (uint32_t i = 0; i < c->n_in; i++) c->state[i] = c->in_buf[i];
    c->n_in = 0;
    zko_poseidon_permute(c->state);
    for (int i = 0; i < 8; i++) c->out_buf[i] = c->state[i];
    c->n_out = 8;
}
void zko_challenger_observe(zko_challenger* c, const uint64_t* e, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->n_out = 0;
        c->in_buf[c->n_in++] = e[i];
        if (c->n_in == 8) duplex(c);
    }
}
uint64_t zko_challenger_get(zko_challenger* c) {
    if (c->n_in != 0 || c->n_out == 0) duplex(c);
    return c->out_buf[--c->n_out];
}
void zko_challenger_compact(zko_challenger* c, uint64_t out[12]) {
    if (c->n_in != 0) duplex(c);
    c->n_out = 0;
    memcpy(out, c->state, sizeof c->state);
}
static gl2_t challenger_get_ext(zko_challenger* c) {
    gl_t a = zko_challenger_get(c), b = zko_challenger_get(c);
    return gl2_make(a, b);
}

/* ------------------------------------------------------------------ synthetic trace */
static inline uint64_t splitmix_at(uint64_t seed, uint64_t k) {
    uint64_t z = seed + k * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void zko_poseidon_trace(uint64_t seed, size_t num_perms, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    uint64_t zero_in[12] = {0}, def_row[ZKO_POSEIDON_COLS];
    zko_poseidon_witness_row(zero_in, 0, 0, def_row);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n; r++) {
        uint64_t row[ZKO_POSEIDON_COLS];
        const uint64_t* src = def_row;
        if (r < num_perms) {
            uint64_t in[12];
            for (int i = 0; i < 12; i++) in[i] = gl_canon(splitmix_at(seed, r * 12 + i + 1));
            zko_poseidon_witness_row(in, 0, 1, row);
            src = row;
        }
        for (size_t c = 0; c < ZKO_POSEIDON_COLS; c++) out[c * n + r] = src[c];
    }
}

/* PoseidonStark::generate_trace (poseidon_stark.rs:104-160) for explicit (input, timestamp) pairs */
void zko_poseidon_trace_inputs(const uint64_t* inputs, const uint64_t* timestamps, size_t num_perms, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    uint64_t zero_in[12] = {0}, def_row[ZKO_POSEIDON_COLS];
    zko_poseidon_witness_row(zero_in, 0, 0, def_row);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n; r++) {
        uint64_t row[ZKO_POSEIDON_COLS];
        const uint64_t* src = def_row;
        if (r < num_perms) {
            zko_poseidon_witness_row(inputs + 12 * r, timestamps[r], 1, row);
            src = row;
        }
        for (size_t c = 0; c < ZKO_POSEIDON_COLS; c++) out[c * n + r] = src[c];
    }
}

/* PoseidonSpongeStark::generate_trace (poseidon_sponge_stark.rs:186-381): one row per 32-byte block (len/32 + 1 rows per
 * operation, pad10*1 on the final row), the block's 8 little-endian u32 words overwrite the rate, one Poseidon permutation per
 * row; padding rows are zero.  Operation layout as zko_keccak_sponge_trace. */
size_t zko_poseidon_sponge_trace(const uint8_t* inputs, const uint64_t* off, const uint64_t* meta, size_t nops, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n, rows = 0;
    for (size_t i = 0; i < nops; i++) {
        if (off[i + 1] <= off[i]) return 0;
        rows += (off[i + 1] - off[i]) / 32 + 1;
    }
    if (rows > n) return 0;
    memset(out, 0, sizeof(uint64_t) * 110 * n);
    size_t row = 0;
    for (size_t op = 0; op < nops; op++) {
        const uint8_t* msg = inputs + off[op];
        size_t len = off[op + 1] - off[op], nwords = (len + 3) / 4, absorbed = 0;
        uint64_t st[12] = {0};
        for (;;) {
            size_t rem = len - absorbed;
            int full = rem >= 32;
            uint8_t block[32] = {0};
            memcpy(block, msg + absorbed, full ? 32 : rem);
#define CELL(c) out[(size_t)(c) * n + row]
            if (full) CELL(0) = 1;
            else {
                if (rem == 31) block[31] = 0x81;
                else { block[rem] = 1; block[31] = 0x80; }
                CELL(14 + rem) = 1;
            }
            CELL(1) = meta[4 * op];
            CELL(2) = meta[4 * op + 1];
            for (size_t i = 0; i < 8; i++) { size_t w = absorbed / 4 + i; CELL(3 + i) = w < nwords ? meta[4 * op + 2] + w : 0; }
            CELL(11) = meta[4 * op + 3];
            CELL(12) = len;
            CELL(13) = absorbed;
            for (int i = 0; i < 12; i++) CELL(46 + i) = st[i];
            for (int i = 0; i < 32; i++) CELL(58 + i) = block[i];
            for (int i = 0; i < 8; i++) {
                st[i] = (uint64_t)block[4 * i] | ((uint64_t)block[4 * i + 1] << 8) | ((uint64_t)block[4 * i + 2] << 16) | ((uint64_t)block[4 * i + 3] << 24);
                CELL(90 + i) = st[i];
            }
            zko_poseidon_permute(st);
            for (int i = 0; i < 8; i++) CELL(98 + i) = st[4 + i];
            for (int i = 0; i < 4; i++) CELL(106 + i) = st[i];
#undef CELL
            row++;
            if (!full) break;
            absorbed += 32;
        }
    }
    return rows;
}

void zko_poseidon_eval_row(const uint64_t* local, const uint64_t* alphas, size_t nalphas, uint64_t* acc_out) {
    b_consumer k;
    memset(&k, 0, sizeof k);
    k.nalphas = nalphas;
    for (size_t j = 0; j < nalphas; j++) k.alphas[j] = alphas[j];
    k.z_last = 1;
    b_eval_poseidon(local, &k);
    for (size_t j = 0; j < nalphas; j++) acc_out[j] = k.acc[j];
}

/* Evaluate table `table_id`'s constraints on the rows of a trace (row r with next row r+1 mod n; transition constraints are
 * switched off on the last row, first/last-row constraints on only there) and report the first nonzero one.
 * Returns the number of constraints per row; *bad_row = -1 if every constraint vanishes on every row. */
long zko_debug_constraints(int table_id, const uint64_t* trace, size_t W, unsigned log_n, long* bad_row, long* bad_index) {
    size_t n = (size_t)1 << log_n;
    gl_t *lv = malloc(W * 8), *nv = malloc(W * 8), *rec = malloc(8192 * 8);
    long count = 0;
    *bad_row = -1; *bad_index = -1;
    for (size_t r = 0; r < n && *bad_row < 0; r++) {
        for (size_t c = 0; c < W; c++) { lv[c] = trace[c * n + r]; nv[c] = trace[c * n + (r + 1) % n]; }
        b_consumer k;
        memset(&k, 0, sizeof k);
        k.z_last = r != n - 1; k.l_first = r == 0; k.l_last = r == n - 1;
        k.rec = rec; k.rec_cap = 8192;
        b_eval_table(table_id, lv, nv, &k);
        count = (long)k.nrec;
        for (size_t i = 0; i < k.nrec && i < 8192; i++)
            if (gl_canon(rec[i]) != 0) { *bad_row = (long)r; *bad_index = (long)i; break; }
    }
    free(lv); free(nv); free(rec);
    return count;
}

/* All constraint values of one row pair (trace-domain flags as in zko_debug_constraints); returns the number of constraints. */
long zko_debug_row_constraints(int table_id, const uint64_t* lv, const uint64_t* nv, int is_first, int is_last, uint64_t* out, size_t cap) {
    b_consumer k;
    memset(&k, 0, sizeof k);
    k.z_last = !is_last; k.l_first = is_first; k.l_last = is_last;
    k.rec = out; k.rec_cap = cap;
    b_eval_table(table_id, lv, nv, &k);
    for (size_t i = 0; i < k.nrec && i < cap; i++) out[i] = gl_canon(out[i]);
    return (long)k.nrec;
}

/* The reference's low-degree test of a table's constraints (prover/src/stark_testing.rs:21-70, test_stark_low_degree), evaluation half:
 * rows = ncols x size evaluations (column-major) of random polynomials of degree < witness_size on the subgroup of order
 * size = witness_size << rate_bits, natural order; out[i] = the alpha-accumulated constraint value at point i with the frame
 * (row i, row i + 2^rate_bits), z_last = w_size^i - w_witness^-1 and the two Lagrange selectors lde'd the same way (:29-52). */
void zko_constraint_evals(int table_id, const uint64_t* rows, size_t ncols, unsigned log_witness, unsigned rate_bits, uint64_t alpha,
                          uint64_t* out) {
    const unsigned log_size = log_witness + rate_bits;
    const size_t size = (size_t)1 << log_size, w = (size_t)1 << log_witness;
    gl_t* lf = (gl_t*)calloc(2 * size, sizeof(gl_t));
    gl_t* ll = lf + size;
    lf[0] = 1; ll[w - 1] = 1;                                      /* PolynomialValues::selector(WITNESS_SIZE, i) */
    zko_ntt(lf, 1, log_witness, 1, 0); zko_ntt(ll, 1, log_witness, 1, 0);
    zko_ntt(lf, 1, log_size, 0, 0); zko_ntt(ll, 1, log_size, 0, 0); /* .lde(rate_bits): zero-padded coefficients, plain subgroup */
    const gl_t last = gl_inv(gl_root_of_unity(log_witness)), ws = gl_root_of_unity(log_size);
    gl_t* lv = (gl_t*)malloc(sizeof(gl_t) * 2 * ncols);
    gl_t* nv = lv + ncols;
    gl_t x = 1;
    for (size_t i = 0; i < size; i++) {
        const size_t in = (i + ((size_t)1 << rate_bits)) % size;
        for (size_t c = 0; c < ncols; c++) { lv[c] = rows[c * size + i]; nv[c] = rows[c * size + in]; }
        b_consumer k;
        memset(&k, 0, sizeof k);
        k.nalphas = 1; k.alphas[0] = alpha;
        k.z_last = gl_sub(x, last); k.l_first = lf[i]; k.l_last = ll[i];
        b_eval_table(table_id, lv, nv, &k);
        out[i] = gl_canon(k.acc[0]);
        x = gl_mul(x, ws);
    }
    free(lv); free(lf);
}

/* ------------------------------------------------------------------ config */
void zko_standard_config(zko_stark_config* c) {
    c->rate_bits = 2; c->cap_height = 4; c->pow_bits = 16; c->num_challenges = 2;
    c->num_queries = 37; c->arity_bits = 4; c->final_poly_bits = 5;
}

/* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) (App. A.8) */
static unsigned fri_layers(const zko_stark_config* c, unsigned degree_bits) {
    unsigned l = 0, d = degree_bits;
    while (d > c->final_poly_bits && d + c->rate_bits - c->arity_bits >= c->cap_height) { d -= c->arity_bits; l++; }
    return l;
}

typedef struct {
    unsigned log_n, lde_bits, L, cap;
    size_t W, A, Q, Z, F, C, nq;
    size_t o_init, o_caps, o_open, o_fri_caps, o_final, o_pow, o_queries, query_words, total;
} layout_t;

static void layout(layout_t* y, const zko_stark_config* c, unsigned log_n, size_t W, size_t A, size_t Z) {
    y->log_n = log_n; y->lde_bits = log_n + c->rate_bits; y->cap = c->cap_height;
    y->W = W; y->A = A; y->Q = c->num_challenges * 2; y->Z = Z;
    y->L = fri_layers(c, log_n);
    y->F = (size_t)1 << (log_n - y->L * c->arity_bits);
    y->C = (size_t)1 << c->cap_height;
    y->nq = c->num_queries;
    size_t o = 16;
    y->o_init = o; o += 12;
    y->o_caps = o; o += 3 * y->C * 4;
    y->o_open = o; o += 4 * W + 4 * A + Z + 2 * y->Q;
    y->o_fri_caps = o; o += y->L * y->C * 4;
    y->o_final = o; o += 2 * y->F;
    y->o_pow = o; o += 1;
    y->o_queries = o;
    size_t sib0 = (size_t)(y->lde_bits - y->cap) * 4;
    size_t q = (W + sib0) + (A + sib0) + (y->Q + sib0);
    for (unsigned i = 0; i < y->L; i++)
        q += 2 * ((size_t)1 << c->arity_bits) + (size_t)(y->lde_bits - c->arity_bits * (i + 1) - y->cap) * 4;
    y->query_words = q;
    y->total = o + q * y->nq;
}

size_t zko_proof_words(const zko_stark_config* c, unsigned log_n, size_t W, size_t A, size_t Z) {
    layout_t y;
    layout(&y, c, log_n, W, A, Z);
    return y.total;
}

/* ------------------------------------------------------------------ quotient (prover.rs:645-789) */
/* fake-CTL shape (poseidon_benchmark): CtlZData with helper columns and no column sets */
static zko_ctl_z* fake_zs(const uint32_t* num_helpers, size_t nctl) {
    zko_ctl_z* zs = (zko_ctl_z*)calloc(nctl ? nctl : 1, sizeof(zko_ctl_z));
    for (size_t i = 0; i < nctl; i++) zs[i].num_helpers = num_helpers[i];
    return zs;
}
static const zko_ctl_table EMPTY_CTL_TABLE = {0};

static void quotient_generic(int table_id, const zko_batch* trace, const zko_batch* aux, const zko_ctl_table* ctl_t, const zko_ctl_z* zs,
                             const uint32_t* colset_ids, size_t nctl, const uint64_t* lookup_ch, const uint64_t* alphas, size_t nalphas,
                             uint64_t* out);

void zko_quotient_poseidon(const zko_batch* trace, const zko_batch* aux, const uint32_t* num_helpers, size_t nctl,
                           const uint64_t* alphas, size_t nalphas, uint64_t* out) {
    zko_ctl_z* zs = fake_zs(num_helpers, nctl);
    quotient_generic(ZKO_TABLE_POSEIDON, trace, aux, &EMPTY_CTL_TABLE, zs, NULL, nctl, NULL, alphas, nalphas, out);
    free(zs);
}

void zko_quotient(int table_id, const zko_batch* trace, const zko_batch* aux, const uint32_t* num_helpers, size_t nctl,
                  const uint64_t* alphas, size_t nalphas, uint64_t* out) {
    zko_ctl_z* zs = fake_zs(num_helpers, nctl);
    quotient_generic(table_id, trace, aux, &EMPTY_CTL_TABLE, zs, NULL, nctl, NULL, alphas, nalphas, out);
    free(zs);
}

static void quotient_generic(int table_id, const zko_batch* trace, const zko_batch* aux, const zko_ctl_table* ctl_t, const zko_ctl_z* zs,
                             const uint32_t* colset_ids, size_t nctl, const uint64_t* lookup_ch, const uint64_t* alphas, size_t nalphas,
                             uint64_t* out) {
    /* auxiliary columns = lookup helper columns, then CTL helper columns, then CTL Zs (prover.rs:495-508) */
    const size_t NL = lookup_ch ? zko_table_num_lookup_columns(table_id, nalphas) : 0;
    unsigned log_n = zko_batch_log_n(trace), rate_bits = 2, qbits = 1;
    unsigned log_N = log_n + rate_bits, log_q = log_n + qbits;
    size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N, size = (size_t)1 << log_q;
    size_t W = zko_batch_ncols(trace), A = zko_batch_ncols(aux);
    size_t step = (size_t)1 << (rate_bits - qbits), next_step = (size_t)1 << qbits;
    const gl_t *tl = zko_batch_lde_ptr(trace), *al = zko_batch_lde_ptr(aux);

    /* Lagrange first / last as LDEs of selector polynomials (prover.rs:678-681) */
    gl_t* lf = (gl_t*)calloc(size, sizeof(gl_t));
    gl_t* ll = (gl_t*)calloc(size, sizeof(gl_t));
    lf[0] = 1;
    ll[n - 1] = 1;
    zko_ntt(lf, 1, log_n, 1, 0);
    zko_ntt(ll, 1, log_n, 1, 0);
    zko_ntt(lf, 1, log_q, 0, GL_GENERATOR); /* upper half already zero == lde(1) */
    zko_ntt(ll, 1, log_q, 0, GL_GENERATOR);

    /* ZeroPolyOnCoset (App. A.10): Z_H(g w_2n^i) = g^n (-1)^i - 1 */
    gl_t gn = gl_exp_pow2(GL_GENERATOR, log_n);
    gl_t zh_inv[2] = {gl_inv(gl_sub(gn, 1)), gl_inv(gl_sub(gl_neg(gn), 1))};
    gl_t last = gl_inv(gl_root_of_unity(log_n));
    gl_t wq = gl_root_of_unity(log_q);

    gl_t* qv = (gl_t*)malloc(sizeof(gl_t) * nalphas * size);
#pragma omp parallel
    {
        gl_t* lv = (gl_t*)malloc(sizeof(gl_t) * (2 * W + 2 * A));
        gl_t *av = lv + W, *an = av + A, *nv = an + A;
#pragma omp for schedule(static)
        for (size_t i = 0; i < size; i++) {
            size_t i_next = (i + next_step) % size;
            size_t j = bitrev(i * step, log_N), jn = bitrev(i_next * step, log_N);
            gl_t x = gl_mul(GL_GENERATOR, gl_pow(wq, i));
            b_consumer k;
            memset(&k, 0, sizeof k);
            k.nalphas = nalphas;
            for (size_t a = 0; a < nalphas; a++) k.alphas[a] = alphas[a];
            k.z_last = gl_sub(x, last);
            k.l_first = lf[i];
            k.l_last = ll[i];
            for (size_t c = 0; c < W; c++) { lv[c] = tl[c * N + j]; nv[c] = tl[c * N + jn]; }
            for (size_t c = 0; c < A; c++) { av[c] = al[c * N + j]; an[c] = al[c * N + jn]; }
            b_eval_table(table_id, lv, nv, &k);
            if (NL) b_eval_lookups(table_id, lookup_ch, nalphas, lv, av, an, &k);
            b_eval_ctl_general(ctl_t, zs, colset_ids, nctl, lv, nv, av + NL, an + NL, &k);
            for (size_t a = 0; a < nalphas; a++) qv[a * size + i] = gl_mul(k.acc[a], zh_inv[i & 1]);
        }
        free(lv);
    }
    zko_ntt(qv, nalphas, log_q, 1, GL_GENERATOR);
    memcpy(out, qv, sizeof(gl_t) * nalphas * size);
    free(qv);
    free(lf);
    free(ll);
}

/* ------------------------------------------------------------------ helpers */
static gl2_t eval_poly_ext(const gl_t* coeffs, size_t n, gl2_t z) {
    gl2_t acc = gl2_from_base(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_from_base(coeffs[i]));
    return acc;
}
static gl2_t eval_ext_poly_ext(const gl2_t* coeffs, size_t n, gl2_t z) {
    gl2_t acc = gl2_from_base(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), coeffs[i]);
    return acc;
}

/* F2 coset NTT: natural coeffs (len 2^log) -> natural values, shift in base field */
static void ext_coset_fft(gl2_t* a, unsigned log, gl_t shift) {
    size_t m = (size_t)1 << log;
    gl_t* t = (gl_t*)malloc(sizeof(gl_t) * 2 * m);
    for (size_t i = 0; i < m; i++) { t[i] = a[i].c[0]; t[m + i] = a[i].c[1]; }
    zko_ntt(t, 2, log, 0, shift == 1 ? 0 : shift);
    for (size_t i = 0; i < m; i++) a[i] = gl2_make(t[i], t[m + i]);
    free(t);
}

typedef struct {
    void* tree;
    gl2_t* leaves; /* values in bit-reversed order; leaf k = leaves[16k .. 16k+16) */
    unsigned log_leaves;
} fri_layer_t;

/* ------------------------------------------------------------------ prove_single_table */
static int prove_generic(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                         const uint64_t* aux, size_t A, const zko_ctl_table* ctl_t, const zko_ctl_z* zs, const uint32_t* colset_ids,
                         size_t Z, const uint64_t* lookup_ch, zko_challenger* ch, uint64_t* proof, double* stage_s);
static int prove_generic_ex(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                            const uint64_t* aux, size_t A, const zko_ctl_table* ctl_t, const zko_ctl_z* zs, const uint32_t* colset_ids,
                            size_t Z, const uint64_t* lookup_ch, zko_challenger* ch, uint64_t* proof, double* stage_s, zko_batch* tb_in,
                            zko_batch* ab_in, zko_batch* qb_in);

/* prove_openings alone on three existing commitments (BASELINE config 4): compact, zeta, openings, FRI */
int zko_prove_openings(const zko_stark_config* cfg, zko_batch* tb, zko_batch* ab, zko_batch* qb, size_t Z, zko_challenger* ch,
                       uint64_t* proof) {
    return prove_generic_ex(-1, cfg, NULL, zko_batch_ncols(tb), zko_batch_log_n(tb), NULL, zko_batch_ncols(ab), NULL, NULL, NULL, Z, NULL,
                            ch, proof, NULL, tb, ab, qb);
}

int zko_prove_single_table(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                           const uint64_t* aux, size_t A, const uint32_t* num_helpers, size_t Z,
                           zko_challenger* ch, uint64_t* proof, double* stage_s) {
    for (size_t i = 0; i < Z; i++) if (!num_helpers[i]) return -2;
    zko_ctl_z* zs = fake_zs(num_helpers, Z);
    int rc = prove_generic(table_id, cfg, trace, W, log_n, aux, A, &EMPTY_CTL_TABLE, zs, NULL, Z, NULL, ch, proof, stage_s);
    free(zs);
    return rc;
}
int zko_prove_single_table_ctl(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                               const uint64_t* aux, size_t A, const zko_ctl_table* t, const zko_ctl_z* zs,
                               const uint32_t* colset_ids, size_t Z, const uint64_t* lookup_challenges, zko_challenger* ch,
                               uint64_t* proof) {
    return prove_generic(table_id, cfg, trace, W, log_n, aux, A, t, zs, colset_ids, Z, lookup_challenges, ch, proof, NULL);
}
size_t zko_num_lookup_columns(int table_id, const zko_stark_config* cfg) { return zko_table_num_lookup_columns(table_id, cfg->num_challenges); }

static int prove_generic(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                         const uint64_t* aux, size_t A, const zko_ctl_table* ctl_t, const zko_ctl_z* zs, const uint32_t* colset_ids,
                         size_t Z, const uint64_t* lookup_ch, zko_challenger* ch, uint64_t* proof, double* stage_s) {
    return prove_generic_ex(table_id, cfg, trace, W, log_n, aux, A, ctl_t, zs, colset_ids, Z, lookup_ch, ch, proof, stage_s, NULL, NULL, NULL);
}

/* lookup_helper_columns (lookup.rs:46-124) for Column::single columns without filters: per pair h = 1/(f0 + x) + 1/(f1 + x),
 * then Z with Z[0] = 0, Z[i+1] = Z[i] + sum_h(i) - freq(i) / (table(i) + x).  out = (ceil(ncols/2) + 1) columns of n. */
static void lookup_columns_simple(const zko_lookup_def* d, gl_t x, const uint64_t* trace, size_t n, gl_t* out) {
    size_t nh = (d->ncols + 1) / 2;
    memset(out, 0, sizeof(gl_t) * nh * n);
    for (size_t q = 0; q < d->ncols; q++) {
        const uint64_t* col = trace + (size_t)d->cols[q] * n;
        gl_t* h = out + (q / 2) * n;
        for (size_t i = 0; i < n; i++) h[i] = gl_add(h[i], gl_inv(gl_add(col[i], x)));
    }
    gl_t* z = out + nh * n;
    const uint64_t *tab = trace + (size_t)d->table_col * n, *freq = trace + (size_t)d->freq_col * n;
    z[0] = 0;
    for (size_t i = 0; i + 1 < n; i++) {
        gl_t acc = 0;
        for (size_t q = 0; q < nh; q++) acc = gl_add(acc, out[q * n + i]);
        acc = gl_sub(acc, gl_mul(freq[i], gl_inv(gl_add(tab[i], x))));
        z[i + 1] = gl_add(z[i], acc);
    }
}

static int prove_generic_ex(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                            const uint64_t* aux_ctl, size_t A_ctl, const zko_ctl_table* ctl_t, const zko_ctl_z* zs, const uint32_t* colset_ids,
                            size_t Z, const uint64_t* lookup_ch, zko_challenger* ch, uint64_t* proof, double* stage_s, zko_batch* tb_in,
                            zko_batch* ab_in, zko_batch* qb_in) {
    const int openings_only = tb_in != NULL;
    if (!openings_only && (b_table_width(table_id) == 0 || (size_t)b_table_width(table_id) != W)) return -1;
    if (cfg->num_challenges > 4) return -1;
    /* lookup helper columns come first among the auxiliary polynomials (prover.rs:467-508) */
    const size_t NL = openings_only ? 0 : zko_table_num_lookup_columns(table_id, cfg->num_challenges);
    if (NL && !lookup_ch) return -5;
    if (!NL) lookup_ch = NULL;
    const size_t A = NL + A_ctl;
    layout_t y;
    layout(&y, cfg, log_n, W, A, Z);
    size_t n = (size_t)1 << log_n, N = (size_t)1 << y.lde_bits;
    size_t total_helpers = 0; /* index of the first CTL Z among the auxiliary polynomials */
    if (openings_only) {
        if (Z > A || zko_batch_ncols(qb_in) != y.Q || zko_batch_log_n(ab_in) != log_n || zko_batch_log_n(qb_in) != log_n) return -3;
    } else {
        size_t ctl_helpers = 0;
        for (size_t i = 0; i < Z; i++) ctl_helpers += zs[i].num_helpers;
        if (ctl_helpers + Z != A_ctl) return -3;
    }
    total_helpers = A - Z;
    const uint64_t* aux = aux_ctl;
    uint64_t* aux_all = NULL;
    if (NL) {
        aux_all = (uint64_t*)malloc(sizeof(uint64_t) * A * n);
        size_t nl, off = 0;
        const zko_lookup_def* defs = zko_table_lookups(table_id, &nl);
        for (size_t l = 0; l < nl; l++)
            for (unsigned c = 0; c < cfg->num_challenges; c++) {
                lookup_columns_simple(&defs[l], lookup_ch[c], trace, n, aux_all + off * n);
                off += (defs[l].ncols + 1) / 2 + 1;
            }
        memcpy(aux_all + NL * n, aux_ctl, sizeof(uint64_t) * A_ctl * n);
        aux = aux_all;
    }
    double t0, ts[8] = {0};

    memset(proof, 0, sizeof(uint64_t) * y.total);
    proof[0] = 0x5a4b4d50524f4f46ULL; proof[1] = log_n; proof[2] = W; proof[3] = A; proof[4] = y.Q; proof[5] = Z;
    proof[6] = y.cap; proof[7] = y.L; proof[8] = y.F; proof[9] = y.nq; proof[10] = cfg->rate_bits; proof[11] = cfg->arity_bits;

    zko_batch *tb = tb_in, *ab = ab_in, *qb = qb_in;
    uint64_t* caps = proof + y.o_caps;
    if (!openings_only) {
        /* trace commitment (done by the caller in the reference, prover.rs:144-167 / poseidon_stark.rs:766-777) */
        t0 = now_s();
        tb = zko_batch_from_values(trace, W, log_n, cfg->rate_bits, cfg->cap_height);
        ts[0] = now_s() - t0;
    }
    zko_challenger_compact(ch, proof + y.o_init);                                   /* :466 */
    zko_batch_cap(tb, caps);
    if (!openings_only) {
        t0 = now_s();
        ab = zko_batch_from_values(aux, A, log_n, cfg->rate_bits, cfg->cap_height); /* :511-522 */
        free(aux_all);
        ts[1] = now_s() - t0;
        zko_batch_cap(ab, caps + y.C * 4);
        zko_challenger_observe(ch, caps + y.C * 4, y.C * 4);                            /* :525 */
        gl_t alphas[4];
        for (unsigned i = 0; i < cfg->num_challenges; i++) alphas[i] = zko_challenger_get(ch); /* :527 */

        t0 = now_s();
        gl_t* quot = (gl_t*)malloc(sizeof(gl_t) * cfg->num_challenges * 2 * n);
        quotient_generic(table_id, tb, ab, ctl_t, zs, colset_ids, Z, lookup_ch, alphas, cfg->num_challenges, quot);  /* :543-559 */
        ts[2] = now_s() - t0;
        /* chunks of n coefficients: [q0_lo, q0_hi, q1_lo, q1_hi] == quot viewed as Q columns of n (:560-575) */
        t0 = now_s();
        qb = zko_batch_from_coeffs(quot, y.Q, log_n, cfg->rate_bits, cfg->cap_height); /* :576-587 */
        ts[3] = now_s() - t0;
        free(quot);
        zko_batch_cap(qb, caps + 2 * y.C * 4);
        zko_challenger_observe(ch, caps + 2 * y.C * 4, y.C * 4);                        /* :589 */
    } else {
        zko_batch_cap(ab, caps + y.C * 4);
        zko_batch_cap(qb, caps + 2 * y.C * 4);
    }

    gl2_t zeta = challenger_get_ext(ch);                                            /* :591 */
    gl_t g = gl_root_of_unity(log_n);
    if (gl2_eq(gl2_exp_pow2(zeta, log_n), gl2_from_base(1))) return -4;             /* :596-599 */
    gl2_t zeta_next = gl2_scalar_mul(zeta, g);

    /* openings (proof.rs:299-334) */
    t0 = now_s();
    const gl_t *tc = zko_batch_coeffs_ptr(tb), *ac = zko_batch_coeffs_ptr(ab), *qc = zko_batch_coeffs_ptr(qb);
    uint64_t* op = proof + y.o_open;
    uint64_t *o_local = op, *o_next = op + 2 * W, *o_aux = op + 4 * W, *o_auxn = o_aux + 2 * A, *o_ctl = o_auxn + 2 * A,
             *o_quot = o_ctl + Z;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < W + A + y.Q; c++) {
        if (c < W) {
            gl2_t a = eval_poly_ext(tc + c * n, n, zeta), b = eval_poly_ext(tc + c * n, n, zeta_next);
            o_local[2 * c] = a.c[0]; o_local[2 * c + 1] = a.c[1];
            o_next[2 * c] = b.c[0]; o_next[2 * c + 1] = b.c[1];
        } else if (c < W + A) {
            size_t k = c - W;
            gl2_t a = eval_poly_ext(ac + k * n, n, zeta), b = eval_poly_ext(ac + k * n, n, zeta_next);
            o_aux[2 * k] = a.c[0]; o_aux[2 * k + 1] = a.c[1];
            o_auxn[2 * k] = b.c[0]; o_auxn[2 * k + 1] = b.c[1];
            if (k >= total_helpers) { /* eval at 1 = sum of coefficients */
                gl_t s = 0;
                for (size_t i = 0; i < n; i++) s = gl_add(s, ac[k * n + i]);
                o_ctl[k - total_helpers] = s;
            }
        } else {
            size_t k = c - W - A;
            gl2_t a = eval_poly_ext(qc + k * n, n, zeta);
            o_quot[2 * k] = a.c[0]; o_quot[2 * k + 1] = a.c[1];
        }
    }
    ts[4] = now_s() - t0;
    /* observe_openings(to_fri_openings) (proof.rs:336-367): zeta batch, zeta_next batch, ctl_zs_first as F2 */
    zko_challenger_observe(ch, o_local, 2 * W);
    zko_challenger_observe(ch, o_aux, 2 * A);
    zko_challenger_observe(ch, o_quot, 2 * y.Q);
    zko_challenger_observe(ch, o_next, 2 * W);
    zko_challenger_observe(ch, o_auxn, 2 * A);
    for (size_t i = 0; i < Z; i++) { uint64_t e[2] = {o_ctl[i], 0}; zko_challenger_observe(ch, e, 2); }

    /* ---- prove_openings (App. A.8) ---- */
    t0 = now_s();
    gl2_t alpha = challenger_get_ext(ch);
    gl2_t* fin = (gl2_t*)calloc(N, sizeof(gl2_t)); /* final poly, zero padded to 4n == lde(rate_bits) */
    gl2_t* comp = (gl2_t*)malloc(sizeof(gl2_t) * n);
    for (int batch = 0; batch < 3; batch++) {
        gl2_t point = batch == 0 ? zeta : batch == 1 ? zeta_next : gl2_from_base(1);
        /* polynomial list of this batch (stark.rs:127-148) */
        size_t np = batch == 0 ? W + A + y.Q : batch == 1 ? W + A : Z;
        const gl_t** polys = (const gl_t**)malloc(sizeof(gl_t*) * np);
        size_t k = 0;
        if (batch < 2) {
            for (size_t c = 0; c < W; c++) polys[k++] = tc + c * n;
            for (size_t c = 0; c < A; c++) polys[k++] = ac + c * n;
            if (batch == 0) for (size_t c = 0; c < y.Q; c++) polys[k++] = qc + c * n;
        } else {
            for (size_t c = total_helpers; c < A; c++) polys[k++] = ac + c * n;
        }
        gl2_t* apow = (gl2_t*)malloc(sizeof(gl2_t) * (np + 1));
        apow[0] = gl2_from_base(1);
        for (size_t j = 1; j <= np; j++) apow[j] = gl2_mul(apow[j - 1], alpha);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) {
            gl2_t acc = gl2_from_base(0);
            for (size_t j = 0; j < np; j++) acc = gl2_add(acc, gl2_scalar_mul(apow[j], polys[j][i]));
            comp[i] = acc;
        }
        /* divide_by_linear(point): q_{k-1} = acc_k, acc_k = acc_{k+1} * z + a_k, remainder dropped, re-padded */
        gl2_t shiftp = apow[np];
        gl2_t acc = gl2_from_base(0);
        for (size_t i = n; i-- > 0;) {
            acc = gl2_add(gl2_mul(acc, point), comp[i]);
            if (i > 0) fin[i - 1] = gl2_add(gl2_mul(fin[i - 1], shiftp), acc);
        }
        fin[n - 1] = gl2_mul(fin[n - 1], shiftp); /* padded zero coefficient of the quotient */
        free(apow);
        free(polys);
    }
    free(comp);
    /* lde_final_values = coset_fft(g) of the zero-padded final poly */
    gl2_t* coeffs = (gl2_t*)malloc(sizeof(gl2_t) * N);
    memcpy(coeffs, fin, sizeof(gl2_t) * N);
    gl2_t* values = fin;
    ext_coset_fft(values, y.lde_bits, GL_GENERATOR);
    ts[5] = now_s() - t0;

    /* fri_committed_trees */
    t0 = now_s();
    fri_layer_t* layers = (fri_layer_t*)calloc(y.L ? y.L : 1, sizeof(fri_layer_t));
    gl_t shift = GL_GENERATOR;
    size_t cur = N;
    unsigned cur_bits = y.lde_bits;
    size_t arity = (size_t)1 << cfg->arity_bits;
    for (unsigned l = 0; l < y.L; l++) {
        gl2_t* lv = (gl2_t*)malloc(sizeof(gl2_t) * cur);
        for (size_t i = 0; i < cur; i++) lv[i] = values[bitrev(i, cur_bits)];
        layers[l].leaves = lv;
        layers[l].log_leaves = cur_bits - cfg->arity_bits;
        layers[l].tree = zko_merkle_from_rows((const uint64_t*)lv, layers[l].log_leaves, 2 * arity, cfg->cap_height);
        uint64_t* capo = proof + y.o_fri_caps + l * y.C * 4;
        zko_merkle_cap(layers[l].tree, capo);
        zko_challenger_observe(ch, capo, y.C * 4);
        gl2_t beta = challenger_get_ext(ch);
        size_t nxt = cur >> cfg->arity_bits;
        for (size_t j = 0; j < nxt; j++) { /* reduce_with_powers(chunk, beta) */
            gl2_t acc = gl2_from_base(0);
            for (size_t i = arity; i-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[j * arity + i]);
            coeffs[j] = acc;
        }
        shift = gl_pow(shift, arity);
        cur = nxt;
        cur_bits -= cfg->arity_bits;
        memcpy(values, coeffs, sizeof(gl2_t) * cur);
        ext_coset_fft(values, cur_bits, shift);
    }
    size_t flen = cur >> cfg->rate_bits;
    if (flen != y.F) return -5;
    for (size_t i = flen; i < cur; i++) if (coeffs[i].c[0] | coeffs[i].c[1]) return -6; /* "should always be zero" */
    uint64_t* fp = proof + y.o_final;
    for (size_t i = 0; i < flen; i++) { fp[2 * i] = coeffs[i].c[0]; fp[2 * i + 1] = coeffs[i].c[1]; }
    zko_challenger_observe(ch, fp, 2 * flen);

    /* proof of work (App. A.9): smallest witness */
    {
        uint64_t st[12], w = 0;
        for (;; w++) {
            memcpy(st, ch->state, sizeof st);
            for (uint32_t i = 0; i < ch->n_in; i++) st[i] = ch->in_buf[i];
            st[ch->n_in] = w;
            zko_poseidon_permute(st);
            if ((st[7] >> (64 - cfg->pow_bits)) == 0) break;
        }
        proof[y.o_pow] = w;
        zko_challenger_observe(ch, &w, 1);
        uint64_t resp = zko_challenger_get(ch);
        if ((resp >> (64 - cfg->pow_bits)) != 0) return -7;
    }
    ts[6] = now_s() - t0;

    /* query rounds */
    t0 = now_s();
    const zko_batch* oracles[3] = {tb, ab, qb};
    for (size_t q = 0; q < y.nq; q++) {
        size_t x = zko_challenger_get(ch) % N;
        uint64_t* o = proof + y.o_queries + q * y.query_words;
        for (int k = 0; k < 3; k++) {
            size_t nc = zko_batch_ncols(oracles[k]);
            zko_batch_leaf(oracles[k], x, o);
            o += nc;
            zko_batch_merkle_path(oracles[k], x, o);
            o += (size_t)(y.lde_bits - y.cap) * 4;
        }
        for (unsigned l = 0; l < y.L; l++) {
            x >>= cfg->arity_bits;
            memcpy(o, layers[l].leaves + x * arity, sizeof(gl2_t) * arity);
            o += 2 * arity;
            zko_merkle_path(layers[l].tree, x, o);
            o += (size_t)(layers[l].log_leaves - y.cap) * 4;
        }
    }
    ts[7] = now_s() - t0;

    for (unsigned l = 0; l < y.L; l++) { zko_merkle_free(layers[l].tree); free(layers[l].leaves); }
    free(layers);
    free(coeffs);
    free(fin);
    if (!openings_only) {
        zko_batch_free(tb);
        zko_batch_free(ab);
        zko_batch_free(qb);
    }
    if (stage_s) memcpy(stage_s, ts, sizeof ts);
    if (getenv("ZKO_TIMING"))
        fprintf(stderr, "[oracle] table %d 2^%u: trace %.2f aux %.2f quotient %.2f qcommit %.2f open %.2f combine %.2f fri+pow %.2f queries %.2f s\n",
                table_id, log_n, ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], ts[6], ts[7]);
    return 0;
}

/* ------------------------------------------------------------------ verifier */
static int merkle_verify(const uint64_t* leaf, size_t leaf_len, size_t index, const uint64_t* cap, const uint64_t* sib,
                         size_t nsib) {
    uint64_t cur[4], nx[4];
    zko_poseidon_hash_or_noop(leaf, leaf_len, cur);
    for (size_t s = 0; s < nsib; s++) {
        if (index & 1) zko_poseidon_two_to_one(sib + 4 * s, cur, nx);
        else zko_poseidon_two_to_one(cur, sib + 4 * s, nx);
        memcpy(cur, nx, sizeof cur);
        index >>= 1;
    }
    return memcmp(cur, cap + 4 * index, 32) == 0;
}

static gl2_t reduce_ext(const gl2_t* v, size_t n, gl2_t alpha) { /* sum_j alpha^j v_j */
    gl2_t acc = gl2_from_base(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), v[i]);
    return acc;
}

/* plonky2 fri/verifier.rs compute_evaluation: interpolate the arity points of a coset, evaluate at beta */
static gl2_t compute_evaluation(gl_t x, size_t x_in_coset, unsigned arity_bits, const gl2_t* evals_in, gl2_t beta) {
    size_t arity = (size_t)1 << arity_bits;
    gl_t g = gl_root_of_unity(arity_bits);
    gl2_t evals[64];
    for (size_t i = 0; i < arity; i++) evals[bitrev(i, arity_bits)] = evals_in[i];
    size_t rev = bitrev(x_in_coset, arity_bits);
    gl_t start = gl_mul(x, gl_pow(g, arity - rev));
    gl_t pts[64];
    gl_t p = start;
    for (size_t i = 0; i < arity; i++) { pts[i] = p; p = gl_mul(p, g); }
    /* Lagrange interpolation at beta */
    gl2_t acc = gl2_from_base(0);
    for (size_t i = 0; i < arity; i++) {
        gl2_t num = evals[i];
        gl_t den = 1;
        for (size_t j = 0; j < arity; j++) {
            if (j == i) continue;
            num = gl2_mul(num, gl2_sub(beta, gl2_from_base(pts[j])));
            den = gl_mul(den, gl_sub(pts[i], pts[j]));
        }
        acc = gl2_add(acc, gl2_scalar_mul(num, gl_inv(den)));
    }
    return acc;
}

static int verify_generic(int table_id, const zko_stark_config* cfg, const uint64_t* proof, size_t W, size_t A,
                          const zko_ctl_table* ctl_t, const zko_ctl_z* zs, const uint32_t* colset_ids, size_t Z, const uint64_t* lookup_ch,
                          zko_challenger* ch);

/* verifier of zko_prove_openings: transcript replay + verify_fri_proof, no constraint check */
int zko_verify_openings(const zko_stark_config* cfg, const uint64_t* proof, size_t W, size_t A, size_t Z, zko_challenger* ch) {
    return verify_generic(-1, cfg, proof, W, A, NULL, NULL, NULL, Z, NULL, ch);
}

int zko_verify_single_table(int table_id, const zko_stark_config* cfg, const uint64_t* proof, size_t W, size_t A,
                            const uint32_t* num_helpers, size_t Z, zko_challenger* ch) {
    zko_ctl_z* zs = fake_zs(num_helpers, Z);
    int rc = verify_generic(table_id, cfg, proof, W, A, &EMPTY_CTL_TABLE, zs, NULL, Z, NULL, ch);
    free(zs);
    return rc;
}
int zko_verify_single_table_ctl(int table_id, const zko_stark_config* cfg, const uint64_t* proof, size_t W, size_t A,
                                const zko_ctl_table* t, const zko_ctl_z* zs, const uint32_t* colset_ids, size_t Z,
                                const uint64_t* lookup_challenges, zko_challenger* ch) {
    return verify_generic(table_id, cfg, proof, W, A, t, zs, colset_ids, Z, lookup_challenges, ch);
}

static int verify_generic(int table_id, const zko_stark_config* cfg, const uint64_t* proof, size_t W, size_t A_ctl,
                          const zko_ctl_table* ctl_t, const zko_ctl_z* zs, const uint32_t* colset_ids, size_t Z, const uint64_t* lookup_ch,
                          zko_challenger* ch) {
    const int openings_only = table_id < 0;
    const size_t NL = openings_only ? 0 : zko_table_num_lookup_columns(table_id, cfg->num_challenges);
    if (NL && !lookup_ch) return 1;
    const size_t A = NL + A_ctl;
    if (!openings_only && (b_table_width(table_id) == 0 || (size_t)b_table_width(table_id) != W)) return 1;
    if (proof[0] != 0x5a4b4d50524f4f46ULL) return 2;
    unsigned log_n = (unsigned)proof[1];
    layout_t y;
    layout(&y, cfg, log_n, W, A, Z);
    if (proof[2] != W || proof[3] != A || proof[4] != y.Q || proof[5] != Z || proof[6] != y.cap || proof[7] != y.L ||
        proof[8] != y.F || proof[9] != y.nq)
        return 3; /* validate_proof_shape verifier.rs:294-342 */
    size_t N = (size_t)1 << y.lde_bits;
    size_t total_helpers = A - Z; /* index of the first CTL Z */
    if (!openings_only) {
        size_t ctl_helpers = 0;
        for (size_t i = 0; i < Z; i++) ctl_helpers += zs[i].num_helpers;
        if (ctl_helpers + Z != A_ctl) return 3;
    }
    size_t arity = (size_t)1 << cfg->arity_bits;

    /* the verifier starts from the recorded transcript state (proof.rs:199): the prover's challenger
     * was compacted at that point, so state == init state with empty buffers. */
    uint64_t st0[12];
    zko_challenger_compact(ch, st0);
    if (memcmp(st0, proof + y.o_init, sizeof st0)) return 4;

    const uint64_t* caps = proof + y.o_caps;
    /* get_challenges.rs:190-233 */
    gl_t alphas[4] = {0};
    if (!openings_only) {
        zko_challenger_observe(ch, caps + y.C * 4, y.C * 4);
        for (unsigned i = 0; i < cfg->num_challenges; i++) alphas[i] = zko_challenger_get(ch);
        zko_challenger_observe(ch, caps + 2 * y.C * 4, y.C * 4);
    }
    gl2_t zeta = challenger_get_ext(ch);
    const uint64_t* op = proof + y.o_open;
    const uint64_t *o_local = op, *o_next = op + 2 * W, *o_aux = op + 4 * W, *o_auxn = o_aux + 2 * A, *o_ctl = o_auxn + 2 * A,
                   *o_quot = o_ctl + Z;
    zko_challenger_observe(ch, o_local, 2 * W);
    zko_challenger_observe(ch, o_aux, 2 * A);
    zko_challenger_observe(ch, o_quot, 2 * y.Q);
    zko_challenger_observe(ch, o_next, 2 * W);
    zko_challenger_observe(ch, o_auxn, 2 * A);
    for (size_t i = 0; i < Z; i++) { uint64_t e[2] = {o_ctl[i], 0}; zko_challenger_observe(ch, e, 2); }
    /* fri_challenges */
    gl2_t fri_alpha = challenger_get_ext(ch);
    gl2_t betas[16];
    for (unsigned l = 0; l < y.L; l++) {
        zko_challenger_observe(ch, proof + y.o_fri_caps + l * y.C * 4, y.C * 4);
        betas[l] = challenger_get_ext(ch);
    }
    zko_challenger_observe(ch, proof + y.o_final, 2 * y.F);
    zko_challenger_observe(ch, proof + y.o_pow, 1);
    uint64_t pow_resp = zko_challenger_get(ch);
    size_t xs[256];
    for (size_t q = 0; q < y.nq; q++) xs[q] = zko_challenger_get(ch) % N;

    /* ---- constraint check at zeta (verifier.rs:205-264) ---- */
    gl2_t *lv = (gl2_t*)malloc(sizeof(gl2_t) * (2 * W + 2 * A)), *av = lv + W, *an = av + A, *nv = an + A;
    for (size_t c = 0; c < W; c++) { lv[c] = gl2_make(o_local[2 * c], o_local[2 * c + 1]); nv[c] = gl2_make(o_next[2 * c], o_next[2 * c + 1]); }
    for (size_t c = 0; c < A; c++) { av[c] = gl2_make(o_aux[2 * c], o_aux[2 * c + 1]); an[c] = gl2_make(o_auxn[2 * c], o_auxn[2 * c + 1]); }
    gl_t g = gl_root_of_unity(log_n);
    gl2_t zeta_n = gl2_exp_pow2(zeta, log_n);
    gl2_t z_h = gl2_sub(zeta_n, gl2_from_base(1));
    /* eval_l_0_and_l_last verifier.rs:344-354 */
    gl_t nn = (gl_t)(((uint64_t)1 << log_n) % GL_P);
    gl2_t d0 = gl2_scalar_mul(gl2_sub(zeta, gl2_from_base(1)), nn);
    gl2_t d1 = gl2_scalar_mul(gl2_sub(gl2_scalar_mul(zeta, g), gl2_from_base(1)), nn);
    e_consumer k;
    memset(&k, 0, sizeof k);
    k.nalphas = cfg->num_challenges;
    for (unsigned i = 0; i < cfg->num_challenges; i++) k.alphas[i] = alphas[i];
    k.z_last = gl2_sub(zeta, gl2_from_base(gl_inv(g)));
    k.l_first = gl2_mul(z_h, gl2_inv(d0));
    k.l_last = gl2_mul(z_h, gl2_inv(d1));
    if (!openings_only) {
        e_eval_table(table_id, lv, nv, &k);
        if (NL) e_eval_lookups(table_id, lookup_ch, cfg->num_challenges, lv, av, an, &k);
        e_eval_ctl_general(ctl_t, zs, colset_ids, Z, lv, nv, av + NL, an + NL, &k);
    }
    for (unsigned i = 0; i < cfg->num_challenges && !openings_only; i++) {
        gl2_t t0 = gl2_make(o_quot[4 * i], o_quot[4 * i + 1]), t1 = gl2_make(o_quot[4 * i + 2], o_quot[4 * i + 3]);
        gl2_t rhs = gl2_mul(z_h, gl2_add(t0, gl2_mul(t1, zeta_n)));
        if (!gl2_eq(k.acc[i], rhs)) { free(lv); return 10; }
    }

    /* ---- FRI (App. A.8 verifier) ---- */
    if ((pow_resp >> (64 - cfg->pow_bits)) != 0) { free(lv); return 11; }
    /* precomputed reduced openings per batch */
    size_t nb[3] = {W + A + y.Q, W + A, Z};
    gl2_t* bvals = (gl2_t*)malloc(sizeof(gl2_t) * (W + A + y.Q));
    gl2_t red_open[3];
    for (int b = 0; b < 3; b++) {
        size_t t = 0;
        if (b == 0) {
            for (size_t c = 0; c < W; c++) bvals[t++] = lv[c];
            for (size_t c = 0; c < A; c++) bvals[t++] = av[c];
            for (size_t c = 0; c < y.Q; c++) bvals[t++] = gl2_make(o_quot[2 * c], o_quot[2 * c + 1]);
        } else if (b == 1) {
            for (size_t c = 0; c < W; c++) bvals[t++] = gl2_make(o_next[2 * c], o_next[2 * c + 1]);
            for (size_t c = 0; c < A; c++) bvals[t++] = an[c];
        } else {
            for (size_t c = 0; c < Z; c++) bvals[t++] = gl2_from_base(o_ctl[c]);
        }
        red_open[b] = reduce_ext(bvals, nb[b], fri_alpha);
    }
    gl2_t points[3] = {zeta, gl2_scalar_mul(zeta, g), gl2_from_base(1)};
    size_t sib0 = (size_t)(y.lde_bits - y.cap);
    int rc = 0;
    for (size_t q = 0; q < y.nq && !rc; q++) {
        size_t x = xs[q];
        const uint64_t* o = proof + y.o_queries + q * y.query_words;
        const uint64_t* ev[3];
        size_t ncs[3] = {W, A, y.Q};
        for (int t = 0; t < 3; t++) {
            ev[t] = o;
            if (!merkle_verify(o, ncs[t], x, caps + t * y.C * 4, o + ncs[t], sib0)) { rc = 20 + t; break; }
            o += ncs[t] + sib0 * 4;
        }
        if (rc) break;
        gl_t sub_x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(y.lde_bits), bitrev(x, y.lde_bits)));
        /* fri_combine_initial */
        gl2_t sum = gl2_from_base(0);
        for (int b = 0; b < 3; b++) {
            size_t t = 0;
            if (b < 2) {
                for (size_t c = 0; c < W; c++) bvals[t++] = gl2_from_base(ev[0][c]);
                for (size_t c = 0; c < A; c++) bvals[t++] = gl2_from_base(ev[1][c]);
                if (b == 0) for (size_t c = 0; c < y.Q; c++) bvals[t++] = gl2_from_base(ev[2][c]);
            } else {
                for (size_t c = total_helpers; c < A; c++) bvals[t++] = gl2_from_base(ev[1][c]);
            }
            gl2_t red = reduce_ext(bvals, nb[b], fri_alpha);
            gl2_t num = gl2_sub(red, red_open[b]);
            gl2_t den = gl2_sub(gl2_from_base(sub_x), points[b]);
            sum = gl2_mul(sum, gl2_pow(fri_alpha, nb[b]));
            sum = gl2_add(sum, gl2_mul(num, gl2_inv(den)));
        }
        gl2_t old_eval = sum;
        for (unsigned l = 0; l < y.L; l++) {
            const gl2_t* evals = (const gl2_t*)o;
            size_t coset = x >> cfg->arity_bits, within = x & (arity - 1);
            if (!gl2_eq(evals[within], old_eval)) { rc = 30; break; }
            old_eval = compute_evaluation(sub_x, within, cfg->arity_bits, evals, betas[l]);
            size_t nsib = (size_t)(y.lde_bits - cfg->arity_bits * (l + 1) - y.cap);
            if (!merkle_verify(o, 2 * arity, coset, proof + y.o_fri_caps + l * y.C * 4, o + 2 * arity, nsib)) { rc = 31; break; }
            o += 2 * arity + nsib * 4;
            sub_x = gl_exp_pow2(sub_x, cfg->arity_bits);
            x = coset;
        }
        if (rc) break;
        gl2_t fe = eval_ext_poly_ext((const gl2_t*)(proof + y.o_final), y.F, gl2_from_base(sub_x));
        if (!gl2_eq(fe, old_eval)) rc = 32;
    }
    free(bvals);
    free(lv);
    return rc;
}
