/* oracle/zkm_oracle.h -- C API of the CPU oracle (libzkm_oracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  This is the CPU restatement of the reference's STARK/FRI hot path
 * (zkMIPS/zkm prover/src/prover.rs:441-789 and the plonky2 0.1.4 internals it calls).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it -- as the checker, never as the
 * product path.  PARITY STATUS: primitives pinned by known-answer vectors (Poseidon: plonky2 test
 * vectors; Keccak-f: reference keccak_util.rs:39-59); the plonky2-internal conventions (leaf order,
 * sponge mode, transcript order, FRI folding, PoW) are RECALLED, not verifiable here ("parity
 * unpinned" for bytes of caps/FRI); the pipeline is pinned by prove -> verify (restated verifier).
 *
 * Conventions: all field elements are canonical uint64_t; matrices are column-major (col*n + row);
 * F2 elements are two consecutive uint64_t [c0, c1].
 */
#ifndef ZKM_ORACLE_H
#define ZKM_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZKO_POSEIDON_COLS 262

/* ---- primitives ---- */
uint64_t zko_gl_mul(uint64_t a, uint64_t b);
uint64_t zko_gl_inv(uint64_t a);
uint64_t zko_gl_pow(uint64_t a, uint64_t e);
uint64_t zko_gl_root_of_unity(unsigned k);
void zko_gl2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void zko_gl2_inv(const uint64_t a[2], uint64_t out[2]);

void zko_poseidon_permute_naive(uint64_t st[12]);
void zko_poseidon_permute(uint64_t st[12]);
void zko_poseidon_permute_batch(uint64_t* states, size_t k);
void zko_poseidon_witness_row(const uint64_t in[12], uint64_t timestamp, int filter, uint64_t row[ZKO_POSEIDON_COLS]);
void zko_poseidon_hash_no_pad(const uint64_t* in, size_t len, uint64_t out[4]);
void zko_poseidon_hash_or_noop(const uint64_t* in, size_t len, uint64_t out[4]);
void zko_poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);

void zko_keccakf(uint64_t st[25]);
void zko_keccakf_batch(uint64_t* states, size_t k);
void zko_keccak256(const uint8_t* msg, size_t len, uint8_t out[32]);

/* KeccakSpongeStark::generate_trace (keccak_sponge_stark.rs:222-444); out = 470 x 2^log_n column-major; returns rows used, or 0 on error */
#define ZKO_KECCAK_SPONGE_COLS 470
size_t zko_keccak_sponge_trace(const uint8_t* inputs, const uint64_t* input_off, const uint64_t* meta, size_t nops, unsigned log_n,
                               uint64_t* out);

/* ---- NTT (natural order in and out) ---- */
void zko_ntt(uint64_t* cols, size_t ncols, unsigned log_n, int inverse, uint64_t coset_shift);

/* ---- polynomial batch commitment == plonky2 PolynomialBatch ---- */
typedef struct zko_batch zko_batch;
zko_batch* zko_batch_from_values(const uint64_t* values, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height);
zko_batch* zko_batch_from_coeffs(const uint64_t* coeffs, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height);
void zko_batch_free(zko_batch*);
void zko_batch_cap(const zko_batch*, uint64_t* out);                       /* 2^cap_height x 4 */
void zko_batch_coeffs(const zko_batch*, uint64_t* out);                    /* ncols x n */
void zko_batch_lde_row(const zko_batch*, size_t natural_index, uint64_t* out); /* get_lde_values(i, 1) */
void zko_batch_leaf(const zko_batch*, size_t leaf_index, uint64_t* out);    /* merkle_tree.leaves[i] */
void zko_batch_merkle_path(const zko_batch*, size_t leaf_index, uint64_t* siblings); /* (lde_bits-cap) x 4 */
void zko_batch_digest_layer(const zko_batch*, unsigned level, uint64_t* out); /* level 0 = leaf digests */

/* ---- Fiat-Shamir challenger (Poseidon duplex sponge) ---- */
typedef struct {
    uint64_t state[12];
    uint64_t in_buf[8];
    uint64_t out_buf[8];
    uint32_t n_in, n_out;
} zko_challenger;
void zko_challenger_init(zko_challenger*);
void zko_challenger_observe(zko_challenger*, const uint64_t* elems, size_t n);
uint64_t zko_challenger_get(zko_challenger*);
void zko_set_wide_threads(int n);   /* threads of the long row-parallel regions (0 = same as the rest) */
void zko_challenger_compact(zko_challenger*, uint64_t state_out[12]);

/* ---- synthetic PoseidonStark trace (poseidon_stark.rs:104-160) ---- */
/* inputs are derived from `seed` with SplitMix64; num_perms real rows (filter=1, timestamp 0), the rest
 * padded with the default row (permutation of zeros, filter=0).  Output column-major 262 x 2^log_n. */
void zko_poseidon_trace(uint64_t seed, size_t num_perms, unsigned log_n, uint64_t* out_cols);

/* ---- prove_single_table for PoseidonStark with the benchmark's fake CTL data ---- */
/* table ids */
#define ZKO_TABLE_POSEIDON 0
#define ZKO_TABLE_LOGIC 1
#define ZKO_TABLE_KECCAK_SPONGE 2
#define ZKO_TABLE_KECCAK 3
#define ZKO_TABLE_MEMORY 4
#define ZKO_TABLE_POSEIDON_SPONGE 5
#define ZKO_TABLE_SHA_EXTEND 6
#define ZKO_TABLE_SHA_EXTEND_SPONGE 7
#define ZKO_TABLE_SHA_COMPRESS 8
#define ZKO_TABLE_SHA_COMPRESS_SPONGE 9
#define ZKO_TABLE_ARITHMETIC 10
size_t zko_sha_compress_trace(const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t k, unsigned log_n, uint64_t* out);
void zko_sha_compress_sponge_trace(const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t k, unsigned log_n, uint64_t* out);
void zko_sha_extend_trace(const uint8_t* inputs, const uint64_t* timestamps, size_t k, unsigned log_n, uint64_t* out);
size_t zko_sha_extend_sponge_trace(const uint32_t* w16, const uint64_t* meta, size_t k, unsigned log_n, uint64_t* out);
#define ZKO_POSEIDON_SPONGE_COLS 110
void zko_poseidon_trace_inputs(const uint64_t* inputs, const uint64_t* timestamps, size_t num_perms, unsigned log_n, uint64_t* out);
size_t zko_poseidon_sponge_trace(const uint8_t* inputs, const uint64_t* input_off, const uint64_t* meta, size_t nops, unsigned log_n, uint64_t* out);
#define ZKO_MEMORY_COLS 13
/* MemoryStark::generate_trace (memory/memory_stark.rs:123-248); ops = nops x 6 {context, segment, virt, timestamp, is_read, value} */
size_t zko_memory_trace(const uint64_t* ops, size_t nops, unsigned log_n, uint64_t* out);
#define ZKO_KECCAK_COLS 2431
/* KeccakStark::generate_trace (keccak/keccak_stark.rs:62-236): inputs = nperms x 25 u64, one timestamp per permutation; out = 2431 x 2^log_n.
 * Returns the rows used (24 per permutation), 0 if they do not fit. */
size_t zko_keccak_trace(const uint64_t* inputs, const uint64_t* timestamps, size_t nperms, unsigned log_n, uint64_t* out);
/* LogicStark::generate_trace (logic.rs:150-183): ops = nops x (op (0 and, 1 or, 2 xor, 3 nor), in0, in1); out = 69 x 2^log_n */
void zko_logic_trace(const uint32_t* ops, size_t nops, unsigned log_n, uint64_t* out);
typedef struct {
    unsigned rate_bits, cap_height, pow_bits, num_challenges, num_queries, arity_bits, final_poly_bits;
} zko_stark_config;
void zko_standard_config(zko_stark_config*);

/* Proof blob layout: see include/zkm_hip.h (shared with the HIP product so they compare bytewise). */
size_t zko_proof_words(const zko_stark_config*, unsigned log_n, size_t ncols, size_t naux, size_t nctl_zs);
/* ctl description for the fake-CTL shape: nctl_zs CtlZData, each with num_helpers[i] helper columns and
 * no column sets; aux = ctl helper columns ++ ctl z columns (column-major naux x n). */
int zko_prove_single_table(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                           const uint64_t* aux, size_t naux, const uint32_t* num_helpers, size_t nctl_zs,
                           zko_challenger* challenger, uint64_t* proof_out, double* stage_seconds /* 8 or NULL */);
/* returns 0 if the proof verifies; otherwise a positive code naming the failed check */
int zko_verify_single_table(int table_id, const zko_stark_config* cfg, const uint64_t* proof, size_t ncols, size_t naux,
                            const uint32_t* num_helpers, size_t nctl_zs, zko_challenger* challenger);


/* ---- cross-table lookups: same array layouts as include/zkm_hip.h (declared again: the oracle shares no header
 * with the product).  Column/Filter/TableWithColumns/CrossTableLookup = cross_table_lookup.rs:31-415. */
typedef struct { uint32_t n_local, n_next, term_off, _pad; uint64_t constant; } zko_column;
typedef struct { uint32_t ncols, col_off, has_filter, nprod, prod_off, nconst, const_off, _pad; } zko_colset;
typedef struct {
    const zko_column* columns; size_t ncolumns;
    const uint32_t* term_col; const uint64_t* term_coeff; size_t nterms;
    const zko_colset* colsets; size_t ncolsets;
    const uint32_t* filter_idx; size_t nfilter_idx;
} zko_ctl_table;
typedef struct { uint32_t ncolsets, colset_off, num_helpers, _pad; uint64_t beta, gamma; } zko_ctl_z;
typedef struct { uint32_t table, colset; } zko_ctl_side;
typedef struct { uint32_t nlooking, looking_off; zko_ctl_side looked; } zko_cross_table_lookup;
/* (same layout as zkm_table_input of include/zkm_hip.h, so the tests pack one array for both; the oracle only reads `trace`) */
typedef struct { int table_id; const uint64_t* trace; size_t ncols; unsigned log_n; const zko_ctl_table* ctl; const uint64_t* const* columns; } zko_table_input;

void zko_ctl_data(const zko_ctl_table* t, const zko_ctl_z* zs, const uint32_t* colset_ids, size_t nzs, const uint64_t* trace,
                  size_t ncols, unsigned log_n, uint64_t* aux_out);
/* lookup_helper_columns (lookup.rs:46-124); out = (ceil(nlookup/2) + 1) x n: helper columns then Z */
void zko_lookup_helper_columns(const zko_ctl_table* t, const uint32_t* colset_ids, size_t nlookup, uint32_t table_col, uint32_t freq_col,
                               uint64_t challenge, const uint64_t* trace, size_t ncols, unsigned log_n, uint64_t* out);
/* check_ctls (cross_table_lookup.rs:1486-1581): 0 if every looking multiset equals its looked multiset */
int zko_check_ctls(const zko_table_input* tables, size_t ntables, const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls);
int zko_prove_single_table_ctl(int table_id, const zko_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                               const uint64_t* aux, size_t naux, const zko_ctl_table* t, const zko_ctl_z* zs,
                               const uint32_t* colset_ids, size_t nzs, const uint64_t* lookup_challenges, zko_challenger* challenger,
                               uint64_t* proof_out);
/* lookup helper columns a table's own logUp lookups add in front of the CTL columns (stark.rs:217-223); naux arguments of
 * the prove / verify entry points count the CTL columns only, zko_proof_words takes the total. */
size_t zko_num_lookup_columns(int table_id, const zko_stark_config* cfg);
int zko_verify_single_table_ctl(int table_id, const zko_stark_config* cfg, const uint64_t* proof, size_t ncols, size_t naux,
                                const zko_ctl_table* t, const zko_ctl_z* zs, const uint32_t* colset_ids, size_t nzs,
                                const uint64_t* lookup_challenges, zko_challenger* challenger);
size_t zko_all_proof_words(const zko_stark_config* cfg, const zko_table_input* tables, size_t ntables,
                           const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls, size_t* proof_offsets_out);
int zko_prove_with_traces(const zko_stark_config* cfg, const zko_table_input* tables, size_t ntables,
                          const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls,
                          const uint64_t* public_values, size_t npublic, uint64_t* proofs_out, uint64_t* ctl_challenges_out);
/* verify_proof (verifier.rs:27-176): 0 = accepted */
int zko_verify_all(const zko_stark_config* cfg, const zko_table_input* tables, size_t ntables,
                   const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls,
                   const uint64_t* public_values, size_t npublic, const uint64_t* proofs, const uint64_t* ctl_challenges);

/* prove_openings alone on three existing commitments (BASELINE config 4) and its verifier (FRI only) */
int zko_prove_openings(const zko_stark_config* cfg, zko_batch* tb, zko_batch* ab, zko_batch* qb, size_t nctl_zs, zko_challenger* ch,
                       uint64_t* proof);
int zko_verify_openings(const zko_stark_config* cfg, const uint64_t* proof, size_t ncols, size_t naux, size_t nctl_zs, zko_challenger* ch);

/* quotient stage alone (for stage-level parity): out = num_challenges*2 chunk polys... returns the
 * num_challenges quotient polys of 2n coefficients each (natural order). */
/* same for any table with constraints (ZKO_TABLE_*) */
void zko_quotient(int table_id, const zko_batch* trace, const zko_batch* aux, const uint32_t* num_helpers, size_t nctl_zs,
                  const uint64_t* alphas, size_t nalphas, uint64_t* out);
void zko_quotient_poseidon(const zko_batch* trace, const zko_batch* aux, const uint32_t* num_helpers, size_t nctl_zs,
                           const uint64_t* alphas, size_t nalphas, uint64_t* out_coeffs);
/* constraint evaluation of one row in the base field (check_constraints building block) */
long zko_debug_constraints(int table_id, const uint64_t* trace, size_t W, unsigned log_n, long* bad_row, long* bad_index);
long zko_debug_row_constraints(int table_id, const uint64_t* lv, const uint64_t* nv, int is_first, int is_last, uint64_t* out, size_t cap);
void zko_poseidon_eval_row(const uint64_t* local, const uint64_t* alphas, size_t nalphas, uint64_t* acc_out);
/* stark_testing.rs:21-70 (test_stark_low_degree): alpha-accumulated constraint values of a table on a low-degree extension of random
 * witness polynomials (rows: ncols x (2^log_witness << rate_bits), column-major, natural order, plain subgroup) */
void zko_constraint_evals(int table_id, const uint64_t* rows, size_t ncols, unsigned log_witness, unsigned rate_bits, uint64_t alpha,
                          uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
