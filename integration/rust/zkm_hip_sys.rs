//! zkm_hip_sys.rs -- `extern "C"` mirror of include/zkm_hip.h (libzkmhip.so), the drop-in boundary of the HIP proving path.
//!
//! Goes into the plonky2 fork as `plonky2/src/hip/sys.rs` (and is re-exported for zkm-prover).  Error convention of the
//! reference's only existing FFI (recursion/src/snark/snarks.rs:7-20, 39-59): `c_int` status, `*mut *mut c_char` message
//! that the caller frees.  `build.rs` links it the way recursion/build.rs:1-31 links the gnark library:
//!     println!("cargo:rustc-link-search=native={}", std::env::var("ZKM_HIP_LIB_DIR").unwrap());
//!     println!("cargo:rustc-link-lib=dylib=zkmhip");
//! NOT COMPILED in the build image (no cargo / rustc there); checked against the header by eye and by
//! tests/test_abi.py::test_rust_sys_block_names_every_export.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_uint, c_void};

#[repr(C)] pub struct zkm_ctx { _p: [u8; 0] }
#[repr(C)] pub struct zkm_batch { _p: [u8; 0] }
#[repr(C)] pub struct zkm_pool { _p: [u8; 0] }
#[repr(C)] pub struct zkm_staged { _p: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_challenger { pub state: [u64; 12], pub in_buf: [u64; 8], pub out_buf: [u64; 8], pub n_in: u32, pub n_out: u32 }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct zkm_stark_config { pub rate_bits: c_uint, pub cap_height: c_uint, pub pow_bits: c_uint, pub num_challenges: c_uint,
                              pub num_queries: c_uint, pub arity_bits: c_uint, pub final_poly_bits: c_uint }

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_proof_layout {
    pub degree_bits: u64, pub trace_cols: u64, pub aux_cols: u64, pub quotient_polys: u64, pub ctl_zs: u64, pub cap_height: u64,
    pub fri_layers: u64, pub final_poly_len: u64, pub num_queries: u64, pub rate_bits: u64, pub arity_bits: u64,
    pub total_words: usize, pub init_challenger_state: usize, pub trace_cap: usize, pub aux_cap: usize, pub quotient_cap: usize,
    pub local_values: usize, pub next_values: usize, pub aux_polys: usize, pub aux_polys_next: usize, pub ctl_zs_first: usize,
    pub quotient_polys_open: usize, pub commit_phase_merkle_caps: usize, pub final_poly: usize, pub pow_witness: usize,
    pub query_round_proofs: usize, pub query_round_words: usize,
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_proof_query_layout {
    pub oracle_evals: [usize; 3], pub oracle_cols: [usize; 3], pub oracle_siblings: [usize; 3], pub initial_siblings: usize,
    pub layer_evals: [usize; 16], pub layer_siblings: [usize; 16], pub layer_siblings_count: [usize; 16],
}

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_column { pub n_local: u32, pub n_next: u32, pub term_off: u32, pub _pad: u32, pub constant: u64 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_colset { pub ncols: u32, pub col_off: u32, pub has_filter: u32, pub nprod: u32, pub prod_off: u32, pub nconst: u32,
                        pub const_off: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct zkm_ctl_table { pub columns: *const zkm_column, pub ncolumns: usize, pub term_col: *const u32, pub term_coeff: *const u64,
                           pub nterms: usize, pub colsets: *const zkm_colset, pub ncolsets: usize, pub filter_idx: *const u32,
                           pub nfilter_idx: usize }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_ctl_z { pub ncolsets: u32, pub colset_off: u32, pub num_helpers: u32, pub _pad: u32, pub beta: u64, pub gamma: u64 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_ctl_side { pub table: u32, pub colset: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct zkm_cross_table_lookup { pub nlooking: u32, pub looking_off: u32, pub looked: zkm_ctl_side }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct zkm_table_input { pub table_id: c_int, pub trace: *const u64, pub ncols: usize, pub log_n: c_uint, pub ctl: *const zkm_ctl_table,
                             pub columns: *const *const u64 }
/// descriptor of one FRI oracle / batch for zkm_prove_openings_fri (FriInstanceInfo, stark.rs:91-148)
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct zkm_fri_poly { pub oracle: u32, pub poly: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct zkm_fri_batch { pub point: [u64; 2], pub polys: *const zkm_fri_poly, pub npolys: usize }

// ZKM_TABLE_* ids (NOT the reference's Table enum order: see zkm_table_enum_index)
pub const ZKM_TABLE_POSEIDON: c_int = 0; pub const ZKM_TABLE_LOGIC: c_int = 1; pub const ZKM_TABLE_KECCAK_SPONGE: c_int = 2;
pub const ZKM_TABLE_KECCAK: c_int = 3; pub const ZKM_TABLE_MEMORY: c_int = 4; pub const ZKM_TABLE_POSEIDON_SPONGE: c_int = 5;
pub const ZKM_TABLE_SHA_EXTEND: c_int = 6; pub const ZKM_TABLE_SHA_EXTEND_SPONGE: c_int = 7; pub const ZKM_TABLE_SHA_COMPRESS: c_int = 8;
pub const ZKM_TABLE_SHA_COMPRESS_SPONGE: c_int = 9; pub const ZKM_TABLE_ARITHMETIC: c_int = 10; pub const ZKM_TABLE_CPU: c_int = 11;

#[link(name = "zkmhip")]
extern "C" {
    // context / memory
    pub fn zkm_ctx_create(device: c_int, out: *mut *mut zkm_ctx, err: *mut *mut c_char) -> c_int;
    pub fn zkm_ctx_destroy(ctx: *mut zkm_ctx);
    pub fn zkm_ctx_synchronize(ctx: *mut zkm_ctx, err: *mut *mut c_char) -> c_int;
    pub fn zkm_ctx_stream(ctx: *mut zkm_ctx) -> *mut c_void;
    pub fn zkm_ctx_memory(ctx: *const zkm_ctx, live_bytes: *mut usize, cached_bytes: *mut usize);
    pub fn zkm_ctx_resident_bytes(ctx: *const zkm_ctx) -> usize;
    pub fn zkm_ctx_trim(ctx: *mut zkm_ctx);
    pub fn zkm_ctx_set_tuning(ctx: *mut zkm_ctx, key: *const c_char, value: u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_dev_alloc(ctx: *mut zkm_ctx, bytes: usize, out: *mut *mut c_void, err: *mut *mut c_char) -> c_int;
    pub fn zkm_dev_free(ctx: *mut zkm_ctx, p: *mut c_void) -> c_int;
    pub fn zkm_dev_upload(ctx: *mut zkm_ctx, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize, err: *mut *mut c_char) -> c_int;
    pub fn zkm_dev_download(ctx: *mut zkm_ctx, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize, err: *mut *mut c_char) -> c_int;
    pub fn zkm_host_alloc(ctx: *mut zkm_ctx, bytes: usize, out: *mut *mut c_void, err: *mut *mut c_char) -> c_int;
    pub fn zkm_host_free(ctx: *mut zkm_ctx, p: *mut c_void) -> c_int;
    pub fn zkm_host_register(ctx: *mut zkm_ctx, p: *mut c_void, bytes: usize, err: *mut *mut c_char) -> c_int;
    pub fn zkm_host_unregister(ctx: *mut zkm_ctx, p: *mut c_void) -> c_int;
    // NTT / PolynomialBatch
    pub fn zkm_ntt(ctx: *mut zkm_ctx, cols: *mut u64, ncols: usize, log_n: c_uint, inverse: c_int, coset_shift: u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_batch_commit_values(ctx: *mut zkm_ctx, values: *const u64, ncols: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint,
                                   out: *mut *mut zkm_batch, err: *mut *mut c_char) -> c_int;
    pub fn zkm_batch_commit_coeffs(ctx: *mut zkm_ctx, coeffs: *const u64, ncols: usize, log_n: c_uint, rate_bits: c_uint, cap_height: c_uint,
                                   out: *mut *mut zkm_batch, err: *mut *mut c_char) -> c_int;
    pub fn zkm_batch_commit_columns(ctx: *mut zkm_ctx, columns: *const *const u64, ncols: usize, log_n: c_uint, columns_are_values: c_int,
                                    rate_bits: c_uint, cap_height: c_uint, out: *mut *mut zkm_batch, err: *mut *mut c_char) -> c_int;
    pub fn zkm_batch_free(b: *mut zkm_batch);
    pub fn zkm_batch_cap(b: *const zkm_batch, out: *mut u64) -> c_int;
    pub fn zkm_batch_coeffs(b: *const zkm_batch, out: *mut u64) -> c_int;
    pub fn zkm_batch_lde_row(b: *const zkm_batch, natural_index: usize, out: *mut u64) -> c_int;
    pub fn zkm_batch_lde_rows(b: *const zkm_batch, index_start: usize, step: usize, count: usize, out: *mut u64) -> c_int;
    pub fn zkm_batch_leaf(b: *const zkm_batch, leaf_index: usize, out: *mut u64) -> c_int;
    pub fn zkm_batch_merkle_path(b: *const zkm_batch, leaf_index: usize, siblings_out: *mut u64) -> c_int;
    pub fn zkm_batch_digest_layer(b: *const zkm_batch, level: c_uint, out: *mut u64) -> c_int;
    pub fn zkm_field_selftest(ctx: *mut zkm_ctx, a: *const u64, b: *const u64, n: usize, out: *mut u64, err: *mut *mut c_char) -> c_int;
    // hash primitives, witness kernels
    pub fn zkm_poseidon_permute_batch(ctx: *mut zkm_ctx, states: *mut u64, k: usize, err: *mut *mut c_char) -> c_int;
    pub fn zkm_keccakf_batch(ctx: *mut zkm_ctx, states: *mut u64, k: usize, err: *mut *mut c_char) -> c_int;
    pub fn zkm_poseidon_trace(ctx: *mut zkm_ctx, seed: u64, num_perms: usize, log_n: c_uint, out_dev: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_poseidon_trace_inputs(ctx: *mut zkm_ctx, inputs: *const u64, timestamps: *const u64, num_perms: usize, log_n: c_uint,
                                     out_dev: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_poseidon_sponge_trace(ctx: *mut zkm_ctx, inputs: *const u8, input_off: *const u64, meta: *const u64, nops: usize, log_n: c_uint,
                                     out_dev: *mut u64, rows_used_out: *mut usize, err: *mut *mut c_char) -> c_int;
    pub fn zkm_keccak_sponge_trace(ctx: *mut zkm_ctx, inputs: *const u8, input_off: *const u64, meta: *const u64, nops: usize, log_n: c_uint,
                                   out_dev: *mut u64, rows_used_out: *mut usize, err: *mut *mut c_char) -> c_int;
    pub fn zkm_keccak_trace(ctx: *mut zkm_ctx, inputs: *const u64, timestamps: *const u64, nperms: usize, log_n: c_uint, out_dev: *mut u64,
                            err: *mut *mut c_char) -> c_int;
    pub fn zkm_sha_extend_trace(ctx: *mut zkm_ctx, inputs: *const u8, timestamps: *const u64, nrows: usize, log_n: c_uint, out_dev: *mut u64,
                                err: *mut *mut c_char) -> c_int;
    pub fn zkm_sha_extend_sponge_trace(ctx: *mut zkm_ctx, w16: *const u32, meta: *const u64, nblocks: usize, log_n: c_uint, out_dev: *mut u64,
                                       err: *mut *mut c_char) -> c_int;
    pub fn zkm_sha_compress_trace(ctx: *mut zkm_ctx, hx: *const u32, w: *const u32, meta: *const u64, ncomp: usize, log_n: c_uint,
                                  out_dev: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_sha_compress_sponge_trace(ctx: *mut zkm_ctx, hx: *const u32, w: *const u32, meta: *const u64, ncomp: usize, log_n: c_uint,
                                         out_dev: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_logic_trace(ctx: *mut zkm_ctx, ops: *const u32, nops: usize, log_n: c_uint, out_dev: *mut u64, err: *mut *mut c_char) -> c_int;
    // Fiat-Shamir
    pub fn zkm_challenger_init(ch: *mut zkm_challenger);
    pub fn zkm_challenger_observe(ch: *mut zkm_challenger, elems: *const u64, n: usize);
    pub fn zkm_challenger_get(ch: *mut zkm_challenger) -> u64;
    pub fn zkm_challenger_compact(ch: *mut zkm_challenger, state_out: *mut u64);
    // STARK
    pub fn zkm_standard_config(cfg: *mut zkm_stark_config);
    pub fn zkm_table_width(table_id: c_int) -> usize;
    pub fn zkm_table_enum_index(table_id: c_int) -> c_int;
    pub fn zkm_num_lookup_columns(table_id: c_int, cfg: *const zkm_stark_config) -> usize;
    pub fn zkm_proof_words(cfg: *const zkm_stark_config, log_n: c_uint, ncols: usize, naux: usize, nctl_zs: usize) -> usize;
    pub fn zkm_proof_get_layout(proof: *const u64, out: *mut zkm_proof_layout) -> c_int;
    pub fn zkm_proof_get_query_layout(proof: *const u64, out: *mut zkm_proof_query_layout) -> c_int;
    pub fn zkm_prove_single_table(ctx: *mut zkm_ctx, table_id: c_int, cfg: *const zkm_stark_config, trace: *const u64, ncols: usize,
                                  log_n: c_uint, trace_batch: *const zkm_batch, aux: *const u64, naux: usize, num_helpers: *const u32,
                                  nctl_zs: usize, challenger: *mut zkm_challenger, proof_out: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_prove_single_table_ctl(ctx: *mut zkm_ctx, table_id: c_int, cfg: *const zkm_stark_config, trace: *const u64, ncols: usize,
                                      log_n: c_uint, trace_batch: *const zkm_batch, aux: *const u64, naux: usize, table: *const zkm_ctl_table,
                                      zs: *const zkm_ctl_z, colset_ids: *const u32, nzs: usize, lookup_challenges: *const u64,
                                      challenger: *mut zkm_challenger, proof_out: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_ctl_data(ctx: *mut zkm_ctx, table: *const zkm_ctl_table, zs: *const zkm_ctl_z, colset_ids: *const u32, nzs: usize,
                        trace: *const u64, ncols: usize, log_n: c_uint, aux_out: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_lookup_helper_columns(ctx: *mut zkm_ctx, table: *const zkm_ctl_table, colset_ids: *const u32, nlookup: usize, table_col: u32,
                                     freq_col: u32, challenge: u64, trace: *const u64, ncols: usize, log_n: c_uint, out: *mut u64,
                                     err: *mut *mut c_char) -> c_int;
    pub fn zkm_all_proof_words(cfg: *const zkm_stark_config, tables: *const zkm_table_input, ntables: usize, ctls: *const zkm_cross_table_lookup,
                               sides: *const zkm_ctl_side, nctls: usize, proof_offsets_out: *mut usize) -> usize;
    pub fn zkm_prove_with_traces(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, tables: *const zkm_table_input, ntables: usize,
                                 ctls: *const zkm_cross_table_lookup, sides: *const zkm_ctl_side, nctls: usize, public_values: *const u64,
                                 npublic: usize, proofs_out: *mut u64, ctl_challenges_out: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_all_stark_ctls(ctls_out: *mut *const zkm_cross_table_lookup, nctls_out: *mut usize, sides_out: *mut *const zkm_ctl_side,
                              nsides_out: *mut usize) -> c_int;
    pub fn zkm_all_stark_ctl_table(table_id: c_int) -> *const zkm_ctl_table;
    pub fn zkm_prove_segment(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, traces: *const *const u64, log_n: *const c_uint,
                             public_values: *const u64, npublic: usize, proofs_out: *mut u64, proof_offsets_out: *mut usize,
                             ctl_challenges_out: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_prove_segment_columns(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, columns: *const *const *const u64, log_n: *const c_uint,
                                     public_values: *const u64, npublic: usize, proofs_out: *mut u64, proof_offsets_out: *mut usize,
                                     ctl_challenges_out: *mut u64, err: *mut *mut c_char) -> c_int;
    // K independent segments in lock-step (include/zkm_hip.h): traces[s][t] / columns[s][t][i], log_n[s][t], one output buffer per segment
    pub fn zkm_prove_segments(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, nseg: usize, traces: *const *const *const u64,
                              log_n: *const *const c_uint, public_values: *const *const u64, npublic: *const usize, proofs_out: *const *mut u64,
                              ctl_challenges_out: *const *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_prove_segments_columns(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, nseg: usize, columns: *const *const *const *const u64,
                                      log_n: *const *const c_uint, public_values: *const *const u64, npublic: *const usize,
                                      proofs_out: *const *mut u64, ctl_challenges_out: *const *mut u64, err: *mut *mut c_char) -> c_int;
    // staged traces: the next proof's upload behind the current proof (copy streams; returns at once)
    pub fn zkm_trace_stage(ctx: *mut zkm_ctx, values: *const u64, ncols: usize, log_n: c_uint, canonical: c_int, out: *mut *mut zkm_staged,
                           err: *mut *mut c_char) -> c_int;
    pub fn zkm_trace_stage_columns(ctx: *mut zkm_ctx, columns: *const *const u64, ncols: usize, log_n: c_uint, canonical: c_int,
                                   out: *mut *mut zkm_staged, err: *mut *mut c_char) -> c_int;
    pub fn zkm_segment_stage(ctx: *mut zkm_ctx, traces: *const *const u64, log_n: *const c_uint, canonical: c_int, out: *mut *mut zkm_staged,
                             err: *mut *mut c_char) -> c_int;
    pub fn zkm_segment_stage_columns(ctx: *mut zkm_ctx, columns: *const *const *const u64, log_n: *const c_uint, canonical: c_int,
                                     out: *mut *mut zkm_staged, err: *mut *mut c_char) -> c_int;
    pub fn zkm_staged_segment_ptrs(staged: *mut zkm_staged, ptrs_out: *mut *const u64) -> c_int;
    pub fn zkm_staged_ptr(staged: *mut zkm_staged) -> *const u64;
    pub fn zkm_staged_ready(staged: *mut zkm_staged, wait: c_int) -> c_int;
    pub fn zkm_staged_free(staged: *mut zkm_staged);
    // one process, many GPUs: contexts_per_device contexts on each device, one worker thread per context, groups of <= max_stack segments
    pub fn zkm_pool_create(devices: *const c_int, ndevices: usize, contexts_per_device: usize, out: *mut *mut zkm_pool, err: *mut *mut c_char) -> c_int;
    pub fn zkm_pool_destroy(pool: *mut zkm_pool);
    pub fn zkm_pool_workers(pool: *const zkm_pool) -> usize;
    pub fn zkm_pool_context(pool: *mut zkm_pool, worker: usize) -> *mut zkm_ctx;
    pub fn zkm_pool_device(pool: *const zkm_pool, worker: usize) -> c_int;
    pub fn zkm_pool_set_tuning(pool: *mut zkm_pool, key: *const c_char, value: u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_pool_prove_segments(pool: *mut zkm_pool, cfg: *const zkm_stark_config, nseg: usize, max_stack: usize,
                                   traces: *const *const *const u64, log_n: *const *const c_uint, public_values: *const *const u64,
                                   npublic: *const usize, proofs_out: *const *mut u64, ctl_challenges_out: *const *mut u64,
                                   err: *mut *mut c_char) -> c_int;
    pub fn zkm_pool_prove_segments_columns(pool: *mut zkm_pool, cfg: *const zkm_stark_config, nseg: usize, max_stack: usize,
                                           columns: *const *const *const *const u64, log_n: *const *const c_uint,
                                           public_values: *const *const u64, npublic: *const usize, proofs_out: *const *mut u64,
                                           ctl_challenges_out: *const *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_pool_plan(nseg: usize, workers: usize, max_stack: usize, group_sizes_out: *mut usize, capacity: usize) -> usize;
    pub fn zkm_pool_last_assignment(pool: *const zkm_pool, segment: usize, worker_out: *mut usize, group_out: *mut usize) -> c_int;
    pub fn zkm_prove_single_tables(ctx: *mut zkm_ctx, table_id: c_int, cfg: *const zkm_stark_config, nproofs: usize, traces: *const *const u64,
                                   ncols: usize, log_n: c_uint, aux: *const *const u64, naux: usize, num_helpers: *const u32, nctl_zs: usize,
                                   challengers: *const *mut zkm_challenger, proofs_out: *const *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_fri_proof_words(cfg: *const zkm_stark_config, log_n: c_uint, oracle_cols: *const usize, noracles: usize) -> usize;
    pub fn zkm_fri_prove(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, oracles: *const *const zkm_batch, noracles: usize,
                         batches: *const zkm_fri_batch, nbatches: usize, challenger: *mut zkm_challenger, proof_out: *mut u64,
                         err: *mut *mut c_char) -> c_int;
    pub fn zkm_prove_openings(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, trace_batch: *const zkm_batch, aux_batch: *const zkm_batch,
                              quot_batch: *const zkm_batch, nctl_zs: usize, challenger: *mut zkm_challenger, proof_out: *mut u64,
                              err: *mut *mut c_char) -> c_int;
    pub fn zkm_segment_image_words(tables: *const zkm_table_input, ntables: usize, ctls: *const zkm_cross_table_lookup, sides: *const zkm_ctl_side,
                                   nctls: usize, npublic: usize) -> usize;
    pub fn zkm_segment_image_write(tables: *const zkm_table_input, ntables: usize, ctls: *const zkm_cross_table_lookup, sides: *const zkm_ctl_side,
                                   nctls: usize, public_values: *const u64, npublic: usize, image_out: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_prove_segment_image(ctx: *mut zkm_ctx, cfg: *const zkm_stark_config, image: *const u64, image_words: usize, proofs_out: *mut u64,
                                   proof_words_out: *mut usize, proof_offsets_out: *mut usize, ctl_challenges_out: *mut u64,
                                   err: *mut *mut c_char) -> c_int;
    pub fn zkm_quotient(ctx: *mut zkm_ctx, table_id: c_int, trace: *const zkm_batch, aux: *const zkm_batch, num_helpers: *const u32, nctl_zs: usize,
                        alphas: *const u64, nalphas: usize, out_coeffs: *mut u64, err: *mut *mut c_char) -> c_int;
    pub fn zkm_eval_openings(ctx: *mut zkm_ctx, b: *const zkm_batch, zeta: *const u64, out: *mut u64, err: *mut *mut c_char) -> c_int;
    // check_constraints (prover.rs:793-910): debugging aid, rc != 0 + "Constraint failed in <Stark>" and the first failing row
    pub fn zkm_check_constraints(ctx: *mut zkm_ctx, table_id: c_int, cfg: *const zkm_stark_config, trace: *const u64, ncols: usize, log_n: c_uint,
                                 aux: *const u64, naux: usize, table: *const zkm_ctl_table, zs: *const zkm_ctl_z, colset_ids: *const u32,
                                 nzs: usize, lookup_challenges: *const u64, alphas: *const u64, nalphas: usize, first_failing_row: *mut u64,
                                 err: *mut *mut c_char) -> c_int;
    // profiling
    pub fn zkm_profile_enable(ctx: *mut zkm_ctx, on: c_int);
    pub fn zkm_profile_reset(ctx: *mut zkm_ctx);
    pub fn zkm_profile_count(ctx: *mut zkm_ctx) -> usize;
    pub fn zkm_profile_get(ctx: *mut zkm_ctx, i: usize, name: *mut *const c_char, launches: *mut u64, total_ms: *mut c_double) -> c_int;
    pub fn zkm_version() -> *const c_char;
}

/// status + message -> anyhow::Error; the message is released with free() like recursion/src/snark/snarks.rs:55-57
pub fn check(rc: c_int, err: *mut c_char) -> anyhow::Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = if err.is_null() { format!("libzkmhip error {rc}") } else { unsafe { std::ffi::CStr::from_ptr(err).to_string_lossy().into_owned() } };
    if !err.is_null() {
        unsafe { libc::free(err as *mut c_void) };
    }
    Err(anyhow::anyhow!(msg))
}
