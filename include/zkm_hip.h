/* include/zkm_hip.h -- C ABI of libzkmhip.so: the MI355X (gfx950) STARK/FRI hot path of the zkMIPS prover.
 *
 * This is the drop-in boundary.  The reference (zkMIPS/zkm @ 2025-04-04) has no FFI on this path: the
 * prover calls the un-vendored plonky2 fork (prover/Cargo.toml:17-20) through Rust generics.  Each entry
 * point below names the reference call site / plonky2 item it replaces; INTEGRATION.md shows the
 * `extern "C"` block a patched plonky2 / zkm-prover would bind.  Error convention follows the reference's
 * only existing FFI (recursion/src/snark/snarks.rs:7-20, 39-59): int status (0 = ok) + optional
 * malloc'd message in *err that the caller releases with free().
 *
 * Data conventions
 *   - field element: canonical uint64_t (< p = 2^64 - 2^32 + 1); F2 = F[X]/(X^2-7) as [c0, c1].
 *   - matrices are column-major: element (row r, column c) of an n-row matrix is at c*n + r -- the layout
 *     of Vec<PolynomialValues<F>> flattened (prover/src/util.rs:37-46).
 *   - every data pointer may be a host pointer or a device (HBM) pointer of the context's GPU; the
 *     library detects which (hipPointerGetAttributes).  Output buffers documented "host" must be host.
 *   - a zkm_ctx is single-owner (one per GPU, driven from one thread at a time), like the reference's
 *     `&mut Challenger` / `&mut TimingTree` threading (prover.rs:441-450).
 */
#ifndef ZKM_HIP_H
#define ZKM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkm_ctx zkm_ctx;
typedef struct zkm_batch zkm_batch;

/* ------------------------------------------------------------------ context / memory */
int zkm_ctx_create(int device, zkm_ctx** out, char** err);
void zkm_ctx_destroy(zkm_ctx* ctx);
int zkm_ctx_synchronize(zkm_ctx* ctx, char** err);
/* the hipStream_t every kernel of this context is launched on (for event timing by the host) */
void* zkm_ctx_stream(zkm_ctx* ctx);
/* Device memory held by the context's caching allocator: bytes in live allocations (DeviceBuffers, batches, scratch of a call in
 * flight) and bytes cached for reuse (freed on zkm_ctx_destroy, or when an allocation would otherwise fail). */
void zkm_ctx_memory(const zkm_ctx* ctx, size_t* live_bytes, size_t* cached_bytes);
/* Of the live bytes: tables the context keeps for reuse (twiddles, coset power tables, block twiddles -- also those of its commit
 * lanes, which build their own on first use).  live - resident == 0 between calls when the caller holds no buffers or batches. */
size_t zkm_ctx_resident_bytes(const zkm_ctx* ctx);
/* Return every cached (not live) block to the device (the free lists are exact-size: a segment of many table shapes leaves one
 * cached block per distinct size behind). */
void zkm_ctx_trim(zkm_ctx* ctx);
/* Size thresholds at which the library switches between two kernels that produce the same words (every switch has a parity test on
 * both sides, tests/test_gpu_prove.py::test_tuning_*).  Keys:
 *   "ingest_chunk_cols"         columns per chunk of the pipelined host ingest (default 32; 0 = one monolithic upload)
 *   "keccak_parts_max_points"   quotient domains of a Keccak table up to this many points use 25 threads per point (default 2^15)
 *   "fri_fused_division_min"    polynomials from this many coefficients on divide by (X - z) with all batches in one workgroup (default: never)
 *   "fri_scan_combine"          the bottom level of the division by (X - z): suffix scan of every opening batch and their weighted sum in ONE
 *                               launch, nothing but the final polynomial written (default 1); 0: a scan launch writing the suffix values of
 *                               every batch and a combine launch reading them again (round 4's form)
 *   "wide_max_hashes"           hashing launches of up to this many hashes give each hash a 16-lane row (default 1024), and
 *   "quad_max_hashes"           up to this many a quad of lanes (default 32768): the latency forms of the Poseidon permutation, at
 *                               4.2x / 1.45x the issue slots of the one-lane form.  0 and 0: one lane per hash for every leaf and
 *                               every tree level of >= 256 parents (the levels below that, a few hundred hashes per tree, keep the
 *                               quad form: the one-lane kernel works on blocks of 256 parents); a GPU shared by many contexts
 *                               proving small segments may prefer throughput -- measured in profiles/r03_hw_queues.txt
 *   "leaf_mfma"                 one-lane-per-leaf hashing (trace / auxiliary leaves, FRI layer leaves) with the dense MDS layers of the FULL
 *                               rounds on the matrix core (v_mfma_i32_32x32x32_i8 with a block-diagonal matrix operand: every lane gets M
 *                               times its own state; csrc/poseidon_mfma_dev.h: 32-bit partial sums, 93 registers, five waves per SIMD) and the
 *                               s-boxes / partial rounds on the vector ALU (default 1: leaf hashing 40.9 -> 39.4 ms at 262 x 2^22, +2 %
 *                               proofs/s and lock-step segments/s); 0: every layer as 32-bit multiply-adds
 *   "small_ntt"                 transforms of 2^9 .. 2^13 points (inverse transforms and coset-LDE blocks of short tables, quotient
 *                               chunks, FRI layers) in ONE launch, a column per workgroup (default 1); 0: two strided passes
 *   "tree_tail"                 Merkle trees of up to 2^15 leaves built in ONE launch that also delivers the cap to the host (default 1);
 *                               0: up to three launches for the levels and a download kernel
 *   "commit_lanes"              trace / auxiliary commitments of one segment built side by side (default 4: the context and three
 *                               lanes, one stream and host thread each; 1 = everything on the context's own stream)
 *   "pow_round_log"             the proof-of-work search tries 2^this candidates per round of its one launch (default 17; 8 .. 22);
 *                               the witness found is the smallest one whatever the value
 *   "max_stack"                 zkm_prove_segments: segments per lock-step group (default and maximum 32; 1 = one segment at a time, the
 *                               path of zkm_prove_segment); a table's group is also bounded by 65535 stacked columns (Keccak: 26 segments)
 *   "segments_memory_budget"    zkm_prove_segments: bytes of HBM the segments proven together may hold (estimated: values, coefficients, 4x LDE
 *                               and digests of every table's three commitments); more segments than that are proven in consecutive waves.
 *                               Default 0 = 80 % of (the blocks this context has cached + free memory) at the call; contexts sharing a GPU
 *                               are sized by the caller (contexts x segments per call x ~1.8 GB for 2^16-cycle segments)
 *   "aux_pipeline"              a segment whose tables are all short (no LDE over 1 GiB), commit_lanes > 1: the lanes build the auxiliary
 *                               commitments of tables 1.. BEHIND the proofs of the earlier tables instead of all of them before the
 *                               first proof (default 1; same transcript, same proofs); 0: all auxiliary commitments first
 *   "throughput_profile"        1: the settings for MANY contexts per GPU proving small segments (commit_lanes 1, wide_max_hashes 256,
 *                               quad_max_hashes 4096, pow_round_log 16: 16 contexts reach 74-75 segments/s of 2^16 cycles against 57-59 for 8 contexts
 *                               with the defaults, profiles/r04_throughput_profile.txt); 0: the defaults again
 *   "block_after_us"            a transcript round trip (cap, opening partials, proof-of-work witness coming down) is waited for by
 *                               polling a flag in pinned memory; after this many microseconds (default 50) a thread of a CROWDED
 *                               process -- more than half as many threads waiting as CPUs the process may run on -- parks on a
 *                               blocking-sync event instead, leaving its core to the other contexts.  0: always park.
 *   "debug_fail_allocs"         TEST HOOK, accepted only when the process environment holds ZKM_ENABLE_TEST_HOOKS=1 (an unknown key otherwise):
 *                               the next `value` device allocations of the context and its lanes fail on their first
 *                               attempt as if out of memory, so the recovery path (trim the caches -- this context's, then the
 *                               parent's and the sibling lanes' -- and retry) runs; proofs are unchanged (tests/test_segment.py)
 * An unknown key is an error.  Applies to the context and its commit lanes. */
int zkm_ctx_set_tuning(zkm_ctx* ctx, const char* key, uint64_t value, char** err);
/* Pinned host memory (N3, trace ingest): host-resident traces (the reference's Vec<PolynomialValues>, prover.rs:144-167) are
 * uploaded in column chunks on a copy stream while earlier chunks are transformed and hashed; that overlap needs page-locked
 * source memory.  Either let the witness generator write into zkm_host_alloc memory, or zkm_host_register its own buffers once.
 * Pageable host pointers are accepted everywhere too (the upload then blocks the calling thread chunk by chunk). */
int zkm_host_alloc(zkm_ctx* ctx, size_t bytes, void** out, char** err);
int zkm_host_free(zkm_ctx* ctx, void* p);
int zkm_host_register(zkm_ctx* ctx, void* p, size_t bytes, char** err);
int zkm_host_unregister(zkm_ctx* ctx, void* p);
int zkm_dev_alloc(zkm_ctx* ctx, size_t bytes, void** out, char** err);
int zkm_dev_free(zkm_ctx* ctx, void* p);
int zkm_dev_upload(zkm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, char** err);
int zkm_dev_download(zkm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, char** err);

/* ------------------------------------------------------------------ K1/K2/K9: batched Goldilocks NTT
 * Replaces PolynomialValues::{ifft, coset_ifft} / PolynomialCoeffs::{fft, coset_fft}
 * (call sites prover.rs:678-681, 787, 829).  In place over `ncols` columns of 2^log_n elements,
 * natural order in and out.  coset_shift 0 or 1 = plain subgroup; otherwise forward = scale coeff i by
 * shift^i then NTT, inverse = iNTT then scale coeff i by shift^-i (plonky2 conventions). */
int zkm_ntt(zkm_ctx* ctx, uint64_t* cols, size_t ncols, unsigned log_n, int inverse, uint64_t coset_shift, char** err);

/* Parity / debug: the loose-arithmetic field primitives of the butterflies on arbitrary 64-bit words (also >= p), n pairs from host
 * memory: out = 7 x n canonical words: a + b, a - b, a 2^24, a 2^48, a 2^72, a b, and a b again through the branch-free product the
 * quotient / opening kernels use (mod p).  Exercises the corrections of the device arithmetic that field data reaches with probability
 * 2^-32 .. 2^-64 (second wrap of add / subtract, borrow of the Goldilocks reduction with and without the product's middle carry). */
int zkm_field_selftest(zkm_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, char** err);

/* ------------------------------------------------------------------ K1-K5: PolynomialBatch
 * zkm_batch_commit_values == PolynomialBatch::from_values(values, rate_bits, false, cap_height, ..)
 *   (prover.rs:154-163, 514-521); zkm_batch_commit_coeffs == from_coeffs (prover.rs:579-586).
 * Inputs are borrowed for the call (no clone, cf. prover.rs:155-157); the batch owns device memory:
 * coefficients (ncols x n), the LDE (ncols x 4n, rows in bit-reversed order == merkle_tree.leaves) and
 * all Merkle digest layers. */
int zkm_batch_commit_values(zkm_ctx* ctx, const uint64_t* values, size_t ncols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, zkm_batch** out, char** err);
int zkm_batch_commit_coeffs(zkm_ctx* ctx, const uint64_t* coeffs, size_t ncols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, zkm_batch** out, char** err);
/* The same two constructors from ONE POINTER PER COLUMN: the reference's argument is a Vec<PolynomialValues<F>> /
 * Vec<PolynomialCoeffs<F>> -- a separate heap allocation per column (prover.rs:154-163 trace_poly_values, :576-587 chunks) -- so the
 * Rust side passes `values.iter().map(|p| p.values.as_ptr())` and nothing is flattened on the host.  columns[i] -> 2^log_n words,
 * host (pageable, pinned or registered) or device; columns_are_values != 0: from_values, 0: from_coeffs.  Wide host-resident
 * matrices take the same pipelined ingest as zkm_batch_commit_values (column chunks on a copy stream). */
int zkm_batch_commit_columns(zkm_ctx* ctx, const uint64_t* const* columns, size_t ncols, unsigned log_n, int columns_are_values,
                             unsigned rate_bits, unsigned cap_height, zkm_batch** out, char** err);
void zkm_batch_free(zkm_batch* b);
/* .merkle_tree.cap (prover.rs:180, 524, 588): 2^cap_height digests x 4 words, host out */
int zkm_batch_cap(const zkm_batch* b, uint64_t* out);
/* .polynomials (proof.rs:311): ncols x n coefficients, natural order; out may be host or device */
int zkm_batch_coeffs(const zkm_batch* b, uint64_t* out);
/* get_lde_values(natural_index, 1) (prover.rs:687): one LDE row, ncols words, host out */
int zkm_batch_lde_row(const zkm_batch* b, size_t natural_index, uint64_t* out);
/* get_lde_values_packed(index_start, step) generalised to `count` consecutive indices (prover.rs:687, 723-748: the quotient
 * loop reads rows i_start .. i_start + P::WIDTH): out[i * ncols + c] = get_lde_values(index_start + i, step)[c], host or device */
int zkm_batch_lde_rows(const zkm_batch* b, size_t index_start, size_t step, size_t count, uint64_t* out);
/* merkle_tree.leaves[leaf_index] and merkle_tree.prove(leaf_index): (lde_bits - cap_height) x 4 words */
int zkm_batch_leaf(const zkm_batch* b, size_t leaf_index, uint64_t* out);
int zkm_batch_merkle_path(const zkm_batch* b, size_t leaf_index, uint64_t* siblings_out);
/* digest layer `level` (0 = leaf digests), (4n >> level) x 4 words, host out -- parity/debug */
int zkm_batch_digest_layer(const zkm_batch* b, unsigned level, uint64_t* out);

/* ------------------------------------------------------------------ hash primitives
 * a13: Poseidon permutation (prover/src/poseidon/poseidon_stark.rs:51-95; == plonky2 PoseidonHash's
 * permutation); k states of 12 words, in place.  K15: Keccak-f[1600] (cpu/kernel/keccak_util.rs:6-31,
 * keccak_sponge_stark.rs:410); k states of 25 words, in place. */
int zkm_poseidon_permute_batch(zkm_ctx* ctx, uint64_t* states, size_t k, char** err);
int zkm_keccakf_batch(zkm_ctx* ctx, uint64_t* states, size_t k, char** err);

/* ------------------------------------------------------------------ a13: PoseidonStark witness
 * PoseidonStark::generate_trace (poseidon_stark.rs:104-160): 262 columns x 2^log_n rows, column-major,
 * from `num_perms` seeded 12-element inputs (SplitMix64(seed); timestamp 0, FILTER 1) padded with the
 * default row (permutation of zeros, FILTER 0).  out must be a device pointer. */
#define ZKM_POSEIDON_COLS 262
int zkm_poseidon_trace(zkm_ctx* ctx, uint64_t seed, size_t num_perms, unsigned log_n, uint64_t* out_dev, char** err);

/* PoseidonStark::generate_trace for explicit permutation inputs: inputs = num_perms x 12 field elements, one timestamp each
 * (host or device); rows past num_perms are the default row (FILTER 0).  This is the table the PoseidonSponge rows look up
 * (all_stark.rs:169-195). */
int zkm_poseidon_trace_inputs(zkm_ctx* ctx, const uint64_t* inputs, const uint64_t* timestamps, size_t num_perms, unsigned log_n,
                              uint64_t* out_dev, char** err);

/* ------------------------------------------------------------------ N2: PoseidonSpongeStark witness
 * PoseidonSpongeStark::generate_trace (poseidon_sponge/poseidon_sponge_stark.rs:186-381, column map poseidon_sponge/columns.rs:17-66):
 * 110 columns, one row per 32-byte block (len/32 + 1 rows per operation, pad10*1 on the final row), the block's eight
 * little-endian u32 words overwrite the rate, one Poseidon permutation per row; padding rows all-zero.  Operation
 * arguments as zkm_keccak_sponge_trace. */
#define ZKM_POSEIDON_SPONGE_COLS 110
int zkm_poseidon_sponge_trace(zkm_ctx* ctx, const uint8_t* inputs, const uint64_t* input_off, const uint64_t* meta, size_t nops,
                              unsigned log_n, uint64_t* out_dev, size_t* rows_used_out, char** err);

/* ------------------------------------------------------------------ a12: KeccakSpongeStark witness (BASELINE config 5)
 * KeccakSpongeStark::generate_trace (keccak_sponge/keccak_sponge_stark.rs:222-251, rows :253-438, column map
 * keccak_sponge/columns.rs:19-70): 470 columns, one row per 136-byte block (len/136 + 1 rows per operation, pad10*1 on
 * the final row :334-341), one Keccak-f[1600] per row (:410), padding rows all-zero (:440-444).
 *   inputs        concatenated input bytes of all operations (host or device)
 *   input_off     nops + 1 byte offsets into `inputs` (host)
 *   meta          nops x 4 words (host): context, segment, virt_base, timestamp; word i of the input is read at
 *                 virt_base + i (contiguous base addresses)
 * Output: 470 x 2^log_n column-major, device pointer.  Fails if the operations need more than 2^log_n rows or if an
 * operation is empty (the reference indexes base_address[0]). */
#define ZKM_KECCAK_SPONGE_COLS 470
int zkm_keccak_sponge_trace(zkm_ctx* ctx, const uint8_t* inputs, const uint64_t* input_off, const uint64_t* meta, size_t nops,
                            unsigned log_n, uint64_t* out_dev, size_t* rows_used_out, char** err);

/* ------------------------------------------------------------------ a12: KeccakStark witness
 * KeccakStark::generate_trace (keccak/keccak_stark.rs:62-236, register map keccak/columns.rs): 2431 columns, 24 rows per
 * permutation (round flags, timestamp, the round's input limbs A, and the bit-decomposed theta / rho-pi / chi / iota
 * intermediates C, C', A', A'', A'''[0,0]); rows past 24*nperms are zero.
 *   inputs      nperms x 25 u64 permutation inputs, A(x, y) = inputs[p][5y + x] (host or device)
 *   timestamps  nperms u64 (host or device)
 * Output: 2431 x 2^log_n column-major, device pointer.  Fails if 24*nperms > 2^log_n. */
#define ZKM_KECCAK_COLS 2431
int zkm_keccak_trace(zkm_ctx* ctx, const uint64_t* inputs, const uint64_t* timestamps, size_t nperms, unsigned log_n,
                     uint64_t* out_dev, char** err);

/* ------------------------------------------------------------------ N2: SHA-256 message-schedule witnesses
 * ShaExtendStark::generate_trace (sha_extend/sha_extend_stark.rs:121-236): 78 columns, one row per schedule step; inputs =
 * nrows x 16 bytes (w[i-15], w[i-2], w[i-16], w[i-7] as little-endian byte quadruples), one timestamp per row; rows past
 * nrows are zero.
 * ShaExtendSpongeStark::generate_trace (sha_extend_sponge/sha_extend_sponge_stark.rs:131-215) for nblocks complete schedules:
 * 76 columns, 48 rows per block; w16 = nblocks x 16 uint32 words w[0..15]; meta = nblocks x 4 {context, segment, address of
 * w[0], timestamp of round 0}; word j lives at address + 4 j and round r is stamped timestamp + 20 r (2 * NUM_CHANNELS). */
#define ZKM_SHA_EXTEND_COLS 78
#define ZKM_SHA_EXTEND_SPONGE_COLS 76
int zkm_sha_extend_trace(zkm_ctx* ctx, const uint8_t* inputs, const uint64_t* timestamps, size_t nrows, unsigned log_n, uint64_t* out_dev,
                         char** err);
int zkm_sha_extend_sponge_trace(zkm_ctx* ctx, const uint32_t* w16, const uint64_t* meta, size_t nblocks, unsigned log_n, uint64_t* out_dev,
                                char** err);

/* SHA-256 compression witnesses for ncomp compressions: hx = ncomp x 8 words (a..h before the block), w = ncomp x 64 schedule
 * words, meta = ncomp x 8 {context, segment, address of hx[0], timestamp, address of w[0], segment of w, context of w, unused}.
 * ShaCompressStark::generate_trace (sha_compress/sha_compress_stark.rs:227-400; rows as emitted by witness/util.rs:605-690):
 * 224 columns, 65 rows per compression -- rounds 0..63 on the running state with w_i and K_i, then the final state with
 * w_i = k_i = 0.  ShaCompressSpongeStark::generate_trace (sha_compress_sponge/sha_compress_sponge_stark.rs:118-230): 127
 * columns, one row per compression (hx, the compressed state, hx + state with carry flags, addresses). */
#define ZKM_SHA_COMPRESS_COLS 224
#define ZKM_SHA_COMPRESS_SPONGE_COLS 127
int zkm_sha_compress_trace(zkm_ctx* ctx, const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t ncomp, unsigned log_n,
                           uint64_t* out_dev, char** err);
int zkm_sha_compress_sponge_trace(zkm_ctx* ctx, const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t ncomp, unsigned log_n,
                                  uint64_t* out_dev, char** err);

/* ------------------------------------------------------------------ N2: LogicStark witness
 * LogicStark::generate_trace (logic.rs:150-183) with Operation::into_row (:122-142): 69 columns x 2^log_n rows,
 * column-major; row r < nops holds operation r (flag column, the 32 little-endian bits of each input, the result),
 * the remaining rows are zero.  ops = nops x 3 uint32 words {op (0 and, 1 or, 2 xor, 3 nor), input0, input1}, host or
 * device.  Fails if nops > 2^log_n or an op code is out of range. */
#define ZKM_LOGIC_COLS 69
int zkm_logic_trace(zkm_ctx* ctx, const uint32_t* ops, size_t nops, unsigned log_n, uint64_t* out_dev, char** err);

/* ------------------------------------------------------------------ Fiat-Shamir (host)
 * plonky2 Challenger<F, PoseidonHash> (uses at prover.rs:182-190, 466, 524-527, 588-591, 610). */
typedef struct {
    uint64_t state[12];
    uint64_t in_buf[8];
    uint64_t out_buf[8];
    uint32_t n_in, n_out;
} zkm_challenger;
void zkm_challenger_init(zkm_challenger* ch);
void zkm_challenger_observe(zkm_challenger* ch, const uint64_t* elems, size_t n);
uint64_t zkm_challenger_get(zkm_challenger* ch);
void zkm_challenger_compact(zkm_challenger* ch, uint64_t state_out[12]);

/* ------------------------------------------------------------------ a5: prove_single_table
 * StarkConfig::standard_fast_config (prover/src/config.rs:17-30). */
typedef struct {
    unsigned rate_bits, cap_height, pow_bits, num_challenges, num_queries, arity_bits, final_poly_bits;
} zkm_stark_config;
void zkm_standard_config(zkm_stark_config* cfg);

/* Tables with a constraint kernel (the reference's Table enum, all_stark.rs:96-110, has 12):
 *   POSEIDON       poseidon/poseidon_stark.rs:554-594   262 columns
 *   LOGIC          logic.rs:199-248                       69 columns
 *   KECCAK_SPONGE  keccak_sponge_stark.rs:456-567        470 columns
 *   KECCAK         keccak/keccak_stark.rs:256-413       2431 columns
 *   MEMORY         memory/memory_stark.rs:253-341         13 columns, plus the range-check lookup :476-483
 *   POSEIDON_SPONGE poseidon_sponge/poseidon_sponge_stark.rs:383-478  110 columns
 *   SHA_EXTEND     sha_extend/sha_extend_stark.rs:238-317              78 columns
 *   SHA_EXTEND_SPONGE sha_extend_sponge/sha_extend_sponge_stark.rs:220-330  76 columns
 *   SHA_COMPRESS   sha_compress/sha_compress_stark.rs:402-606         224 columns
 *   SHA_COMPRESS_SPONGE sha_compress_sponge/sha_compress_sponge_stark.rs:233-268  127 columns
 *   ARITHMETIC     arithmetic/arithmetic_stark.rs:214-240 and its nine operation modules, 54 columns, plus the 18-column range-check lookup
 *                  :269-276 (at least 2^16 rows); rows come from the CPU-side witness generator (no witness kernel) */
#define ZKM_TABLE_POSEIDON 0
#define ZKM_TABLE_LOGIC 1
#define ZKM_TABLE_KECCAK_SPONGE 2
#define ZKM_TABLE_KECCAK 3
#define ZKM_TABLE_MEMORY 4
#define ZKM_TABLE_POSEIDON_SPONGE 5
#define ZKM_TABLE_SHA_EXTEND 6
#define ZKM_TABLE_SHA_EXTEND_SPONGE 7
#define ZKM_TABLE_SHA_COMPRESS 8
#define ZKM_TABLE_SHA_COMPRESS_SPONGE 9
#define ZKM_TABLE_ARITHMETIC 10
#define ZKM_TABLE_CPU 11
/* Position of a table in the reference's Table enum (all_stark.rs:96-110: Arithmetic = 0, Cpu, Poseidon, PoseidonSponge, Keccak,
 * KeccakSponge, ShaExtend, ShaExtendSponge, ShaCompress, ShaCompressSponge, Logic, Memory = 11), or -1 for an unknown id.  The
 * ZKM_TABLE_* ids above are NOT in that order.  prove_with_traces (prover.rs:144-200, 234-438) commits, observes and proves the
 * tables in Table::all() order on one transcript: zkm_prove_with_traces / zkm_prove_segment_image use the caller's array order and
 * reject an array that holds all twelve tables in any other order (sub-segments of fewer tables, as the tests prove them, are the
 * caller's responsibility). */
int zkm_table_enum_index(int table_id);
#define ZKM_ARITHMETIC_COLS 54
#define ZKM_CPU_COLS 259
#define ZKM_MEMORY_COLS 13
size_t zkm_table_width(int table_id); /* 0 for an unknown id */
/* Auxiliary columns the table's own logUp lookups (Stark::lookups(), lookup.rs:22-40) put in front of the CTL columns:
 * per lookup and challenge, ceil(columns / 2) helper columns and one Z (stark.rs:217-223).  0 for tables without lookups.
 * The naux arguments of the prove entry points count the CTL columns only; zkm_proof_words takes the total. */
size_t zkm_num_lookup_columns(int table_id, const zkm_stark_config* cfg);

/* Proof blob (uint64_t words) -- the fields of StarkProofWithMetadata (proof.rs:178-201) flattened:
 *   [0] magic "ZKMPROOF" [1] degree_bits [2] W trace cols [3] A aux cols [4] Q quotient polys [5] Z ctl zs
 *   [6] cap_height [7] L fri layers [8] F final_poly_len [9] num_queries [10] rate_bits [11] arity_bits
 *   [12..15] 0
 *   init_challenger_state[12]
 *   trace_cap[C*4] aux_cap[C*4] quotient_cap[C*4]                    C = 2^cap_height
 *   local_values[2W] next_values[2W] aux[2A] aux_next[2A] ctl_zs_first[Z] quotient[2Q]
 *   commit_phase_merkle_caps[L][C*4]   final_poly[2F]   pow_witness[1]
 *   query_round_proofs[num_queries]:
 *       oracle 0..2: evals[ncols_o], siblings[(lde_bits - cap_height)*4]
 *       layer i in 0..L: evals[2*2^arity_bits], siblings[(lde_bits - arity_bits*(i+1) - cap_height)*4]
 */
#define ZKM_PROOF_MAGIC 0x5a4b4d50524f4f46ULL
size_t zkm_proof_words(const zkm_stark_config* cfg, unsigned log_n, size_t ncols, size_t naux, size_t nctl_zs);

/* N4, proof hand-off: word offsets of every field of a proof blob, read from the blob's own header, so the caller can rebuild
 * StarkProofWithMetadata (proof.rs:178-201), StarkOpeningSet (:281-297) and plonky2's FriProof with plain slice copies.  F2 values
 * are pairs of consecutive words (c0, c1); digests are 4 words.  Returns 0, or nonzero if the header is not a proof header. */
typedef struct {
    uint64_t degree_bits, trace_cols, aux_cols, quotient_polys, ctl_zs, cap_height, fri_layers, final_poly_len, num_queries, rate_bits,
        arity_bits;
    size_t total_words;
    size_t init_challenger_state;                 /* 12 words */
    size_t trace_cap, aux_cap, quotient_cap;      /* 2^cap_height digests each */
    size_t local_values, next_values;             /* trace_cols F2 values each */
    size_t aux_polys, aux_polys_next;             /* aux_cols F2 values each */
    size_t ctl_zs_first;                          /* ctl_zs base-field words */
    size_t quotient_polys_open;                   /* quotient_polys F2 values */
    size_t commit_phase_merkle_caps;              /* fri_layers x 2^cap_height digests */
    size_t final_poly;                            /* final_poly_len F2 coefficients */
    size_t pow_witness;                           /* 1 word */
    size_t query_round_proofs, query_round_words; /* num_queries rounds of query_round_words words, see zkm_proof_query_layout */
} zkm_proof_layout;
int zkm_proof_get_layout(const uint64_t* proof, zkm_proof_layout* out);
/* Offsets inside ONE query round (relative to query_round_proofs + q * query_round_words):
 *   oracle o in {0 trace, 1 aux, 2 quotient}: evals at oracle_evals[o] (oracle_cols[o] base-field words), Merkle siblings at
 *   oracle_siblings[o] (initial_siblings digests) -- FriInitialTreeProof;
 *   layer i < fri_layers: 2^arity_bits F2 evals at layer_evals[i], layer_siblings_count[i] digests at layer_siblings[i] -- FriQueryStep.
 * Arrays are sized for at most 16 layers.  recover_degree_bits (proof.rs:205-212) reads initial_siblings + cap_height - rate_bits. */
typedef struct {
    size_t oracle_evals[3], oracle_cols[3], oracle_siblings[3], initial_siblings;
    size_t layer_evals[16], layer_siblings[16], layer_siblings_count[16];
} zkm_proof_query_layout;
int zkm_proof_get_query_layout(const uint64_t* proof, zkm_proof_query_layout* out);

/* prove_single_table (prover.rs:441-641) for table `table_id`.
 *   trace       ncols x 2^log_n trace values (host or device) -- used only if trace_batch is NULL
 *   trace_batch optional existing commitment of the trace (prover.rs:445); NULL = commit here
 *   aux         naux x 2^log_n auxiliary values = ctl helper columns ++ ctl z columns (prover.rs:497-508)
 *   num_helpers helper-column count of each of the nctl_zs CtlZData (cross_table_lookup.rs:474-481)
 *   challenger  in/out transcript (host)
 *   proof_out   host buffer of zkm_proof_words() words */
int zkm_prove_single_table(zkm_ctx* ctx, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols,
                           unsigned log_n, const zkm_batch* trace_batch, const uint64_t* aux, size_t naux,
                           const uint32_t* num_helpers, size_t nctl_zs, zkm_challenger* challenger, uint64_t* proof_out,
                           char** err);


/* K proofs of the same table at the same height in lock-step (cf. zkm_prove_segments): K calls of zkm_prove_single_table -- traces[k],
 * aux[k], challengers[k], proofs_out[k] as that entry point takes them (trace values required; the benchmark's CtlData shape) -- as one:
 * stacked trace / auxiliary / quotient commitments, one launch per stage and one transcript round trip per stage for all K.  Every
 * blob and every challenger state equals the single call's.  At most 32 proofs per call; tables with lookups of their own (Memory,
 * Arithmetic) go through zkm_prove_segments.  ~14.6 GB of HBM per 262 x 2^20 proof in flight. */
int zkm_prove_single_tables(zkm_ctx* ctx, int table_id, const zkm_stark_config* cfg, size_t nproofs, const uint64_t* const* traces, size_t ncols,
                            unsigned log_n, const uint64_t* const* aux, size_t naux, const uint32_t* num_helpers, size_t nctl_zs,
                            zkm_challenger* const* challengers, uint64_t* const* proofs_out, char** err);

/* ------------------------------------------------------------------ a3/a7: cross-table lookups, data-driven
 * Column / Filter / TableWithColumns / CrossTableLookup (prover/src/cross_table_lookup.rs:31-415) restated as
 * plain arrays so that the same description drives CTL data generation (a3), the CTL constraint checks inside
 * the quotient kernel (a7) and the verifier. */
typedef struct {            /* Column<F>: sum_k coeff_k*local[col_k] + sum_k coeff_k*next[col_k] + constant */
    uint32_t n_local, n_next, term_off, _pad;   /* terms [term_off, +n_local) are local, the next n_next are next-row */
    uint64_t constant;
} zkm_column;
typedef struct {            /* TableWithColumns minus the table id: Vec<Column> + Option<Filter> */
    uint32_t ncols, col_off;                    /* columns[col_off .. col_off+ncols) */
    uint32_t has_filter, nprod, prod_off, nconst, const_off, _pad;
    /* Filter = sum_{k<nprod} columns[filter_idx[prod_off+2k]] * columns[filter_idx[prod_off+2k+1]]
     *        + sum_{k<nconst} columns[filter_idx[const_off+k]]                       (cross_table_lookup.rs:40-103) */
} zkm_colset;
typedef struct {            /* all descriptor arrays of one table (host pointers) */
    const zkm_column* columns; size_t ncolumns;
    const uint32_t* term_col; const uint64_t* term_coeff; size_t nterms;
    const zkm_colset* colsets; size_t ncolsets;
    const uint32_t* filter_idx; size_t nfilter_idx;
} zkm_ctl_table;
typedef struct {            /* CtlZData (cross_table_lookup.rs:424-438) */
    uint32_t ncolsets, colset_off;              /* colset_ids[colset_off .. +ncolsets) index the table's colsets */
    uint32_t num_helpers, _pad;                 /* helper columns of this Z: ceil(ncolsets/2) if ncolsets > 1 else 0
                                                   (the benchmark's fake CTL data has ncolsets == 0, num_helpers == 1) */
    uint64_t beta, gamma;                       /* GrandProductChallenge */
} zkm_ctl_z;
typedef struct { uint32_t table, colset; } zkm_ctl_side;
typedef struct { uint32_t nlooking, looking_off; zkm_ctl_side looked; } zkm_cross_table_lookup;  /* looking sides: sides[looking_off..] */

/* a3: cross_table_lookup_data for one table (get_helper_cols :709-797, partial_sums :841-872): helper columns and
 * the upside-down running-sum Z of every CtlZData.  aux_out = all helper columns (zs order) ++ all Z columns,
 * column-major (sum num_helpers + nzs) x 2^log_n, host or device; trace = ncols x 2^log_n values, host or device. */
int zkm_ctl_data(zkm_ctx* ctx, const zkm_ctl_table* table, const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs,
                 const uint64_t* trace, size_t ncols, unsigned log_n, uint64_t* aux_out, char** err);

/* a4: lookup_helper_columns (lookup.rs:46-124): logUp helper columns of one Lookup for one challenge --
 *   h_j[i] = sum over the (<= 2) looking columns f of chunk j of filter_f[i] / (challenge + f[i]),
 *   Z[0] = 0, Z[i+1] = Z[i] + sum_j h_j[i] - frequencies[i] / (challenge + table[i]).
 * colset_ids: the nlookup looking columns as single-column sets (column + optional filter); table_col / freq_col:
 * column indices of Lookup::table_column / frequencies_column.  out = (ceil(nlookup/2) + 1) x 2^log_n: helpers, then Z
 * (the order prove_single_table appends them, prover.rs:480-493). */
int zkm_lookup_helper_columns(zkm_ctx* ctx, const zkm_ctl_table* table, const uint32_t* colset_ids, size_t nlookup, uint32_t table_col,
                              uint32_t freq_col, uint64_t challenge, const uint64_t* trace, size_t ncols, unsigned log_n,
                              uint64_t* out, char** err);

/* prove_single_table with real CTL data: like zkm_prove_single_table, the CtlZData described by (table, zs, colset_ids).
 * lookup_challenges: for a table with its own lookups (zkm_num_lookup_columns() > 0) the num_challenges lookup challenges --
 * the betas of the CTL challenges (prover.rs:468-474); the helper columns are computed here (prover.rs:475-493) from the
 * trace VALUES, so `trace` is required for such tables even when trace_batch is given.  NULL otherwise. */
int zkm_prove_single_table_ctl(zkm_ctx* ctx, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols,
                               unsigned log_n, const zkm_batch* trace_batch, const uint64_t* aux, size_t naux,
                               const zkm_ctl_table* table, const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs,
                               const uint64_t* lookup_challenges, zkm_challenger* challenger, uint64_t* proof_out, char** err);

/* a1: prove_with_traces (prover.rs:130-232): commit every trace, seed the transcript with all trace caps and the
 * public values, draw the CTL challenges, build every table's CtlData, then prove the tables in order on the shared
 * transcript.  Output: per-table proof blobs concatenated (offsets in proof_offsets_out[ntables+1]) and the
 * num_challenges (beta, gamma) pairs. */
typedef struct {
    int table_id;
    const uint64_t* trace;      /* ncols x 2^log_n in one block, host or device; NULL when `columns` is given */
    size_t ncols;
    unsigned log_n;
    const zkm_ctl_table* ctl;   /* column sets referenced by the cross-table lookups */
    const uint64_t* const* columns;  /* or: one pointer per column (2^log_n words each, host or device) -- the reference's
                                        Vec<PolynomialValues<F>> (prover.rs:130-142) without a host-side flatten; NULL = use `trace` */
} zkm_table_input;
size_t zkm_all_proof_words(const zkm_stark_config* cfg, const zkm_table_input* tables, size_t ntables,
                           const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides, size_t nctls, size_t* proof_offsets_out);
int zkm_prove_with_traces(zkm_ctx* ctx, const zkm_stark_config* cfg, const zkm_table_input* tables, size_t ntables,
                          const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides, size_t nctls,
                          const uint64_t* public_values, size_t npublic, uint64_t* proofs_out, uint64_t* ctl_challenges_out,
                          char** err);

/* The AllStark of the reference (all_stark.rs:96-155, 136-542) ships with the library, so a caller proves a whole segment from
 * the twelve traces alone: zkm_all_stark_ctls returns the fifteen cross-table lookups (sides name tables by their position in
 * the Table enum, all_stark.rs:96-110), zkm_all_stark_ctl_table the column sets of one table, and zkm_prove_segment ==
 * prove_with_traces (prover.rs:130-232) on traces[t] / log_n[t] of Table::all()[t] (widths: zkm_table_width).  Pass
 * proofs_out = NULL to size the output first: proof_offsets_out[13] (offsets of the twelve blobs + the total).
 * (csrc/all_stark_ctl.inc is generated from zkm_amd/tables.py by tools/gen_all_stark_ctl.py.) */
int zkm_all_stark_ctls(const zkm_cross_table_lookup** ctls_out, size_t* nctls_out, const zkm_ctl_side** sides_out, size_t* nsides_out);
const zkm_ctl_table* zkm_all_stark_ctl_table(int table_id);
int zkm_prove_segment(zkm_ctx* ctx, const zkm_stark_config* cfg, const uint64_t* const* traces, const unsigned* log_n,
                      const uint64_t* public_values, size_t npublic, uint64_t* proofs_out, size_t* proof_offsets_out,
                      uint64_t* ctl_challenges_out, char** err);
/* zkm_prove_segment with one pointer per column: columns[t][i] -> column i of table t (Table::all() order), 2^log_n[t] words. */
int zkm_prove_segment_columns(zkm_ctx* ctx, const zkm_stark_config* cfg, const uint64_t* const* const* columns, const unsigned* log_n,
                              const uint64_t* public_values, size_t npublic, uint64_t* proofs_out, size_t* proof_offsets_out,
                              uint64_t* ctl_challenges_out, char** err);

/* K INDEPENDENT segments in lock-step: the reference proves the segments of a program one after the other, each a prove_with_traces
 * on its own transcript (prover/examples/utils/src/utils.rs:57-68, 105-133; prover.rs:130-232).  Here the K segments of one call
 * advance through every stage of prove_with_traces TOGETHER: for every table, the segments whose table has the same height form a
 * group (at most "max_stack", see zkm_ctx_set_tuning; heights may differ between segments -- a table's segments of another height form
 * another group), a group's trace / auxiliary / quotient commitments, CTL data, constraint evaluation, openings, FRI layers, proof of
 * work and query gathers are ONE launch (or group of launches) for all its segments, and each of its transcript round trips (caps,
 * opening partials, final polynomial, proof-of-work witness, query rounds) brings the words of all of them down together.  The
 * transcripts stay per segment, on the host; every proof blob and every challenge is word for word what zkm_prove_segment returns
 * for that segment alone (tests/test_segments_batch.py).  A 2^16-cycle segment is mostly tables of 2^6 .. 2^13 rows whose launches
 * cannot fill the GPU one segment at a time: K of them do.
 *   traces[s][t]        table t (Table::all() order) of segment s: zkm_table_width x 2^log_n[s][t] words, host or device
 *   public_values[s]    npublic[s] words (arrays may be NULL when no segment has public values)
 *   proofs_out[s]       zkm_prove_segment(.., proofs_out = NULL, ..) sizes a segment's buffer and gives its thirteen offsets
 *   ctl_challenges_out[s]   2 * num_challenges words
 * Memory: every commitment of every segment of a group is alive until its table has been proven (~1.8 GB of HBM per 2^16-cycle segment);
 * a call with more segments than 80 % of (the blocks this context has cached + the memory that is free) holds is proven in
 * consecutive, equally sized waves (the words are the same; only fewer segments share a launch).
 * A failure leaves every output undefined (no partial results). */
int zkm_prove_segments(zkm_ctx* ctx, const zkm_stark_config* cfg, size_t nseg, const uint64_t* const* const* traces,
                       const unsigned* const* log_n, const uint64_t* const* public_values, const size_t* npublic, uint64_t* const* proofs_out,
                       uint64_t* const* ctl_challenges_out, char** err);
/* zkm_prove_segments with one pointer per column: columns[s][t][i] -> column i of table t of segment s (2^log_n[s][t] words) -- K of the
 * reference's [Vec<PolynomialValues<F>>; NUM_TABLES] (prover.rs:130-142) as they come out of generate_traces, nothing flattened. */
int zkm_prove_segments_columns(zkm_ctx* ctx, const zkm_stark_config* cfg, size_t nseg, const uint64_t* const* const* const* columns,
                               const unsigned* const* log_n, const uint64_t* const* public_values, const size_t* npublic,
                               uint64_t* const* proofs_out, uint64_t* const* ctl_challenges_out, char** err);

/* ------------------------------------------------------------------ staged traces: the NEXT proof's upload behind the CURRENT proof
 * The reference commits from host Vecs (prover.rs:144-167).  A prove call handed a host trace pipelines the upload INSIDE the proof
 * (column chunks absorbed as they arrive: zkm_batch_commit_values); a driver that has the next segment's trace while the current one is
 * being proven -- the witness generator runs ahead of the prover -- stages it instead: zkm_trace_stage[_columns] queues the upload of
 * an ncols x 2^log_n host matrix on the context's two copy streams (alternate pieces of >= 64 MB: 8 columns at 2^20 rows, a short table in one copy) into a block of the context's
 * allocator and RETURNS AT ONCE; zkm_staged_ptr gives the device matrix to pass as `trace` / `traces[k]` of a later prove call ON THE
 * SAME CONTEXT, ordered behind the upload on the context's compute stream (a device-side wait; NULL on a runtime error).  That call
 * runs the device-resident path at full speed while the copy engines bring in the trace after it.
 *   canonical   nonzero: the caller vouches that every word is < p; 0: the staged copy is canonicalised once when first consumed
 *   host memory pinned (zkm_host_alloc, or a Vec under zkm_host_register): the copies are asynchronous.  Pageable memory works and
 *               makes zkm_trace_stage itself take the time of the upload (the runtime stages it through its own pinned buffer).
 *   The host matrix must stay valid and unchanged until zkm_staged_ready(h, ..) returns 1 or zkm_staged_free(h) has returned.
 * zkm_staged_ready: 1 = uploaded, 0 = in flight (wait = 0 only), -1 = runtime error.  zkm_staged_free waits for the upload and returns
 * the block; call it after the prove call that consumed the matrix has returned. */
typedef struct zkm_staged zkm_staged;
int zkm_trace_stage(zkm_ctx* ctx, const uint64_t* values, size_t ncols, unsigned log_n, int canonical, zkm_staged** out, char** err);
int zkm_trace_stage_columns(zkm_ctx* ctx, const uint64_t* const* columns, size_t ncols, unsigned log_n, int canonical, zkm_staged** out,
                            char** err);
/* All twelve tables of ONE segment in one call (Table::all() order; traces[t]: zkm_table_width x 2^log_n[t] words, or columns[t][i]): one
 * device block, one pair of events; zkm_staged_segment_ptrs gives the twelve device matrices to pass as traces[s] of zkm_prove_segments.
 * A lock-step call of K segments then costs K stage calls instead of 12 K. */
int zkm_segment_stage(zkm_ctx* ctx, const uint64_t* const* traces, const unsigned* log_n, int canonical, zkm_staged** out, char** err);
int zkm_segment_stage_columns(zkm_ctx* ctx, const uint64_t* const* const* columns, const unsigned* log_n, int canonical, zkm_staged** out,
                              char** err);
int zkm_staged_segment_ptrs(zkm_staged* staged, const uint64_t** ptrs_out);
const uint64_t* zkm_staged_ptr(zkm_staged* staged);
int zkm_staged_ready(zkm_staged* staged, int wait);
void zkm_staged_free(zkm_staged* staged);

/* ------------------------------------------------------------------ one process, many GPUs: a pool of contexts
 * The reference drives all segments of a program from ONE process (prover/examples/utils/src/utils.rs:57-68 prove_single_seg_common,
 * :105-133 prove_multi_seg_common: a loop of prove_with_traces calls); segments are independent proofs, so N GPUs take them
 * side by side with no data-path collective.  A pool owns `contexts_per_device` contexts on each of `devices[0 .. ndevices)` (a device
 * may be listed once only) and, per call, one worker thread per context: the nseg segments of a call are cut into consecutive groups of
 * at most `max_stack` segments (every worker the same number of groups, sizes as even as the count allows), the workers pull groups
 * from one queue, and a group is ONE lock-step call (zkm_prove_segments[_columns]) on the worker's context.  Every proof blob and
 * every challenge is word for word what zkm_prove_segment returns for that segment alone, whatever the device count.
 *   traces / columns / log_n / public_values / npublic / proofs_out / ctl_challenges_out   as zkm_prove_segments[_columns]
 *   max_stack   segments per lock-step call, 1 .. 32 (0 = 8; also bounded by the contexts' "max_stack" tuning)
 * The inputs must be readable from every device of the pool: HOST memory (pageable, zkm_host_alloc'ed or zkm_host_register'ed -- what a
 * Rust Vec is), or device memory when the pool has one device.  The first failing group stops the queue; its message names the worker,
 * the device and the positions of its segments in the call.  A failure leaves every output undefined.  A pool is single-owner like
 * a context: one call at a time.  zkm_pool_context hands out worker w's context (0 <= w < zkm_pool_workers) for zkm_ctx_set_tuning,
 * zkm_ctx_memory, zkm_profile_*; zkm_pool_set_tuning applies one key to all of them. */
typedef struct zkm_pool zkm_pool;
int zkm_pool_create(const int* devices, size_t ndevices, size_t contexts_per_device, zkm_pool** out, char** err);
void zkm_pool_destroy(zkm_pool* pool);
size_t zkm_pool_workers(const zkm_pool* pool);
zkm_ctx* zkm_pool_context(zkm_pool* pool, size_t worker);
int zkm_pool_device(const zkm_pool* pool, size_t worker);
int zkm_pool_set_tuning(zkm_pool* pool, const char* key, uint64_t value, char** err);
int zkm_pool_prove_segments(zkm_pool* pool, const zkm_stark_config* cfg, size_t nseg, size_t max_stack, const uint64_t* const* const* traces,
                            const unsigned* const* log_n, const uint64_t* const* public_values, const size_t* npublic,
                            uint64_t* const* proofs_out, uint64_t* const* ctl_challenges_out, char** err);
int zkm_pool_prove_segments_columns(zkm_pool* pool, const zkm_stark_config* cfg, size_t nseg, size_t max_stack,
                                    const uint64_t* const* const* const* columns, const unsigned* const* log_n,
                                    const uint64_t* const* public_values, const size_t* npublic, uint64_t* const* proofs_out,
                                    uint64_t* const* ctl_challenges_out, char** err);
/* The groups a pool call of nseg segments is cut into for `workers` workers (a pure function: no pool, no GPU): returns their number and
 * writes the first min(capacity, number) sizes, in segment order.  20 segments, 2 workers, max_stack 4 -> 4, 4, 3, 3, 3, 3. */
size_t zkm_pool_plan(size_t nseg, size_t workers, size_t max_stack, size_t* group_sizes_out, size_t capacity);
/* which worker proved segment s of the pool's LAST call, and in which group (diagnostics / tests): 0 on success */
int zkm_pool_last_assignment(const zkm_pool* pool, size_t segment, size_t* worker_out, size_t* group_out);

/* a10 alone (BASELINE config 4): PolynomialBatch::prove_openings (call site prover.rs:618-628) for the STARK FRI
 * instance (stark.rs:91-148: batches at zeta, g*zeta and 1 over the trace / auxiliary / quotient oracles) on three
 * existing commitments.  Transcript: compact, zeta <- challenger, openings observed (proof.rs:336-367), prove_openings.
 * The last nctl_zs auxiliary polynomials are the ones opened at 1.  Output: the usual proof blob. */
int zkm_prove_openings(zkm_ctx* ctx, const zkm_stark_config* cfg, const zkm_batch* trace_batch, const zkm_batch* aux_batch,
                       const zkm_batch* quot_batch, size_t nctl_zs, zkm_challenger* challenger, uint64_t* proof_out, char** err);

/* PolynomialBatch::prove_openings (plonky2 fri/oracle.rs) for an ARBITRARY FriInstanceInfo: `noracles` (<= 8) commitments of equal
 * degree, `nbatches` FriBatchInfo { point, polynomials: (oracle_index, polynomial_index)... } in instance order.  Transcript as in
 * plonky2: alpha <- challenger; per batch the alpha-reduced polynomial divided by (X - point), batches chained with
 * ReducingFactor::shift_poly; commit phase with one cap + beta per reduction arity; final polynomial; proof of work; query rounds.
 * (The openings themselves are the caller's: observe them before the call, as StarkOpeningSet / OpeningSet do.)  With the three
 * STARK oracles and the three batches of stark.rs:91-148 the output equals the FRI part of zkm_prove_openings' blob.
 * FRI proof blob: [0] magic "ZKMFRIPF" [1] degree_bits [2] noracles [3] cap_height [4] L fri layers [5] F final_poly_len
 *   [6] num_queries [7] rate_bits [8] arity_bits [9..15] 0 [16..23] columns of each oracle
 *   commit_phase_merkle_caps[L][C*4]  final_poly[2F]  pow_witness[1]
 *   query_round_proofs[num_queries]: per oracle evals[cols] + siblings[(lde_bits - cap_height)*4]; per layer evals[2*2^arity_bits] +
 *   siblings[(lde_bits - arity_bits*(i+1) - cap_height)*4] */
#define ZKM_FRI_PROOF_MAGIC 0x46504952464d4b5aULL
typedef struct { uint32_t oracle, poly; } zkm_fri_poly;
typedef struct { uint64_t point[2]; const zkm_fri_poly* polys; size_t npolys; } zkm_fri_batch;
size_t zkm_fri_proof_words(const zkm_stark_config* cfg, unsigned log_n, const size_t* oracle_cols, size_t noracles);
int zkm_fri_prove(zkm_ctx* ctx, const zkm_stark_config* cfg, const zkm_batch* const* oracles, size_t noracles, const zkm_fri_batch* batches,
                  size_t nbatches, zkm_challenger* challenger, uint64_t* proof_out, char** err);

/* ------------------------------------------------------------------ stage entry points (parity / reuse)
 * a6: compute_quotient_polys (prover.rs:645-789): nalphas polys of 2n coefficients, natural order;
 * out host or device. */
/* ------------------------------------------------------------------ N3: trace ingest
 * A segment's traces as one file / memory image, the hand-off from the CPU-side witness generator (generate_traces,
 * witness/traces.rs:230-320 returns [Vec<PolynomialValues<F>>; 12]; each table is written column-major as util.rs:37-46 lays it
 * out).  All integers little-endian uint64 unless noted:
 *   [0] magic "ZKMTRACE"  [1] version 1  [2] ntables  [3] npublic  [4] nctls  [5] nsides  [6..7] 0
 *   public values[npublic]
 *   per table (8 words): table_id, ncols, log_n, data offset (words from file start), ncolumns, nterms, ncolsets, nfilter_idx
 *   per table, its column-set description: columns[ncolumns] (zkm_column, 3 words each), term_col[nterms] (uint32, padded to 8 B),
 *       term_coeff[nterms], colsets[ncolsets] (zkm_colset, 4 words each), filter_idx[nfilter_idx] (uint32, padded to 8 B)
 *   cross-table lookups[nctls] (zkm_cross_table_lookup, 2 words each), sides[nsides] (zkm_ctl_side, 1 word each)
 *   trace data: per table ncols x 2^log_n canonical field elements
 * zkm_segment_image_words sizes an image; zkm_segment_image_write fills one from the in-memory description (traces must be host
 * pointers); zkm_prove_segment_image proves straight from an image (e.g. an mmap of the file): the traces are uploaded table by
 * table, never copied on the host. */
size_t zkm_segment_image_words(const zkm_table_input* tables, size_t ntables, const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides,
                               size_t nctls, size_t npublic);
int zkm_segment_image_write(const zkm_table_input* tables, size_t ntables, const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides,
                            size_t nctls, const uint64_t* public_values, size_t npublic, uint64_t* image_out, char** err);
/* proofs_out / ctl_challenges_out as zkm_prove_with_traces; *proof_words_out receives the total; pass proofs_out = NULL to query
 * the size first (returns 0 and only fills *proof_words_out and proof_offsets_out[ntables + 1] if non-NULL). */
int zkm_prove_segment_image(zkm_ctx* ctx, const zkm_stark_config* cfg, const uint64_t* image, size_t image_words, uint64_t* proofs_out,
                            size_t* proof_words_out, size_t* proof_offsets_out, uint64_t* ctl_challenges_out, char** err);

int zkm_quotient(zkm_ctx* ctx, int table_id, const zkm_batch* trace, const zkm_batch* aux, const uint32_t* num_helpers,
                 size_t nctl_zs, const uint64_t* alphas, size_t nalphas, uint64_t* out_coeffs, char** err);
/* check_constraints (prover.rs:793-910; the reference calls it under debug assertions, :529-541): evaluate the table's whole vanishing
 * polynomial -- table constraints, the table's lookups, the cross-table-lookup checks, in eval_vanishing_poly's order
 * (vanishing_poly.rs:17-46) -- on every row of the TRACE DOMAIN (rate_bits = 0: row i with next row i + 1 mod n, the Lagrange selectors
 * are the indicators of the first / last row) with the given constraint challenges, and report whether every accumulator vanishes.
 *   trace   ncols x 2^log_n trace values          aux   naux x 2^log_n auxiliary values: the table's lookup helper columns (zkm_num_lookup_columns,
 *           as zkm_lookup_helper_columns makes them per challenge), then the CTL helper columns and Zs (zkm_ctl_data) -- prover.rs:497-508 order
 *   table / zs / colset_ids / nzs, lookup_challenges   as zkm_prove_single_table_ctl
 * Returns 0 and *first_failing_row = ~0 when all constraints hold; nonzero with the reference's message "Constraint failed in <Stark>"
 * and the first failing row otherwise.  A debugging aid: not on the proving path. */
int zkm_check_constraints(zkm_ctx* ctx, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                          const uint64_t* aux, size_t naux, const zkm_ctl_table* table, const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs,
                          const uint64_t* lookup_challenges, const uint64_t* alphas, size_t nalphas, uint64_t* first_failing_row, char** err);
/* a9: StarkOpeningSet::new building block (proof.rs:299-334): p(zeta) in F2 for every polynomial of the
 * batch; out = ncols x 2 words, host. */
int zkm_eval_openings(zkm_ctx* ctx, const zkm_batch* b, const uint64_t zeta[2], uint64_t* out, char** err);

/* ------------------------------------------------------------------ profiling (HIP events on the ctx stream) */
void zkm_profile_enable(zkm_ctx* ctx, int on);
void zkm_profile_reset(zkm_ctx* ctx);
/* number of distinct kernel names recorded since reset */
size_t zkm_profile_count(zkm_ctx* ctx);
/* i-th record: name (static string), number of launches, summed device milliseconds */
int zkm_profile_get(zkm_ctx* ctx, size_t i, const char** name, uint64_t* launches, double* total_ms);

const char* zkm_version(void);

#ifdef __cplusplus
}
#endif
#endif
