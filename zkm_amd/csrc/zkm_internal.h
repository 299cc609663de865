// zkm_internal.h -- shared host-side declarations of libzkmhip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <stdexcept>
#include <string>
#include <initializer_list>
#include <vector>

#include "../../include/zkm_hip.h"
#include "gl_dev.h"

#define ZKM_HIP_CHECK(expr)                                                                                  \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                                     std::to_string(__LINE__) + ")");                                        \
    } while (0)

// an error that belongs to ONE segment of a stacked launch (position in the stack); what() is the reference's message
struct zkm_segment_error : std::runtime_error {
    size_t seg;
    zkm_segment_error(size_t s, const char* msg) : std::runtime_error(msg), seg(s) {}
};

struct zkm_prof_rec {
    const char* name;
    hipEvent_t start, stop;
};

#ifndef ZKM_MAX_SEG
#define ZKM_MAX_SEG 32       // segments one lock-step group may hold (per-segment challenges travel in kernel-argument arrays of this size)
#endif
struct seg_gl { gl_t v[ZKM_MAX_SEG]; };          // one base-field word per segment
struct seg_gl2 { gl_t v[2 * ZKM_MAX_SEG]; };     // two per segment (an F2 element, or the <= 2 constraint challenges)

#ifndef ZKM_COMMIT_LANES
#define ZKM_COMMIT_LANES 4   // trace commitments in flight per context (the context itself + 3 lanes)
#endif
#ifndef ZKM_LEAF_MFMA_DEFAULT
#define ZKM_LEAF_MFMA_DEFAULT 1
#endif

struct zkm_twiddles {
    gl_t* fwd = nullptr;  // per-stage tables concatenated: entry (1<<s) + j = w_{2^(s+1)}^j, j < 2^s
    gl_t* inv = nullptr;  // same with inverse roots
    unsigned log_max = 0;
};

struct zkm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // host -> device ingest, overlapped with the compute stream (created on first use)
    hipStream_t copy_stream2 = nullptr; // second upload stream of staged traces (zkm_trace_stage: alternate pieces, two copy engines)
    size_t ingest_chunk_cols = 32;      // columns per ingest chunk (0 = monolithic upload)           } zkm_ctx_set_tuning
    size_t keccak_parts_max_points = (size_t)1 << 15;   // k_quotient_keccak_parts up to this many points  }
    size_t fri_fused_division_min = ~(size_t)0;         // k_seg_scan_final from this many coefficients (default: never -- since the
                                                        // LDS-tiled kernels of round 4 the per-batch scans are faster at every size)  }
    int fri_scan_combine = 1;           // bottom level of the division by (X - z): scan and weighted sum in one launch (0: two launches)   } zkm_ctx_set_tuning
    size_t small_ntt = 1;               // transforms of 2^9 .. 2^13 points in one launch (k_ntt_small); 0: the two-pass plan          } zkm_ctx_set_tuning
    unsigned pow_round_log = 17;        // proof-of-work search: 2^this candidates per round of the search launch                          } zkm_ctx_set_tuning
    int aux_pipeline = 1;               // segments of short tables: lanes build later tables' auxiliary commitments behind the proofs      } zkm_ctx_set_tuning
    size_t commit_lanes = ZKM_COMMIT_LANES;   // trace / auxiliary commitments of one segment in flight (this context + lanes)   } zkm_ctx_set_tuning
    size_t segments_memory_budget = 0;  // bytes one wave of a zkm_prove_segments call may hold (0: 80 % of cached + free HBM, shared out over the contexts of the process that are inside such a call)   } zkm_ctx_set_tuning
    size_t last_stack = 0;              // segments of the previous prove_with_traces call of this context (0: none yet)
    size_t prev_stack = 0;              // ... and the other height whose blocks may still be cached (two heights are kept: ctl.hip prove_segments_impl)
    size_t max_stack = ZKM_MAX_SEG;     // segments of one lock-step group (zkm_prove_segments): 1 .. ZKM_MAX_SEG               } zkm_ctx_set_tuning
    int leaf_mfma = ZKM_LEAF_MFMA_DEFAULT;   // one-lane-per-leaf hashing: MDS layers of the full rounds on the matrix core (poseidon_mfma_dev.h)   } zkm_ctx_set_tuning
    size_t wide_max_hashes = 1024;      // launches of up to this many hashes use 16 lanes per hash (latency form)   } 0 / 0: one lane
    size_t quad_max_hashes = 32768;     // ... and up to this many four lanes per hash                                } per hash always
    int num_cus = 256;
    int cu_part_k = -1, cu_part_n = 0;   // measurement aid (ZKM_CU_MASK_PART): the streams of this context are confined to one part of the CUs
    // profiling
    bool profiling = false;
    std::vector<zkm_prof_rec> prof;
    std::vector<hipEvent_t> event_pool;
    struct agg { const char* name; uint64_t launches; double ms; };
    std::vector<agg> prof_agg;
    bool prof_agg_valid = false;
    // caching allocator: exact-size free lists
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live_blocks;
    std::mutex alloc_mu;                // guards the two maps (the out-of-memory path of a relative trims this cache from its thread)
    zkm_ctx* parent = nullptr;          // of a commit lane: the context that owns it
    std::atomic<int> debug_fail_allocs{0};   // test hook (root context only): pretend the next k hipMalloc first attempts fail
    // twiddles
    zkm_twiddles tw;
    // power tables for coset scaling: key (shift, log_n) -> device ptr [lo table 2^h | hi table 2^(log_n-h)]
    std::map<std::pair<uint64_t, unsigned>, gl_t*> pow_tables;
    // block twiddles of the coset-split LDE's upper stages: key (shift, log_n, stages) -> device ptr [4 cosets][2^stages]
    std::map<std::tuple<uint64_t, unsigned, unsigned>, gl_t*> lde_ct_tables;
    size_t resident_bytes = 0;  // of the live blocks: tables kept for reuse (twiddles, power tables, block twiddles)
    // pinned host staging
    uint64_t* h_staging = nullptr;
    size_t h_staging_words = 0;
    // commit lanes (ctl.hip zkm_prove_with_traces): sub-contexts -- own stream, allocator, twiddle / power tables, profiler records --
    // on which the independent trace commitments of one segment are built side by side, one host thread each.  Owned by this
    // context (created on first use, destroyed with it); a lane has no lanes of its own.  A batch remembers the (sub-)context
    // that built it and returns its memory there.
    std::vector<zkm_ctx*> lanes;
    void ensure_lanes(size_t k);

    void* alloc(size_t bytes);
    void release(void* p);
    void trim_self();  // hipFree every cached (not live) block of THIS allocator (any thread: used by a relative's out-of-memory retry)
    void trim();       // ... and of the lanes; only between calls
    void drop_copy_streams();   // trim(): the idle upload streams give their hardware queues back
    void shrink_down();   // the pinned download area back to its base size (trim(): between calls, owner's thread)
    void ensure_twiddles(unsigned log_n);
    const gl_t* pow_table(uint64_t shift, unsigned log_n);  // lo: 2^ceil(log_n/2) entries, then hi
    uint64_t* staging(size_t words);
    // Small transfers of the transcript round trips (caps, opening partials, FRI final polynomial, proof-of-work witness, query rounds
    // down; challenge powers, query indices, descriptors up) go through pinned memory: a copy from / to pageable memory is staged by the
    // runtime inside the call (a blit into its own pinned buffer, a host copy, and both of them behind the runtime's locks).
    struct xfer { void* dst; const void* src; size_t bytes; };
    void download(std::initializer_list<xfer> xs);               // all device -> host, then ONE stream synchronisation
    void download(void* dst, const void* src, size_t bytes) { download({xfer{dst, src, bytes}}); }
    void upload(void* dst, const void* src, size_t bytes);       // host -> device on the stream; `src` may be reused on return
    char* h_xfer = nullptr;                                      // [0, XFER_UP): upload ring,
                                                                 // then one cache line for the completion flag of k_download
    uint64_t down_seq = 0;
    void wait_flag(const uint64_t* flag, uint64_t seq);          // spin, then (crowded process) block: core.hip
    uint64_t block_after_us = 50;                                // } zkm_ctx_set_tuning "block_after_us" (0: always block)
    hipEvent_t block_event = nullptr;
    uint64_t blocked_waits = 0;                                  // round trips that ended in the blocking wait (diagnostic)
    void ensure_xfer();
    // a kernel that delivers a small result to the host ITSELF (hash.hip k_merkle_tail): xfer_begin hands out the pinned slot, the flag,
    // a zeroed device word for "last workgroup publishes" and the sequence number to publish; xfer_finish waits for it and copies out
    uint64_t xfer_begin(size_t bytes, uint64_t** host_slot, uint64_t** flag, unsigned** counter);
    void xfer_finish(uint64_t seq, void* dst, size_t bytes);
    unsigned* d_counter = nullptr;
    unsigned long long* d_pow_best = nullptr;                    // the proof-of-work search's result word: all ones between searches
    unsigned long long* pow_best();
    size_t tree_tail = 1;               // trees of <= 2^15 leaves in one launch incl. the cap's trip to the host; 0: levels + download      } zkm_ctx_set_tuning
    size_t up_off = 0;
    char* h_down = nullptr;                                      // pinned download area, down_cap bytes (grows: ensure_down)
    size_t down_cap = 0;
    void ensure_down(size_t bytes);
    static constexpr size_t XFER_DOWN = (size_t)1 << 20, XFER_DOWN_MAX = (size_t)1 << 28, XFER_UP = (size_t)1 << 18;
    hipEvent_t get_event();
    size_t prof_begin(const char* name);   // returns the record's index (scopes nest: a stage scope holds kernel scopes)
    void prof_end(size_t idx);
    void sync() { ZKM_HIP_CHECK(hipStreamSynchronize(stream)); up_off = 0; }
};


// RAII owner of one scratch block from the context's allocator (released on every exit path)
struct zkm_scratch {
    zkm_ctx* c;
    void* p;
    zkm_scratch(zkm_ctx* ctx, size_t bytes) : c(ctx), p(ctx->alloc(bytes)) {}
    zkm_scratch(const zkm_scratch&) = delete;
    zkm_scratch& operator=(const zkm_scratch&) = delete;
    ~zkm_scratch() { c->release(p); }
    template <class T> T* as() const { return (T*)p; }
};

// RAII profiling scope around one kernel launch (or a small group).  Names starting with "stage/" are the reference's timed!
// scopes (prover.rs:146-153, 193, 204, 479, 513, 545, 578, 620): event pairs at the stage boundaries, reported beside the kernel
// records (consumers separate the two by the prefix; a stage's time includes its transcript round trips and launch gaps).
struct zkm_prof_scope {
    zkm_ctx* c;
    size_t idx = 0;
    bool on;
    zkm_prof_scope(zkm_ctx* ctx, const char* name) : c(ctx), on(ctx->profiling) {
        if (on) idx = c->prof_begin(name);
    }
    zkm_prof_scope(const zkm_prof_scope&) = delete;
    zkm_prof_scope& operator=(const zkm_prof_scope&) = delete;
    ~zkm_prof_scope() {
        if (on) c->prof_end(idx);
    }
};

// A batch may hold the SAME-SHAPED polynomials of `nseg` independent segments (zkm_prove_segments: K segments advance through every
// stage of prove_with_traces in lock-step, one launch per stage): segment s owns columns [s ncols, (s + 1) ncols) of the coefficient and
// LDE matrices (column strides n and N as ever, so the transforms see nseg * ncols columns), its own block of dig_words digest words
// (level_off is relative to the block) and cap words [s capw, (s + 1) capw).  Kernels take the segment from blockIdx.z.
struct zkm_batch {
    zkm_ctx* ctx = nullptr;
    size_t ncols = 0;         // per segment
    size_t nseg = 1;
    unsigned log_n = 0, rate_bits = 0, cap_height = 0;
    gl_t* coeffs = nullptr;   // nseg x ncols x n; position P of a column = coefficient of X^zkm_coeff_exponent(P, coeff_s1)
    unsigned coeff_s1 = 0;    // 0: natural order; else the digit layout of the two-pass inverse transform (2^18 .. 2^20 rows)
    gl_t* lde = nullptr;      // nseg x ncols x N, rows bit-reversed
    gl_t* digests = nullptr;  // per segment: levels 0..top concatenated, 4 words per node
    size_t dig_words = 0;     // digest words of one segment's tree
    std::vector<size_t> level_off;  // word offsets (within a segment's block)
    std::vector<uint64_t> cap;      // host copy, nseg x (4 << cap_height) words
    size_t coeff_seg() const { return ncols << log_n; }
    size_t lde_seg() const { return ncols << (log_n + rate_bits); }
    size_t n() const { return (size_t)1 << log_n; }
    size_t N() const { return (size_t)1 << (log_n + rate_bits); }
    unsigned lde_bits() const { return log_n + rate_bits; }
    unsigned top() const { return lde_bits() - cap_height; }
};

inline bool zkm_is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

// ---- hash.hip
void zkm_launch_poseidon_permute(zkm_ctx*, gl_t* states, size_t k);
void zkm_launch_mul_selftest_branchfree(zkm_ctx*, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);  // stark.hip
void zkm_launch_keccakf(zkm_ctx*, uint64_t* states, size_t k);
// leaf digests of a column-major matrix (row j across ncols columns of stride `col_stride` words)
// nseg > 1: segment s reads lde + s * lde_seg and writes digests + s * dig_seg (blockIdx.z)
void zkm_launch_merkle_leaves(zkm_ctx*, const gl_t* lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* digests, size_t nseg = 1,
                              size_t lde_seg = 0, size_t dig_seg = 0);
// the same digests from column chunks: state = 12 x nrows words kept between the chunks of one matrix (see hash.hip)
void zkm_launch_merkle_leaves_chunk(zkm_ctx*, const gl_t* lde, size_t nrows, size_t nc, size_t col_stride, gl_t* state, bool first,
                                    bool last, gl_t* digests);
void zkm_ntt_natural_ex(zkm_ctx* c, const gl_t* in, size_t cs_in, gl_t* scratch, size_t cs_s, gl_t* out, size_t cs_out, size_t ncols,
                        unsigned log_n, bool inverse, uint64_t shift);
// leaf digests of row-major leaves formed from F2 SoA arrays: leaf k = 16 consecutive (c0,c1) pairs
// (nseg > 1: segment s reads c0 / c1 + s * val_seg and writes digests + s * dig_seg)
void zkm_launch_merkle_leaves_ext(zkm_ctx*, const gl_t* c0, const gl_t* c1, size_t nleaves, unsigned arity, gl_t* digests, size_t nseg = 1,
                                  size_t val_seg = 0, size_t dig_seg = 0);
void zkm_launch_poseidon_trace(zkm_ctx*, uint64_t seed, const uint64_t* d_inputs, const uint64_t* d_ts, size_t num_perms, unsigned log_n,
                               gl_t* out);
void zkm_launch_poseidon_sponge_trace(zkm_ctx*, const uint8_t* d_inputs, const uint64_t* d_off, const uint64_t* d_meta,
                                      const uint64_t* d_row_off, size_t nops, unsigned log_n, gl_t* out);
void zkm_launch_keccak_sponge_trace(zkm_ctx*, const uint8_t* d_inputs, const uint64_t* d_off, const uint64_t* d_meta,
                                    const uint64_t* d_row_off, size_t nops, size_t rows_used, unsigned log_n, gl_t* out);
// build all digest layers above level 0; fills level_off and returns total words needed (call with digests==nullptr to size)
size_t zkm_merkle_layout(unsigned log_leaves, unsigned cap_height, std::vector<size_t>& level_off);
// (nseg trees of the same shape, dig_seg words apart: one launch per group of levels for all of them)
void zkm_merkle_build_inner(zkm_ctx*, gl_t* digests, const std::vector<size_t>& level_off, unsigned log_leaves, unsigned cap_height,
                            size_t nseg = 1, size_t dig_seg = 0);
// ... and the caps into cap_out (host, nseg x (4 << cap_height) words): one launch for small trees (zkm_merkle_tail), levels + download otherwise
void zkm_merkle_build_inner_cap(zkm_ctx*, gl_t* digests, const std::vector<size_t>& level_off, unsigned log_leaves, unsigned cap_height,
                                uint64_t* cap_out, size_t nseg = 1, size_t dig_seg = 0);
bool zkm_merkle_tail(zkm_ctx*, gl_t* digests, const std::vector<size_t>& level_off, unsigned log_leaves, unsigned cap_height, uint64_t* cap_out,
                     unsigned l0, size_t nseg = 1, size_t dig_seg = 0);

// ---- ntt.hip
// in-place, natural -> bit-reversed order, forward or inverse roots, no scaling
void zkm_ntt_dif_bitrev(zkm_ctx*, gl_t* data, size_t ncols, size_t col_stride, unsigned log_n, bool inverse);
// natural -> natural into `out` (may not alias `in`), using `in` as scratch (destroyed).
//   inverse: out = iNTT(in) * shift^-i ; forward: out = NTT(in * shift^i)   (shift 0/1 = none)
void zkm_ntt_natural(zkm_ctx*, gl_t* in_scratch, gl_t* out, size_t ncols, size_t col_stride_in, size_t col_stride_out,
                     unsigned log_n, bool inverse, uint64_t shift);
// out[c][i] = in[c][i] * shift^i for i < n_in, 0 for n_in <= i < n_out  (columns strided)
void zkm_launch_scale_pad(zkm_ctx*, const gl_t* in, size_t col_stride_in, gl_t* out, size_t col_stride_out, size_t ncols,
                          unsigned log_n_in, unsigned log_n_out, uint64_t shift);
// coset LDE of coefficients (natural order, or the digit layout when coeff_s1 != 0) into bit-reversed evaluations: out (ncols x 2^(log_n+rate_bits))
void zkm_lde_bitrev(zkm_ctx*, const gl_t* coeffs, gl_t* out, size_t ncols, unsigned log_n, unsigned rate_bits, uint64_t shift,
                    unsigned coeff_s1 = 0);
// ---- coefficient layout of a batch (ntt.hip zkm_intt_digit): position P of a column holds the coefficient of X^zkm_coeff_exponent(P).
// s1 = 0: natural order.  s1 = log_n - 12 for 2^18 .. 2^20 rows: what the two-pass inverse transform leaves (see ntt.hip).
inline unsigned zkm_coeff_layout_s1(unsigned log_n) { return (log_n >= 18 && log_n <= 20) ? log_n - 12 : 0; }
GL_HD uint32_t zkm_coeff_exponent(uint32_t P, unsigned s1) {
    if (!s1) return P;
    const uint32_t t_hi = P & ((1u << s1) - 1), cc = (P >> s1) & ((1u << (12 - s1)) - 1), blk = P >> 12;
    return (t_hi << 12) | (cc << s1) | bitrev32(blk, s1);
}
// values (natural, read-only) -> coefficients in the digit layout, two passes; coefficient columns to / from natural order (in != out)
void zkm_intt_digit(zkm_ctx*, const gl_t* values, size_t cs_in, gl_t* coeffs, size_t cs_out, size_t ncols, unsigned log_n);
void zkm_coeff_layout_convert(zkm_ctx*, const gl_t* in, size_t cs_in, gl_t* out, size_t cs_out, size_t ncols, unsigned log_n, bool to_natural);

// ---- core.hip
int zkm_live_contexts();   // contexts of this process that exist right now (zkm_ctx_create .. zkm_ctx_destroy; lanes not counted)
// dev_values (optional, ncols x n words of device memory): host values are uploaded THERE and stay (the caller reuses them, e.g. for
// the CTL columns of prove_with_traces) instead of being staged inside the batch's LDE buffer.
// src_cols (optional, instead of src): one pointer per column (each n words, host or device).
void zkm_batch_build(zkm_batch* b, const uint64_t* src, bool src_is_values, gl_t* dev_values = nullptr,
                     const uint64_t* const* src_cols = nullptr, const uint64_t* const* seg_srcs = nullptr);
zkm_batch* zkm_batch_commit_values_keep(zkm_ctx* c, const uint64_t* values, size_t ncols, unsigned log_n, unsigned rate_bits,
                                        unsigned cap_height, gl_t* dev_values, const uint64_t* const* columns = nullptr);
// stark.hip: prove_single_table on an existing trace and auxiliary commitment (throws)
void zkm_prove_single_table_aux(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, size_t ncols, unsigned log_n, const zkm_batch* trace_batch,
                                const zkm_batch* aux_batch, size_t naux_ctl, const zkm_ctl_table* table, const zkm_ctl_z* zs,
                                const uint32_t* colset_ids, size_t nzs, const uint64_t* lookup_challenges, const std::vector<zkm_challenger*>& chs,
                                const std::vector<uint64_t*>& proofs);
void zkm_launch_canon(zkm_ctx* c, gl_t* v, size_t total);   // v[i] = canonical representative of v[i], in place
void zkm_host_poseidon_permute(uint64_t st[12]);
void zkm_host_poseidon_permute_reference(uint64_t st[12]);   // poseidon_dev.h compiled for the host (cross-check)
// ---- hash.hip (LogicStark witness)
// ---- tables' own logUp lookups (core.hip: definitions; ctl.hip: helper columns)
struct zkm_table_lookup { uint32_t ncols; const uint32_t* cols; uint32_t table_col, freq_col; };
const zkm_table_lookup* zkm_table_lookups(int table_id, size_t* n);
void zkm_table_lookup_columns_device(zkm_ctx* c, int table_id, const uint64_t* challenges, size_t nch, const gl_t* d_trace, size_t n,
                                     gl_t* d_out, size_t nseg = 1, size_t trace_seg = 0, size_t out_seg = 0);
void zkm_launch_sha_extend_trace(zkm_ctx* c, const uint8_t* d_inputs, const uint64_t* d_ts, size_t k, size_t n, gl_t* out);
void zkm_launch_sha_extend_sponge_trace(zkm_ctx* c, const uint32_t* d_w16, const uint64_t* d_meta, size_t k, size_t n, gl_t* out);
void zkm_launch_sha_compress_trace(zkm_ctx* c, bool sponge, const uint32_t* d_hx, const uint32_t* d_w, const uint64_t* d_meta, size_t k,
                                   size_t n, gl_t* out);
void zkm_launch_keccak_trace(zkm_ctx* c, const uint64_t* d_inputs, const uint64_t* d_ts, size_t nperms, size_t n, gl_t* out);
void zkm_launch_logic_trace(zkm_ctx* c, const uint32_t* d_ops, size_t nops, size_t n, gl_t* out, int* d_bad);

// ctl.hip: the body of zkm_prove_segments[_columns] (exactly one of traces / columns non-null); seg_base = position of segment 0 in the
// caller's larger call (csrc/pool.hip deals groups of one pool call to its workers) -- used in error messages only
extern "C" int zkm_prove_segments_entry(const char* what, zkm_ctx* c, const zkm_stark_config* cfg, size_t nseg, const uint64_t* const* const* traces,
                                        const uint64_t* const* const* const* columns, const unsigned* const* log_n, const uint64_t* const* pub,
                                        const size_t* npub, uint64_t* const* proofs, uint64_t* const* challenges, char** err, size_t seg_base);
