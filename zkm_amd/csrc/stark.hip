// stark.hip -- prove_single_table on the GPU: quotient (K7-K9), openings (K10), FRI (K11-K14), queries.
//
// Host orchestration restates /root/reference/prover/src/prover.rs:441-641 (stage and transcript order);
// the kernels replace the rayon loops of compute_quotient_polys (:700-782), StarkOpeningSet::new
// (proof.rs:299-334) and plonky2's PolynomialBatch::prove_openings / fri_proof.
// The Challenger stays on the host (a few hundred field elements per table); only caps, openings,
// the final polynomial and the query openings cross PCIe.
#define GL_REDUCE_BRANCHFREE 1   // (gl_dev.h: these kernels interleave independent products at low occupancy)
#include <memory>

#include "constraints_dev.h"
#include "ctl_dev.h"

// ------------------------------------------------------------------ proof layout (include/zkm_hip.h)
struct proof_layout {
    unsigned log_n, lde_bits, L, cap;
    size_t W, A, Q, Z, F, C, nq;
    size_t o_init, o_caps, o_open, o_fri_caps, o_final, o_pow, o_queries, query_words, total;
};

// FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) (config.rs:25)
static unsigned fri_num_layers(const zkm_stark_config* c, unsigned degree_bits) {
    unsigned l = 0, d = degree_bits;
    // (d >= arity_bits keeps the subtraction from wrapping when final_poly_bits < arity_bits)
    while (d > c->final_poly_bits && d >= c->arity_bits && d + c->rate_bits - c->arity_bits >= c->cap_height) { d -= c->arity_bits; l++; }
    return l;
}

// Every field of the caller's config is checked once, here, before any allocation or transcript mutation (make_layout is the first
// thing every prove entry point and zkm_proof_words call).  The quotient kernels carry at most two alpha accumulators
// (StarkConfig::num_challenges = 2 in standard_fast_config, config.rs:17-30).
static void validate_config(const zkm_stark_config* c, unsigned log_n) {
    if (!c) throw std::runtime_error("stark config: null");
    if (log_n == 0 || log_n > 32) throw std::runtime_error("stark config: degree_bits out of range");
    if (c->rate_bits != 2) throw std::runtime_error("stark config: only rate_bits = 2 is supported");
    if (c->arity_bits < 2 || c->arity_bits > 6) throw std::runtime_error("stark config: arity_bits must be in 2..6");
    if (c->pow_bits == 0 || c->pow_bits > 32) throw std::runtime_error("stark config: proof_of_work_bits must be in 1..32");
    if (c->num_challenges < 1 || c->num_challenges > 2) throw std::runtime_error("stark config: num_challenges must be 1 or 2");
    if (c->num_queries == 0 || c->num_queries > 4096) throw std::runtime_error("stark config: num_query_rounds must be in 1..4096");
    if (c->cap_height > log_n + c->rate_bits) throw std::runtime_error("stark config: cap_height exceeds the height of the Merkle tree");
    if (c->final_poly_bits > 32) throw std::runtime_error("stark config: final_poly_bits out of range");
}

static void make_layout(proof_layout& y, const zkm_stark_config* c, unsigned log_n, size_t W, size_t A, size_t Z) {
    validate_config(c, log_n);
    y.log_n = log_n; y.lde_bits = log_n + c->rate_bits; y.cap = c->cap_height;
    y.W = W; y.A = A; y.Q = (size_t)c->num_challenges * 2; y.Z = Z;
    y.L = fri_num_layers(c, log_n);
    y.F = (size_t)1 << (log_n - y.L * c->arity_bits);
    y.C = (size_t)1 << c->cap_height;
    y.nq = c->num_queries;
    size_t o = 16;
    y.o_init = o; o += 12;
    y.o_caps = o; o += 3 * y.C * 4;
    y.o_open = o; o += 4 * W + 4 * A + Z + 2 * y.Q;
    y.o_fri_caps = o; o += y.L * y.C * 4;
    y.o_final = o; o += 2 * y.F;
    y.o_pow = o; o += 1;
    y.o_queries = o;
    size_t sib0 = (size_t)(y.lde_bits - y.cap) * 4;
    size_t q = (W + sib0) + (A + sib0) + (y.Q + sib0);
    for (unsigned i = 0; i < y.L; i++)
        q += 2 * ((size_t)1 << c->arity_bits) + (size_t)(y.lde_bits - c->arity_bits * (i + 1) - y.cap) * 4;
    y.query_words = q;
    y.total = o + q * y.nq;
}

// ------------------------------------------------------------------ K7: quotient evaluation
// Table constraints live in constraints_dev.h; constraint order = alpha-power order (constraint_consumer.rs:57-62): table
// constraints, then the table's lookups, then the CTL checks (vanishing_poly.rs:17-46).
// CTL checks driven by the column-set description (eval_helper_columns cross_table_lookup.rs:1006-1058,
// eval_cross_table_lookup_checks :1067-1150).  The benchmark's fake CTL data (helper columns, no column sets)
// is the ncolsets == 0 case: only the last-row / transition checks on Z are emitted.
// Constraints of CtlZData `i`: helper-column checks q0 <= q < q1, then (if `closing`) the last-row / transition checks on Z.
// `start` = index of the z's first helper column among the auxiliary columns.
template <int NA>
__device__ void eval_ctl_z(const ctl_dev& d, uint32_t i, uint32_t start, uint32_t q0, uint32_t q1, bool closing, const gl_t* __restrict__ tl,
                           size_t N, ptrdiff_t dnext, const gl_t* __restrict__ aux, size_t j, size_t jn, consumer_t<NA>& k) {
    const zkm_ctl_z z = d.zs[i];
    const uint32_t* ids = d.colset_ids + z.colset_off;
    if (z.num_helpers) {
        for (uint32_t q = q0; q < q1; q++) {
            gl_t h = aux[(size_t)(start + q) * N + j];
            const zkm_colset c0 = d.colsets[ids[2 * q]];
            gl_t combin0 = ctl_combine(d, c0, z.beta, z.gamma, tl, N, dnext, true), f0 = ctl_eval_filter(d, c0, tl, N, dnext, true);
            if (2 * q + 1 < z.ncolsets) {
                const zkm_colset c1 = d.colsets[ids[2 * q + 1]];
                gl_t combin1 = ctl_combine(d, c1, z.beta, z.gamma, tl, N, dnext, true), f1 = ctl_eval_filter(d, c1, tl, N, dnext, true);
                k.constraint(gl_sub(gl_sub(gl_mul(gl_mul(combin1, combin0), h), gl_mul(f0, combin1)), gl_mul(f1, combin0)));
            } else {
                k.constraint(gl_sub(gl_mul(combin0, h), f0));
            }
        }
    }
    if (!closing) return;
    gl_t local_z = aux[(size_t)(d.total_helpers + i) * N + j], next_z = aux[(size_t)(d.total_helpers + i) * N + jn];
    if (z.num_helpers) {
        gl_t h_sum = 0;
        for (uint32_t q = 0; q < z.num_helpers; q++) h_sum = gl_add(h_sum, aux[(size_t)(start + q) * N + j]);
        k.last_row(gl_sub(local_z, h_sum));
        k.transition(gl_sub(gl_sub(local_z, next_z), h_sum));
    } else if (z.ncolsets > 1) {
        const zkm_colset c0 = d.colsets[ids[0]], c1 = d.colsets[ids[1]];
        gl_t combin0 = ctl_combine(d, c0, z.beta, z.gamma, tl, N, dnext, true), combin1 = ctl_combine(d, c1, z.beta, z.gamma, tl, N, dnext, true);
        gl_t f0 = ctl_eval_filter(d, c0, tl, N, dnext, true), f1 = ctl_eval_filter(d, c1, tl, N, dnext, true);
        gl_t cc = gl_mul(combin0, combin1), rhs = gl_add(gl_mul(f0, combin1), gl_mul(f1, combin0));
        k.last_row(gl_sub(gl_mul(cc, local_z), rhs));
        k.transition(gl_sub(gl_mul(cc, gl_sub(local_z, next_z)), rhs));
    } else {
        const zkm_colset c0 = d.colsets[ids[0]];
        gl_t combin0 = ctl_combine(d, c0, z.beta, z.gamma, tl, N, dnext, true), f0 = ctl_eval_filter(d, c0, tl, N, dnext, true);
        k.last_row(gl_sub(gl_mul(combin0, local_z), f0));
        k.transition(gl_sub(gl_mul(combin0, gl_sub(local_z, next_z)), f0));
    }
}
template <int NA>
__device__ void eval_ctl_constraints(const ctl_dev& d, const gl_t* __restrict__ tl, size_t N, ptrdiff_t dnext,
                                     const gl_t* __restrict__ aux, size_t j, size_t jn, consumer_t<NA>& k) {
    uint32_t start = 0;
    for (uint32_t i = 0; i < d.nzs; i++) {
        const zkm_ctl_z z = d.zs[i];
        eval_ctl_z<NA>(d, i, start, 0, z.num_helpers ? (z.ncolsets + 1) / 2 : 0, true, tl, N, dnext, aux, j, jn, k);
        start += z.num_helpers;
    }
}

// One thread per point of the quotient domain g<w_2n>, visited in LDE storage order: storage row j < 2n
// of the 4n-row LDE is natural quotient index i = bitrev_{L-1}(j) (every `step` = 2nd natural LDE row,
// prover.rs:668-675); "next" is natural +2 in the 2n domain (prover.rs:704) = +4 in the 4n domain.
// Per-point setup shared by the quotient kernels.
struct quotient_point {
    size_t j, jn;
    uint32_t i;
};
// the (<= 2) constraint challenges of every segment of a stack travel in the kernel-argument segment (seg_gl2: v[2 s + a]): no device
// copy, no upload per table.  Every quotient kernel takes its segment from blockIdx.z: the segment's columns start W N (trace) / A N
// (auxiliary) words after the previous segment's, its outputs NA * size words.
typedef seg_gl2 alpha_args;
// check mode (zkm_check_constraints == check_constraints, prover.rs:793-910, rate_bits = 0): the "LDE" is the trace itself, N = n rows in
// natural order, point j = w_n^j (wpow: the table of w_n), next row j + 1 mod n, and the Lagrange selectors are the indicator values
// of rows 0 and n - 1 -- signalled by gn == 0 (g^n is never 0 on the coset).
template <int NA>
__device__ __forceinline__ quotient_point quotient_setup(size_t j, unsigned lde_bits, const gl_t* __restrict__ alphas, const gl_t* __restrict__ wpow,
                                                         gl_t gn, gl_t last, gl_t w_n, gl_t n_inv, consumer_t<NA>& k) {
    const size_t N = (size_t)1 << lde_bits;
    if (gn == 0) {
        const unsigned h = (lde_bits + 1) / 2;
        const gl_t x = gl_mul_loose(wpow[j & ((1u << h) - 1)], wpow[((size_t)1 << h) + (j >> h)]);
#pragma unroll
        for (int a = 0; a < NA; a++) { k.alpha[a] = alphas[a]; k.acc[a] = 0; }
        k.z_last = gl_sub(x, last);
        k.l_first = j == 0 ? 1 : 0;
        k.l_last = j == N - 1 ? 1 : 0;
        return quotient_point{j, (j + 1) & (N - 1), (uint32_t)j};
    }
    uint32_t t = bitrev32((uint32_t)j, lde_bits);  // natural index in the 4n domain (even)
    uint32_t i = t >> 1;                           // natural index in the 2n quotient domain
    uint32_t tn = (t + 4) & (uint32_t)(N - 1);
    // x = g * w_{4n}^t
    unsigned h = (lde_bits + 1) / 2;
    gl_t x = gl_mul(GL_GENERATOR, gl_mul_loose(wpow[t & ((1u << h) - 1)], wpow[((size_t)1 << h) + (t >> h)]));
#pragma unroll
    for (int a = 0; a < NA; a++) { k.alpha[a] = alphas[a]; k.acc[a] = 0; }
    k.z_last = gl_sub(x, last);
    // Z_H(x) = x^n - 1 = g^n (-1)^i - 1;  L_first = Z_H / (n (x - 1)),  L_last = Z_H / (n (w x - 1))
    gl_t zh = gl_sub((i & 1) ? gl_neg(gn) : gn, 1);
    gl_t d0 = gl_sub(x, 1), d1 = gl_sub(gl_mul(w_n, x), 1);
    gl_t dinv = gl_inv(gl_mul(d0, d1));
    gl_t zn = gl_mul(zh, n_inv);
    k.l_first = gl_mul(zn, gl_mul(dinv, d1));
    k.l_last = gl_mul(zn, gl_mul(dinv, d0));
    return quotient_point{j, (size_t)bitrev32(tn, lde_bits), i};
}

// The vanishing polynomial is accumulated by separate launches (vanishing_poly.rs:30-45 order: table constraints, the table's
// lookups, then the cross-table lookup checks): k_quotient<TABLE> stores the running accumulators after the table's own
// constraints; the table-independent CTL kernels continue the same Horner recurrence acc = acc * alpha + c through the CTL checks
// and multiply by Z_H^-1.  The descriptor interpreter stays out of the per-table kernels' register budgets, and a table with few
// rows and hundreds of looking column sets (KeccakSponge) can spread its CTL checks over many workgroups:
//   k_quotient_ctl          all CTL constraints of a point in one thread (large tables: already enough parallelism)
//   k_quotient_ctl_chunk    blockIdx.y = a chunk of consecutive constraints [k_begin, k_end); Horner from 0 within the chunk,
//                           scaled by alpha^(K - k_end) (K = number of CTL constraints) into tmp[chunk]
//   k_quotient_ctl_sum      acc * alpha^K + sum of the chunks, times Z_H^-1
// -- the same polynomial in alpha, so the values are identical.
template <int TABLE, int NA>
__global__ __launch_bounds__(256, (TABLE == ZKM_TABLE_CPU || TABLE == ZKM_TABLE_POSEIDON || TABLE == ZKM_TABLE_ARITHMETIC ? 4 : 1)) void k_quotient(const gl_t* __restrict__ trace, const gl_t* __restrict__ aux,
                                                           unsigned log_n, unsigned lde_bits, lookup_dev lookups, alpha_args alphas_v,
                                                           seg_gl2 lookup_ch,
                                                           const gl_t* __restrict__ wpow /* w_{4n}^t two-level table */,
                                                           gl_t gn, gl_t last, gl_t w_n, gl_t n_inv, gl_t* __restrict__ out, size_t trace_seg,
                                                           size_t aux_seg) {
    const gl_t* const alphas = alphas_v.v + 2 * blockIdx.z;
    size_t N = (size_t)1 << lde_bits;
    size_t size = gn == 0 ? N : N >> 1;       // (check mode: every row of the trace)
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= size) return;
    trace += (size_t)blockIdx.z * trace_seg;
    aux += (size_t)blockIdx.z * aux_seg;
    out += (size_t)blockIdx.z * NA * size;
    consumer_t<NA> k;
    const quotient_point q = quotient_setup<NA>(j, lde_bits, alphas, wpow, gn, last, w_n, n_inv, k);
    eval_table_constraints<TABLE, NA>(trace + j, N, (ptrdiff_t)q.jn - (ptrdiff_t)j, k);
    // auxiliary columns: the table's lookup helper columns first, then the CTL helper columns and Zs (prover.rs:495-508)
    if constexpr (TABLE == ZKM_TABLE_MEMORY || TABLE == ZKM_TABLE_ARITHMETIC)
        eval_lookup_constraints<NA>(lookups, lookup_ch.v + 2 * blockIdx.z, trace + j, N, aux, j, q.jn, k);
#pragma unroll
    for (int a = 0; a < NA; a++) out[(size_t)a * size + q.i] = k.acc[a];
}

// Short Keccak tables: KECCAK_CONSTRAINT_PARTS threads per point (constraints_dev.h, eval_keccak_constraints_part); blockIdx.y = part.
template <int NA>
__global__ __launch_bounds__(256) void k_quotient_keccak_parts(const gl_t* __restrict__ trace, unsigned lde_bits, alpha_args alphas_v,
                                                               const gl_t* __restrict__ wpow, gl_t gn, gl_t last, gl_t w_n, gl_t n_inv,
                                                               const gl_t* __restrict__ apw /* [segment][NA][K + 1] */,
                                                               gl_t* __restrict__ tmp /* [segment][part][NA][size] */, size_t trace_seg) {
    const gl_t* const alphas = alphas_v.v + 2 * blockIdx.z;
    size_t N = (size_t)1 << lde_bits;
    size_t size = N >> 1;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= size) return;
    trace += (size_t)blockIdx.z * trace_seg;
    apw += (size_t)blockIdx.z * NA * (KECCAK_NUM_CONSTRAINTS + 1);
    tmp += (size_t)blockIdx.z * gridDim.y * NA * size;
    consumer_t<NA> k;
    const quotient_point q = quotient_setup<NA>(j, lde_bits, alphas, wpow, gn, last, w_n, n_inv, k);
    eval_keccak_constraints_part<NA>(trace + j, N, (ptrdiff_t)q.jn - (ptrdiff_t)j, k, (int)blockIdx.y, apw);
#pragma unroll
    for (int a = 0; a < NA; a++) tmp[((size_t)blockIdx.y * NA + a) * size + q.i] = k.acc[a];
}
__global__ __launch_bounds__(256) void k_sum_parts(const gl_t* __restrict__ tmp, unsigned nparts, size_t words, gl_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    tmp += (size_t)blockIdx.z * nparts * words;
    out += (size_t)blockIdx.z * words;
    gl_t acc = 0;
    for (unsigned p = 0; p < nparts; p++) acc = gl_add(acc, tmp[(size_t)p * words + i]);
    out[i] = acc;
}

template <int NA>
__global__ __launch_bounds__(256) void k_quotient_ctl(const gl_t* __restrict__ trace, const gl_t* __restrict__ aux, unsigned lde_bits, ctl_dev ctl,
                                                      uint32_t num_lookup_cols, alpha_args alphas_v, const gl_t* __restrict__ wpow, gl_t gn,
                                                      gl_t zh_inv0, gl_t zh_inv1, gl_t last, gl_t w_n, gl_t n_inv, gl_t* __restrict__ out,
                                                      size_t trace_seg, size_t aux_seg) {
    const gl_t* const alphas = alphas_v.v + 2 * blockIdx.z;
    size_t N = (size_t)1 << lde_bits;
    size_t size = gn == 0 ? N : N >> 1;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= size) return;
    trace += (size_t)blockIdx.z * trace_seg;
    aux += (size_t)blockIdx.z * aux_seg;
    out += (size_t)blockIdx.z * NA * size;
    ctl.zs += (size_t)blockIdx.z * ctl.nzs;
    consumer_t<NA> k;
    const quotient_point q = quotient_setup<NA>(j, lde_bits, alphas, wpow, gn, last, w_n, n_inv, k);
#pragma unroll
    for (int a = 0; a < NA; a++) k.acc[a] = out[(size_t)a * size + q.i];
    eval_ctl_constraints<NA>(ctl, trace + j, N, (ptrdiff_t)q.jn - (ptrdiff_t)j, aux + (size_t)num_lookup_cols * N, j, q.jn, k);
    gl_t zi = (q.i & 1) ? zh_inv1 : zh_inv0;
#pragma unroll
    for (int a = 0; a < NA; a++) out[(size_t)a * size + q.i] = gl_mul(k.acc[a], zi);
}

struct ctl_chunk {
    uint32_t z, helper_start, q0, q1, closing, tail;  // tail = K - k_end: constraints that follow the chunk
};
template <int NA>
__global__ __launch_bounds__(256) void k_quotient_ctl_chunk(const gl_t* __restrict__ trace, const gl_t* __restrict__ aux, unsigned lde_bits, ctl_dev ctl,
                                                            const ctl_chunk* __restrict__ chunks, uint32_t num_lookup_cols, alpha_args alphas_v,
                                                            const gl_t* __restrict__ wpow, gl_t gn, gl_t last, gl_t w_n, gl_t n_inv,
                                                            gl_t* __restrict__ tmp /* [segment][chunk][NA][size] */, size_t trace_seg, size_t aux_seg) {
    const gl_t* const alphas = alphas_v.v + 2 * blockIdx.z;
    size_t N = (size_t)1 << lde_bits;
    size_t size = N >> 1;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= size) return;
    trace += (size_t)blockIdx.z * trace_seg;
    aux += (size_t)blockIdx.z * aux_seg;
    tmp += (size_t)blockIdx.z * gridDim.y * NA * size;
    ctl.zs += (size_t)blockIdx.z * ctl.nzs;
    const ctl_chunk ch = chunks[blockIdx.y];
    consumer_t<NA> k;
    const quotient_point q = quotient_setup<NA>(j, lde_bits, alphas, wpow, gn, last, w_n, n_inv, k);
    eval_ctl_z<NA>(ctl, ch.z, ch.helper_start, ch.q0, ch.q1, ch.closing != 0, trace + j, N, (ptrdiff_t)q.jn - (ptrdiff_t)j,
                   aux + (size_t)num_lookup_cols * N, j, q.jn, k);
#pragma unroll
    for (int a = 0; a < NA; a++)
        tmp[((size_t)blockIdx.y * NA + a) * size + q.i] = gl_mul(k.acc[a], gl_pow(k.alpha[a], ch.tail));
}
template <int NA>
__global__ __launch_bounds__(256) void k_quotient_ctl_sum(const gl_t* __restrict__ tmp, uint32_t nchunks, uint32_t nconstraints, alpha_args alphas_v,
                                                          gl_t zh_inv0, gl_t zh_inv1, size_t size, gl_t* __restrict__ out) {
    const gl_t* const alphas = alphas_v.v + 2 * blockIdx.z;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    tmp += (size_t)blockIdx.z * nchunks * NA * size;
    out += (size_t)blockIdx.z * NA * size;
    gl_t zi = (i & 1) ? zh_inv1 : zh_inv0;
#pragma unroll
    for (int a = 0; a < NA; a++) {
        gl_t acc = gl_mul(out[(size_t)a * size + i], gl_pow(alphas[a], nconstraints));
        for (uint32_t c = 0; c < nchunks; c++) acc = gl_add(acc, tmp[((size_t)c * NA + a) * size + i]);
        out[(size_t)a * size + i] = gl_mul(acc, zi);
    }
}

// quotient polys: d_out = nalphas x 2n natural-order coefficients (device) -- per segment of the stack (trace / aux: stacked batches of
// the same nseg; alphas_host: nseg x nalphas; lookup_challenges: nseg x nalphas; own: a description with nseg lists of CtlZData)
// check = true (zkm_check_constraints): trace / aux describe the VALUES (lde = ncols x n values, rate_bits 0) and d_out receives the
// alpha-accumulated constraint values of every row, nalphas x n, instead of quotient coefficients.
static void quotient_device(zkm_ctx* c, int table_id, const zkm_batch* trace, const zkm_batch* aux, const ctl_dev_owner& own,
                            const uint64_t* lookup_challenges, const gl_t* alphas_host, size_t nalphas, gl_t* d_out, bool check = false) {
    lookup_dev lookups{};
    seg_gl2 lookup_ch{};
    uint32_t NL = 0;
    const size_t nseg = trace->nseg;
    if (aux->nseg != nseg || nseg == 0 || nseg > ZKM_MAX_SEG) throw std::runtime_error("zkm_quotient: batches of different stacks");
    const unsigned z = (unsigned)nseg;
    {
        size_t nl = 0;
        const zkm_table_lookup* defs = zkm_table_lookups(table_id, &nl);
        if (nl && !lookup_challenges) throw std::runtime_error("zkm_quotient: this table has lookups; lookup challenges are required");
        if (nl > 2) throw std::runtime_error("zkm_quotient: too many lookups");
        lookups.nlookups = (uint32_t)nl;
        lookups.nch = (uint32_t)nalphas;
        uint32_t off = 0;
        for (size_t l = 0; l < nl; l++) {
            if (off + defs[l].ncols > 24) throw std::runtime_error("zkm_quotient: too many lookup columns");
            lookups.lk[l] = {defs[l].ncols, off, defs[l].table_col, defs[l].freq_col};
            for (uint32_t i = 0; i < defs[l].ncols; i++) lookups.cols[off + i] = defs[l].cols[i];
            off += defs[l].ncols;
            NL += ((defs[l].ncols + 1) / 2 + 1) * (uint32_t)nalphas;
        }
        for (size_t sg = 0; nl && sg < nseg; sg++)
            for (size_t i = 0; i < nalphas; i++) lookup_ch.v[2 * sg + i] = lookup_challenges[sg * nalphas + i];
    }
    if (zkm_table_width(table_id) == 0 || trace->ncols != zkm_table_width(table_id))
        throw std::runtime_error("zkm_quotient: unknown table id, or the trace width does not match the table");
    if (trace->rate_bits != (check ? 0u : 2u) || aux->rate_bits != trace->rate_bits || trace->log_n != aux->log_n)
        throw std::runtime_error("zkm_quotient: rate_bits must be 2 and the batches must have equal degree");
    if (nalphas < 1 || nalphas > 2) throw std::runtime_error("zkm_quotient: 1 or 2 challenges supported");
    if (own.naux + NL != aux->ncols) throw std::runtime_error("zkm_quotient: aux column count does not match the CTL description");
    const ctl_dev& ctl = own.d;
    unsigned log_n = trace->log_n, lde_bits = check ? log_n : log_n + 2, log_q = check ? log_n : log_n + 1;
    size_t size = (size_t)1 << log_q;
    const size_t trace_seg = trace->lde_seg(), aux_seg = aux->lde_seg();
    gl_t w4 = gl_root_of_unity(lde_bits);
    const gl_t* wpow = c->pow_table(w4, lde_bits);
    gl_t gn = check ? 0 : gl_exp_pow2(GL_GENERATOR, log_n);      // (0: the kernels' check mode, quotient_setup)
    gl_t zh0 = check ? 1 : gl_inv(gl_sub(gn, 1)), zh1 = check ? 1 : gl_inv(gl_sub(gl_neg(gn), 1));
    gl_t w_n = gl_root_of_unity(log_n), last = gl_inv(w_n);
    gl_t n_inv = gl_inv((gl_t)(((uint64_t)1 << log_n) % GL_P));
    alpha_args d_alphas{};
    for (size_t sg = 0; sg < nseg; sg++)
        for (size_t a = 0; a < nalphas && a < 2; a++) d_alphas.v[2 * sg + a] = alphas_host[sg * nalphas + a];
    gl_t* d_vals = (gl_t*)c->alloc(nseg * nalphas * size * sizeof(gl_t));
    {
        static const char* const names[] = {"quotient_poseidon", "quotient_logic", "quotient_keccak_sponge", "quotient_keccak", "quotient_memory", "quotient_poseidon_sponge", "quotient_sha_extend", "quotient_sha_extend_sponge", "quotient_sha_compress",
                                            "quotient_sha_compress_sponge", "quotient_arithmetic", "quotient_cpu"};
        zkm_prof_scope ps(c, names[table_id]);
        dim3 grid((size + 255) / 256, 1, z), block(256);
#define ZKM_LAUNCH_QUOTIENT(T, NA)                                                                                              \
    hipLaunchKernelGGL((k_quotient<T, NA>), grid, block, 0, c->stream, trace->lde, aux->lde, log_n, lde_bits, lookups, d_alphas, lookup_ch, wpow, gn, \
                       last, w_n, n_inv, d_vals, trace_seg, aux_seg)
        if (table_id == ZKM_TABLE_KECCAK && size * nseg <= c->keccak_parts_max_points && !check) {
            // short table: 25 threads per point, then the sum of the parts (constraints_dev.h)
            const size_t per = nalphas * (KECCAK_NUM_CONSTRAINTS + 1);
            std::vector<gl_t> apw(nseg * per);
            for (size_t sg = 0; sg < nseg; sg++)
                for (size_t a = 0; a < nalphas; a++) {
                    gl_t p = 1;
                    for (size_t e = 0; e <= KECCAK_NUM_CONSTRAINTS; e++) {
                        apw[sg * per + a * (KECCAK_NUM_CONSTRAINTS + 1) + e] = p;
                        p = gl_mul(p, gl_canon(alphas_host[sg * nalphas + a]));
                    }
                }
            zkm_scratch d_apw(c, apw.size() * sizeof(gl_t)), d_tmp(c, nseg * (size_t)KECCAK_CONSTRAINT_PARTS * nalphas * size * sizeof(gl_t));
            c->upload(d_apw.p, apw.data(), apw.size() * sizeof(gl_t));
            dim3 gridp((unsigned)((size + 255) / 256), KECCAK_CONSTRAINT_PARTS, z);
            if (nalphas == 1)
                hipLaunchKernelGGL((k_quotient_keccak_parts<1>), gridp, block, 0, c->stream, trace->lde, lde_bits, d_alphas, wpow, gn, last, w_n, n_inv,
                                   d_apw.as<gl_t>(), d_tmp.as<gl_t>(), trace_seg);
            else
                hipLaunchKernelGGL((k_quotient_keccak_parts<2>), gridp, block, 0, c->stream, trace->lde, lde_bits, d_alphas, wpow, gn, last, w_n, n_inv,
                                   d_apw.as<gl_t>(), d_tmp.as<gl_t>(), trace_seg);
            hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((nalphas * size + 255) / 256), 1, z), block, 0, c->stream, d_tmp.as<gl_t>(),
                               (unsigned)KECCAK_CONSTRAINT_PARTS, nalphas * size, d_vals);
        } else
        switch (table_id * 2 + (int)nalphas - 1) {
            case 0: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON, 1); break;
            case 1: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON, 2); break;
            case 2: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_LOGIC, 1); break;
            case 3: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_LOGIC, 2); break;
            case 4: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK_SPONGE, 1); break;
            case 5: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK_SPONGE, 2); break;
            case 6: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK, 1); break;
            case 7: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK, 2); break;
            case 8: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_MEMORY, 1); break;
            case 9: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_MEMORY, 2); break;
            case 10: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON_SPONGE, 1); break;
            case 11: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON_SPONGE, 2); break;
            case 12: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND, 1); break;
            case 13: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND, 2); break;
            case 14: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND_SPONGE, 1); break;
            case 15: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND_SPONGE, 2); break;
            case 16: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS, 1); break;
            case 17: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS, 2); break;
            case 18: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS_SPONGE, 1); break;
            case 19: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS_SPONGE, 2); break;
            case 20: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_ARITHMETIC, 1); break;
            case 21: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_ARITHMETIC, 2); break;
            case 22: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_CPU, 1); break;
            default: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_CPU, 2); break;
        }
#undef ZKM_LAUNCH_QUOTIENT
        ZKM_HIP_CHECK(hipGetLastError());
    }
    {
        zkm_prof_scope ps(c, "quotient_ctl");
        dim3 grid((size + 255) / 256, 1, z), block(256);
        // chunk plan: at most CHUNK helper checks per chunk, the closing checks with the last chunk of their Z.  A thread's work is a
        // chain of column-combination evaluations whatever the table's height: on short tables (a launch cannot fill the GPU anyway)
        // one helper check per chunk -- four times the workgroups, a quarter of the chain (KeccakSponge at 2^8 rows: 340 us per launch)
        const bool short_table = size * nseg <= ((size_t)1 << 15);
        const uint32_t CHUNK = short_table ? 1 : 4;
        std::vector<ctl_chunk> plan;
        uint32_t K = 0, hstart = 0;
        std::vector<uint32_t> kend;
        for (uint32_t zi = 0; zi < own.h_zs.size(); zi++) {
            // helper checks exist only for real column sets (the benchmark's fake CTL data has helper columns but no sets)
            const uint32_t np = own.h_zs[zi].num_helpers ? (own.h_zs[zi].ncolsets + 1) / 2 : 0;
            uint32_t q = 0;
            do {
                const uint32_t q1 = np - q > CHUNK ? q + CHUNK : np;
                const bool closing = q1 == np;
                K += (q1 - q) + (closing ? 2 : 0);
                plan.push_back(ctl_chunk{zi, hstart, q, q1, closing ? 1u : 0u, 0});
                kend.push_back(K);
                q = q1;
            } while (q < np);
            hstart += own.h_zs[zi].num_helpers;
        }
        for (size_t i = 0; i < plan.size(); i++) plan[i].tail = K - kend[i];
        const bool chunked = plan.size() >= (short_table ? 2u : 8u) && size * nseg <= ((size_t)1 << 18) && !check;
        if (!chunked) {
            if (nalphas == 1)
                hipLaunchKernelGGL((k_quotient_ctl<1>), grid, block, 0, c->stream, trace->lde, aux->lde, lde_bits, ctl, NL, d_alphas, wpow, gn, zh0, zh1,
                                   last, w_n, n_inv, d_vals, trace_seg, aux_seg);
            else
                hipLaunchKernelGGL((k_quotient_ctl<2>), grid, block, 0, c->stream, trace->lde, aux->lde, lde_bits, ctl, NL, d_alphas, wpow, gn, zh0, zh1,
                                   last, w_n, n_inv, d_vals, trace_seg, aux_seg);
        } else {
            ctl_chunk* d_plan = (ctl_chunk*)c->alloc(plan.size() * sizeof(ctl_chunk));
            gl_t* d_tmp = (gl_t*)c->alloc(nseg * plan.size() * nalphas * size * sizeof(gl_t));
            c->upload(d_plan, plan.data(), plan.size() * sizeof(ctl_chunk));
            dim3 grid2((unsigned)((size + 255) / 256), (unsigned)plan.size(), z);
            if (nalphas == 1) {
                hipLaunchKernelGGL((k_quotient_ctl_chunk<1>), grid2, block, 0, c->stream, trace->lde, aux->lde, lde_bits, ctl, d_plan, NL, d_alphas, wpow,
                                   gn, last, w_n, n_inv, d_tmp, trace_seg, aux_seg);
                hipLaunchKernelGGL((k_quotient_ctl_sum<1>), grid, block, 0, c->stream, d_tmp, (uint32_t)plan.size(), K, d_alphas, zh0, zh1, size, d_vals);
            } else {
                hipLaunchKernelGGL((k_quotient_ctl_chunk<2>), grid2, block, 0, c->stream, trace->lde, aux->lde, lde_bits, ctl, d_plan, NL, d_alphas, wpow,
                                   gn, last, w_n, n_inv, d_tmp, trace_seg, aux_seg);
                hipLaunchKernelGGL((k_quotient_ctl_sum<2>), grid, block, 0, c->stream, d_tmp, (uint32_t)plan.size(), K, d_alphas, zh0, zh1, size, d_vals);
            }
            ZKM_HIP_CHECK(hipGetLastError());
            // (no host sync: upload() has copied `plan` into the pinned ring -- or waited itself if it was too large for it -- and the
            // blocks released here are only reused by later work on this stream)
            c->release(d_plan);
            c->release(d_tmp);
        }
        ZKM_HIP_CHECK(hipGetLastError());
    }
    if (check) {
        ZKM_HIP_CHECK(hipMemcpyAsync(d_out, d_vals, nseg * nalphas * size * sizeof(gl_t), hipMemcpyDeviceToDevice, c->stream));
        c->release(d_vals);
        return;
    }
    // coset_ifft(g) of each challenge's evaluations (prover.rs:784-788)
    zkm_ntt_natural(c, d_vals, d_out, nseg * nalphas, size, size, log_q, true, GL_GENERATOR);
    c->release(d_vals);  // stream-ordered reuse: no host sync needed
}

// ------------------------------------------------------------------ K10: openings
// partial[col][chunk] = sum_{k in chunk} c_k z^(k - chunk_start) for z in {zeta, g*zeta}, plus the plain sum (evaluation
// at 1).  The coefficients are base-field elements, so with a table of the powers z^k = (a_k, b_k) of the chunk (shared by
// every column and chunk; 4 x 2^14 words, L2-resident) each coefficient costs four BASE multiplications -- sum c_k a_k,
// sum c_k b_k for the two points -- instead of two extension-field Horner steps, and the sums are accumulated lazily
// (64-bit wrapping adds with a wrap counter, 2^64 == EPS applied once at the end).  A workgroup handles OPEN_CPB columns
// of one chunk so a power is loaded once per OPEN_CPB coefficients; all loads are coalesced.
#define OPEN_CHUNK_LOG 14
#define OPEN_CPB 4
// pw[.][k] = z^(exponent of position k of a column): the batch's coefficient layout (zkm_coeff_exponent) is a permutation of the BITS
// of the position, so the exponent of position chunk * chunk_len + k is e(k) + e(chunk * chunk_len) and the table of one chunk serves all.
// (blockIdx.z = segment of a stack: its own two points, its own table)
__global__ __launch_bounds__(256) void k_open_powers(seg_gl2 z0s, seg_gl2 z1s, unsigned chunk_len, unsigned coeff_s1,
                                                     gl_t* __restrict__ pw /* [segment][4][chunk_len] */) {
    unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= chunk_len) return;
    const gl2_t z0{z0s.v[2 * blockIdx.z], z0s.v[2 * blockIdx.z + 1]}, z1{z1s.v[2 * blockIdx.z], z1s.v[2 * blockIdx.z + 1]};
    pw += (size_t)blockIdx.z * 4 * chunk_len;
    const uint64_t e = zkm_coeff_exponent(k, coeff_s1);
    gl2_t p0 = gl2_pow(z0, e), p1 = gl2_pow(z1, e);
    pw[k] = p0.c0; pw[chunk_len + k] = p0.c1; pw[2 * chunk_len + k] = p1.c0; pw[3 * chunk_len + k] = p1.c1;
}
struct lazy_sum {
    uint64_t lo;
    uint32_t wraps;
    __device__ __forceinline__ void add(uint64_t v) {
        lo += v;
        wraps += lo < v;
    }
    __device__ __forceinline__ gl_t value() const {  // lo + wraps 2^64 == lo + wraps EPS (wraps < 2^32: one multiply-add)
        uint64_t r = (uint64_t)wraps * 0xFFFFFFFFu + lo;
        r += (r < lo) ? GL_EPS : 0;
        return gl_canon(r);
    }
};
// (CPB columns per workgroup: 4 amortises the power loads when the launch fills the machine anyway; 1 when it would be a few hundred
// workgroups -- tables of 2^16 / 2^17 rows: four times the waves for the same work)
template <int CPB>
__global__ __launch_bounds__(256) void k_open_partials(const gl_t* __restrict__ coeffs, unsigned log_n, size_t ncols,
                                                       const gl_t* __restrict__ pw, gl_t* __restrict__ partial /* [col][chunk][5] */,
                                                       size_t coeff_seg, size_t partial_seg) {
    __shared__ gl_t red[256 * 5];
    const unsigned chunk_log = log_n < OPEN_CHUNK_LOG ? log_n : OPEN_CHUNK_LOG;
    const size_t chunk_len = (size_t)1 << chunk_log, nchunks = (size_t)1 << (log_n - chunk_log);
    coeffs += (size_t)blockIdx.z * coeff_seg;
    pw += (size_t)blockIdx.z * 4 * chunk_len;
    partial += (size_t)blockIdx.z * partial_seg;
    const size_t col0 = (size_t)blockIdx.y * CPB, chunk = blockIdx.x;
    const unsigned t = threadIdx.x;
    lazy_sum acc[CPB][5];
#pragma unroll
    for (int c = 0; c < CPB; c++)
#pragma unroll
        for (int q = 0; q < 5; q++) acc[c][q] = lazy_sum{0, 0};
    // (software-pipelined by hand: the loads of step k + 1 -- four powers, CPB coefficients -- are issued before the 4 CPB products of step
    // k; with one load-use pair per iteration every one of the 64 steps of a chunk exposed a trip to L2 / HBM at four waves per SIMD)
    gl_t np[4], ncv[CPB];
    auto fetch = [&](size_t idx) {
#pragma unroll
        for (int q = 0; q < 4; q++) np[q] = pw[(size_t)q * chunk_len + idx];
#pragma unroll
        for (int c = 0; c < CPB; c++) {
            // (columns past the end of the batch read column ncols - 1 again; their sums are never stored)
            const size_t col = col0 + c < ncols ? col0 + c : ncols - 1;
            ncv[c] = coeffs[(col << log_n) + chunk * chunk_len + idx];
        }
    };
    if (t < chunk_len) fetch(t);
    for (size_t idx = t; idx < chunk_len; idx += 256) {
        gl_t p[4], cvs[CPB];
#pragma unroll
        for (int q = 0; q < 4; q++) p[q] = np[q];
#pragma unroll
        for (int c = 0; c < CPB; c++) cvs[c] = ncv[c];
        if (idx + 256 < chunk_len) fetch(idx + 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CPB; c++) {
            const gl_t cv = cvs[c];
#pragma unroll
            for (int q = 0; q < 4; q++) acc[c][q].add(gl_mul_loose(cv, p[q]));
            acc[c][4].add(cv);
        }
    }
#pragma unroll
    for (int c = 0; c < CPB; c++) {
#pragma unroll
        for (int q = 0; q < 5; q++) red[t * 5 + q] = acc[c][q].value();
        __syncthreads();
        for (unsigned s = 128; s > 0; s >>= 1) {
            if (t < s)
                for (int q = 0; q < 5; q++) red[t * 5 + q] = gl_add(red[t * 5 + q], red[(t + s) * 5 + q]);
            __syncthreads();
        }
        if (t < 5 && col0 + c < ncols) partial[((col0 + c) * nchunks + chunk) * 5 + t] = red[t];
        __syncthreads();
    }
}

struct open_vals { gl2_t at_z0, at_z1; gl_t at_one; };

// evaluate every polynomial of the batches (same height, same coefficient layout) at z0, z1 (F2) and 1: one power table, one launch
// per batch, one download.  Stacked batches (nseg segments each): z0 / z1 hold one point per segment, out[s][batch][poly].
static std::vector<std::vector<std::vector<open_vals>>> eval_batches(zkm_ctx* c, const std::vector<const zkm_batch*>& bs, const gl2_t* z0,
                                                                     const gl2_t* z1) {
    const zkm_batch* b0 = bs.at(0);
    const size_t nseg = b0->nseg;
    const unsigned log_n = b0->log_n, chunk_log = log_n < OPEN_CHUNK_LOG ? log_n : OPEN_CHUNK_LOG, chunk_len = 1u << chunk_log;
    const size_t nchunks = (size_t)1 << (log_n - chunk_log);
    size_t total_cols = 0;
    for (const zkm_batch* b : bs) {
        if (b->log_n != log_n || b->coeff_s1 != b0->coeff_s1 || b->nseg != nseg) throw std::runtime_error("internal: openings of batches with different shapes");
        total_cols += b->ncols;
    }
    if (nseg == 0 || nseg > ZKM_MAX_SEG) throw std::runtime_error("internal: openings of a stack of unsupported size");
    const unsigned z = (unsigned)nseg;
    const size_t words_seg = total_cols * nchunks * 5, words = nseg * words_seg;
    zkm_scratch d_part(c, words * sizeof(gl_t)), d_pw(c, nseg * 4 * chunk_len * sizeof(gl_t));
    {
        zkm_prof_scope ps(c, "open_partials");
        seg_gl2 z0s{}, z1s{};
        for (size_t sg = 0; sg < nseg; sg++) { z0s.v[2 * sg] = z0[sg].c0; z0s.v[2 * sg + 1] = z0[sg].c1; z1s.v[2 * sg] = z1[sg].c0; z1s.v[2 * sg + 1] = z1[sg].c1; }
        hipLaunchKernelGGL(k_open_powers, dim3((chunk_len + 255) / 256, 1, z), dim3(256), 0, c->stream, z0s, z1s, chunk_len, b0->coeff_s1, d_pw.as<gl_t>());
        size_t col0 = 0;
        for (const zkm_batch* b : bs) {
            if (nseg * nchunks * ((b->ncols + OPEN_CPB - 1) / OPEN_CPB) < 1024)
                hipLaunchKernelGGL(k_open_partials<1>, dim3(nchunks, b->ncols, z), dim3(256), 0, c->stream, b->coeffs, log_n, b->ncols, d_pw.as<gl_t>(),
                                   d_part.as<gl_t>() + col0 * nchunks * 5, b->coeff_seg(), words_seg);
            else
                hipLaunchKernelGGL(k_open_partials<OPEN_CPB>, dim3(nchunks, (b->ncols + OPEN_CPB - 1) / OPEN_CPB, z), dim3(256), 0, c->stream, b->coeffs,
                                   log_n, b->ncols, d_pw.as<gl_t>(), d_part.as<gl_t>() + col0 * nchunks * 5, b->coeff_seg(), words_seg);
            col0 += b->ncols;
        }
        ZKM_HIP_CHECK(hipGetLastError());
    }
    std::vector<gl_t> part(words);
    c->download(part.data(), d_part.p, words * sizeof(gl_t));
    std::vector<std::vector<std::vector<open_vals>>> all(nseg);
    std::vector<gl2_t> f0(nchunks), f1(nchunks);
    for (size_t sg = 0; sg < nseg; sg++) {
        // chunk ch contributes z^(exponent of its first position) * partial (natural order: z^(ch * chunk_len), i.e. Horner over the chunks)
        for (size_t ch = 0; ch < nchunks; ch++) {
            const uint64_t e = zkm_coeff_exponent((uint32_t)(ch << chunk_log), b0->coeff_s1);
            f0[ch] = gl2_pow(z0[sg], e);
            f1[ch] = gl2_pow(z1[sg], e);
        }
        size_t col0 = 0;
        for (const zkm_batch* b : bs) {
            std::vector<open_vals> o(b->ncols);
            for (size_t col = 0; col < b->ncols; col++) {
                gl2_t a0{0, 0}, a1{0, 0};
                gl_t sum = 0;
                for (size_t ch = 0; ch < nchunks; ch++) {
                    const gl_t* q = &part[sg * words_seg + ((col0 + col) * nchunks + ch) * 5];
                    a0 = gl2_add(a0, gl2_mul(f0[ch], gl2_t{q[0], q[1]}));
                    a1 = gl2_add(a1, gl2_mul(f1[ch], gl2_t{q[2], q[3]}));
                    sum = gl_add(sum, q[4]);
                }
                o[col] = open_vals{a0, a1, sum};
            }
            all[sg].push_back(std::move(o));
            col0 += b->ncols;
        }
    }
    return all;
}

// ------------------------------------------------------------------ K11: FRI batch combination
// comp1 = sum_{j < W+A} alpha^j p_j (g*zeta batch), comp0 = comp1 + sum_q alpha^(W+A+q) quot_q (zeta batch),
// comp2 = sum_k alpha^k ctl_z_k (batch at 1)   -- polynomial order of stark.rs:127-148.  One pass over all
// coefficient polynomials; alpha powers are wave-uniform.
__global__ __launch_bounds__(256) void k_fri_combine(const gl_t* __restrict__ tc, size_t W, const gl_t* __restrict__ ac, size_t A,
                                                     const gl_t* __restrict__ qc, size_t Q, size_t ctl_start,
                                                     const gl_t* __restrict__ apow /* [(W+A+Q)][2] */, size_t n,
                                                     gl_t* __restrict__ comp /* [3][2][n] */, size_t apow_seg) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // (blockIdx.z = segment of a stack: its coefficient columns follow the previous segment's, its own powers of its own alpha)
    tc += (size_t)blockIdx.z * W * n; ac += (size_t)blockIdx.z * A * n; qc += (size_t)blockIdx.z * Q * n;
    apow += (size_t)blockIdx.z * apow_seg;
    comp += (size_t)blockIdx.z * 6 * n;
    gl_t a0 = 0, a1 = 0, z0 = 0, z1 = 0;
    size_t j = 0;
    for (size_t cidx = 0; cidx < W; cidx++, j++) {
        gl_t v = tc[cidx * n + i];
        a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
        a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
    }
    for (size_t cidx = 0; cidx < A; cidx++, j++) {
        gl_t v = ac[cidx * n + i];
        a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
        a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
        if (cidx >= ctl_start) {
            size_t k = cidx - ctl_start;
            z0 = gl_add(z0, gl_mul(v, apow[2 * k]));
            z1 = gl_add(z1, gl_mul(v, apow[2 * k + 1]));
        }
    }
    comp[2 * n + i] = a0;
    comp[3 * n + i] = a1;
    for (size_t cidx = 0; cidx < Q; cidx++, j++) {
        gl_t v = qc[cidx * n + i];
        a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
        a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
    }
    comp[i] = a0;
    comp[n + i] = a1;
    comp[4 * n + i] = z0;
    comp[5 * n + i] = z1;
}

// Short, wide tables (Keccak: 2431 columns x 2^11 rows): one thread per coefficient index leaves the machine empty and walks thousands
// of columns serially (1 ms for that table).  Here blockIdx.y takes a slice of the W + A + Q polynomials and writes its partial sums
// (trace + aux part, quotient part, CTL part) to part[slice][6][n]; k_fri_combine_sum adds the slices.  Field addition is exact, so the
// result is the same words as k_fri_combine's.
__global__ __launch_bounds__(256) void k_fri_combine_slice(const gl_t* __restrict__ tc, size_t W, const gl_t* __restrict__ ac, size_t A,
                                                           const gl_t* __restrict__ qc, size_t Q, size_t ctl_start,
                                                           const gl_t* __restrict__ apow, size_t n, size_t per_slice,
                                                           gl_t* __restrict__ part /* [segment][slices][6][n] */, size_t apow_seg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tc += (size_t)blockIdx.z * W * n; ac += (size_t)blockIdx.z * A * n; qc += (size_t)blockIdx.z * Q * n;
    apow += (size_t)blockIdx.z * apow_seg;
    part += (size_t)blockIdx.z * gridDim.y * 6 * n;
    const size_t j0 = (size_t)blockIdx.y * per_slice, total = W + A + Q;
    const size_t j1 = j0 + per_slice < total ? j0 + per_slice : total;
    gl_t a0 = 0, a1 = 0, q0 = 0, q1 = 0, z0 = 0, z1 = 0;
    for (size_t j = j0; j < j1; j++) {
        if (j < W + A) {
            const gl_t v = j < W ? tc[j * n + i] : ac[(j - W) * n + i];
            a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
            a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
            if (j >= W + ctl_start) {
                const size_t k = j - W - ctl_start;
                z0 = gl_add(z0, gl_mul(v, apow[2 * k]));
                z1 = gl_add(z1, gl_mul(v, apow[2 * k + 1]));
            }
        } else {
            const gl_t v = qc[(j - W - A) * n + i];
            q0 = gl_add(q0, gl_mul(v, apow[2 * j]));
            q1 = gl_add(q1, gl_mul(v, apow[2 * j + 1]));
        }
    }
    gl_t* o = part + (size_t)blockIdx.y * 6 * n + i;
    o[0] = a0; o[n] = a1; o[2 * n] = q0; o[3 * n] = q1; o[4 * n] = z0; o[5 * n] = z1;
}
__global__ __launch_bounds__(256) void k_fri_combine_sum(const gl_t* __restrict__ part, unsigned slices, size_t n, gl_t* __restrict__ comp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    part += (size_t)blockIdx.z * slices * 6 * n;
    comp += (size_t)blockIdx.z * 6 * n;
    gl_t t[6] = {0, 0, 0, 0, 0, 0};
    for (unsigned s = 0; s < slices; s++)
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = gl_add(t[k], part[((size_t)s * 6 + k) * n + i]);
    comp[2 * n + i] = t[0];
    comp[3 * n + i] = t[1];
    comp[i] = gl_add(t[0], t[2]);
    comp[n + i] = gl_add(t[1], t[3]);
    comp[4 * n + i] = t[4];
    comp[5 * n + i] = t[5];
}

// divide_by_linear as a hierarchical suffix scan with segments of FRI_SEG (profiles/HISTORY.md), for ALL batches of the instance in one set of
// launches (blockIdx.y = batch; the STARK instance has three: zeta, g zeta, 1).  Per batch b: q_b = (comp_b(X) - comp_b(z_b)) / (X - z_b),
// q_b[k - 1] = S_b[k] with S_b[k] = a_b[k] + z_b S_b[k + 1]; the final polynomial is sum_b w_b q_b, w_b = prod_{b' > b} shift_b'
// (plonky2 accumulates final = final * alpha^(#polys of the batch) + quotient, batch by batch: the same sum, regrouped).
// A thread walks ONE segment of ONE batch -- a chain of FRI_SEG dependent extension-field multiply-adds; the weighted sum over the
// batches is a separate element-wise launch.  (Until round 3: segments of 64 and the bottom level walking all batches in one
// thread, 192 dependent steps: 158 us per table in a 2^16-cycle segment, whatever its size.)
#define FRI_MAX_BATCHES 8
// (segment length: the chain a thread walks.  With the LDS tiles the kernels are chains of dependent extension-field steps at low
// occupancy, so shorter segments -- more, cheaper levels -- win: 2^22 coefficients, three batches: 0.70 ms at 32, 0.46 at 16, 0.36 at 8)
#ifndef FRI_SEG
#define FRI_SEG 8
#endif
struct seg_batches {
    uint32_t nb;
    const gl_t* a0[FRI_MAX_BATCHES];   // this level's arrays of every batch (level 0: the composite polynomials)
    const gl_t* a1[FRI_MAX_BATCHES];
    // Stacked proofs (blockIdx.z = segment): every array of segment s starts a_seg words after segment s - 1's, and the points / weights
    // are per segment: zw[(s * nb + b) * 2] = z_b^(FRI_SEG^level), [.. + 1] = weight of the batch in the final sum
    size_t a_seg;
    const gl2_t* zw;
};
__device__ __forceinline__ gl2_t seg_point(const seg_batches& p, unsigned b) { return p.zw[((size_t)blockIdx.z * p.nb + b) * 2]; }
__device__ __forceinline__ gl2_t seg_weight(const seg_batches& p, unsigned b) { return p.zw[((size_t)blockIdx.z * p.nb + b) * 2 + 1]; }
// A thread's segment is FRI_SEG consecutive words of each array, and the walk over it is a dependent chain: with one 8-byte load per
// step the lanes of a wave sit 256 B apart and every step fetched a whole line from HBM for 8 bytes of it (config 4: 6.4 GB of traffic
// for 0.27 GB of data, profiles/r04_e).  All three kernels therefore work on LDS TILES: the 64 segments of a 64-thread workgroup
// (64 x FRI_SEG consecutive words) are loaded with lane-contiguous accesses into rows of FRI_SEG + 1 words (a thread walking its own row is
// bank-conflict free), results that are arrays over k go back the same way.  Words past the end of the array read as zero -- the
// carry into the last segment is zero, so the chain stays zero there.
#define SEG_TILE (64 * FRI_SEG)
#define SEG_ROW (FRI_SEG + 1)
// (a wave's own LDS accesses execute in order: a wave that reads back what only IT wrote needs no workgroup barrier)
#define ZKM_WAVE_SYNC_LDS()                                  \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
__device__ __forceinline__ void seg_tile_load(gl_t* __restrict__ t, const gl_t* __restrict__ a, size_t base, size_t m) {
    for (unsigned e = threadIdx.x; e < SEG_TILE; e += 64) {
        const size_t k = base + e;
        t[(e / FRI_SEG) * SEG_ROW + e % FRI_SEG] = k < m ? a[k] : 0;
    }
}
// totals:  out_b[s] = sum_{k<FRI_SEG} a_b[FRI_SEG s + k] z_b^k          (out: [batch][2][nseg])
__global__ __launch_bounds__(64) void k_seg_totals(seg_batches p, size_t m, gl_t* __restrict__ out) {
    __shared__ gl_t t0[64 * SEG_ROW], t1[64 * SEG_ROW];
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x, nseg = (m + FRI_SEG - 1) / FRI_SEG;
    const unsigned b = blockIdx.y;
    seg_tile_load(t0, p.a0[b] + (size_t)blockIdx.z * p.a_seg, (size_t)blockIdx.x * SEG_TILE, m);
    seg_tile_load(t1, p.a1[b] + (size_t)blockIdx.z * p.a_seg, (size_t)blockIdx.x * SEG_TILE, m);
    __syncthreads();
    if (s >= nseg) return;
    out += (size_t)blockIdx.z * 2 * gridDim.y * nseg;
    const gl2_t z = seg_point(p, b);
    gl2_t acc{0, 0};
    const gl_t *r0 = t0 + threadIdx.x * SEG_ROW, *r1 = t1 + threadIdx.x * SEG_ROW;
    for (unsigned k = FRI_SEG; k-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_t{r0[k], r1[k]});
    out[(2 * b) * nseg + s] = acc.c0;
    out[(2 * b + 1) * nseg + s] = acc.c1;
}
// scan of a level:  S_b[k] = a_b[k] + z_b S_b[k+1] inside each segment, carry-in = upper_b[s+1] (0 past the end)
// (upper: [batch][2][nupper] suffix values of the level above, or null at the top; out: [batch][2][m])
__global__ __launch_bounds__(64) void k_seg_scan(seg_batches p, size_t m, const gl_t* __restrict__ upper, size_t nupper, gl_t* __restrict__ out) {
    __shared__ gl_t t0[64 * SEG_ROW], t1[64 * SEG_ROW];
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x, base = (size_t)blockIdx.x * SEG_TILE;
    const unsigned b = blockIdx.y;
    seg_tile_load(t0, p.a0[b] + (size_t)blockIdx.z * p.a_seg, base, m);
    seg_tile_load(t1, p.a1[b] + (size_t)blockIdx.z * p.a_seg, base, m);
    __syncthreads();
    if (upper) upper += (size_t)blockIdx.z * 2 * gridDim.y * nupper;
    out += (size_t)blockIdx.z * 2 * gridDim.y * m;
    {
        const gl2_t z = seg_point(p, b);
        gl2_t acc{0, 0};
        if (upper && s + 1 < nupper) acc = gl2_t{upper[(2 * b) * nupper + s + 1], upper[(2 * b + 1) * nupper + s + 1]};
        gl_t *r0 = t0 + threadIdx.x * SEG_ROW, *r1 = t1 + threadIdx.x * SEG_ROW;
        for (unsigned k = FRI_SEG; k-- > 0;) {                 // (a segment past the end: zeros in, zero carry, zeros out -- never stored)
            acc = gl2_add(gl2_mul(acc, z), gl2_t{r0[k], r1[k]});
            r0[k] = acc.c0;
            r1[k] = acc.c1;
        }
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < SEG_TILE; e += 64) {
        const size_t k = base + e;
        if (k < m) {
            out[(2 * b) * m + k] = t0[(e / FRI_SEG) * SEG_ROW + e % FRI_SEG];
            out[(2 * b + 1) * m + k] = t1[(e / FRI_SEG) * SEG_ROW + e % FRI_SEG];
        }
    }
}
// bottom level of LONG polynomials (>= 2^21 coefficients: enough segments to fill the machine, and the suffix values of every batch
// would be 6 n words written and read again): all batches in one workgroup, one after the other through the same input tile, with
// fin[k - 1] = sum_b w_b S_b[k] accumulated in an output tile
__global__ __launch_bounds__(64) void k_seg_scan_final(seg_batches p, size_t m, const gl_t* __restrict__ upper, size_t nupper, gl_t* __restrict__ f0,
                                                       gl_t* __restrict__ f1) {
    extern __shared__ __attribute__((aligned(16))) gl_t seg_lds[];   // four tiles of 64 rows x (FRI_SEG + 1) words: 18 KB at FRI_SEG = 8 (the dynamic-LDS limit is only raised for builds with longer segments)
    gl_t *const t0 = seg_lds, *const t1 = t0 + 64 * SEG_ROW, *const g0 = t1 + 64 * SEG_ROW, *const g1 = g0 + 64 * SEG_ROW;
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x, base = (size_t)blockIdx.x * SEG_TILE;
    gl_t *r0 = t0 + threadIdx.x * SEG_ROW, *r1 = t1 + threadIdx.x * SEG_ROW, *q0 = g0 + threadIdx.x * SEG_ROW, *q1 = g1 + threadIdx.x * SEG_ROW;
    if (upper) upper += (size_t)blockIdx.z * 2 * p.nb * nupper;
    f0 += (size_t)blockIdx.z * 2 * m;
    f1 += (size_t)blockIdx.z * 2 * m;
    for (unsigned b = 0; b < p.nb; b++) {
        if (b) __syncthreads();                                // everyone is done with the previous batch's tile
        seg_tile_load(t0, p.a0[b] + (size_t)blockIdx.z * p.a_seg, base, m);
        seg_tile_load(t1, p.a1[b] + (size_t)blockIdx.z * p.a_seg, base, m);
        __syncthreads();
        const gl2_t z = seg_point(p, b), w = seg_weight(p, b);
        gl2_t acc{0, 0};
        if (upper && s + 1 < nupper) acc = gl2_t{upper[(2 * b) * nupper + s + 1], upper[(2 * b + 1) * nupper + s + 1]};
        for (unsigned k = FRI_SEG; k-- > 0;) {
            acc = gl2_add(gl2_mul(acc, z), gl2_t{r0[k], r1[k]});
            gl2_t f = gl2_mul(acc, w);
            if (b) f = gl2_add(f, gl2_t{q0[k], q1[k]});
            q0[k] = f.c0;
            q1[k] = f.c1;
        }
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < SEG_TILE; e += 64) {   // word k of the tile is fin[k - 1]
        const size_t k = base + e;
        if (k > 0 && k < m) {
            f0[k - 1] = g0[(e / FRI_SEG) * SEG_ROW + e % FRI_SEG];
            f1[k - 1] = g1[(e / FRI_SEG) * SEG_ROW + e % FRI_SEG];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { f0[m - 1] = 0; f1[m - 1] = 0; }
}
// Bottom level, scan and weighted sum in ONE launch (round 5): wave b of the workgroup scans batch b's 64 segments into its own LDS
// tile (the same 8-step chains as k_seg_scan, the batches side by side instead of one after the other as in k_seg_scan_final), then
// all waves add the tiles up: fin[k - 1] = sum_b w_b S_b[k].  The suffix values of the bottom level never go to HBM: the division
// reads the composites twice (totals, then here) and writes fin once -- 1.75x its algorithmic bytes instead of 3.9x (k_seg_scan writing
// 6 n words that k_seg_combine read again).  Field sums are exact, so the words are those of k_seg_scan + k_seg_combine.
__global__ __launch_bounds__(512) void k_seg_scan_combine(seg_batches p, size_t m, const gl_t* __restrict__ upper, size_t nupper, gl_t* __restrict__ f0,
                                                          gl_t* __restrict__ f1) {
    extern __shared__ __attribute__((aligned(16))) gl_t sc_lds[];        // [batch][2][64 * SEG_ROW]
    const unsigned b = threadIdx.x >> 6, lane = threadIdx.x & 63;
    gl_t *const t0 = sc_lds + (size_t)(2 * b) * 64 * SEG_ROW, *const t1 = t0 + 64 * SEG_ROW;
    const size_t s = (size_t)blockIdx.x * 64 + lane, base = (size_t)blockIdx.x * SEG_TILE;
    {
        const gl_t* a0 = p.a0[b] + (size_t)blockIdx.z * p.a_seg;
        const gl_t* a1 = p.a1[b] + (size_t)blockIdx.z * p.a_seg;
        for (unsigned e = lane; e < SEG_TILE; e += 64) {
            const size_t k = base + e;
            const unsigned at = (e / FRI_SEG) * SEG_ROW + e % FRI_SEG;
            t0[at] = k < m ? a0[k] : 0;
            t1[at] = k < m ? a1[k] : 0;
        }
    }
    ZKM_WAVE_SYNC_LDS();
    {
        if (upper) upper += (size_t)blockIdx.z * 2 * p.nb * nupper;
        const gl2_t z = seg_point(p, b);
        gl2_t acc{0, 0};
        if (upper && s + 1 < nupper) acc = gl2_t{upper[(2 * b) * nupper + s + 1], upper[(2 * b + 1) * nupper + s + 1]};
        gl_t *r0 = t0 + lane * SEG_ROW, *r1 = t1 + lane * SEG_ROW;
        const gl2_t w = seg_weight(p, b);
        for (unsigned k = FRI_SEG; k-- > 0;) {
            acc = gl2_add(gl2_mul(acc, z), gl2_t{r0[k], r1[k]});
            const gl2_t f = gl2_mul(acc, w);                   // (the weight goes in here: the sum below is plain additions)
            r0[k] = f.c0;
            r1[k] = f.c1;
        }
    }
    __syncthreads();
    f0 += (size_t)blockIdx.z * 2 * m;
    f1 += (size_t)blockIdx.z * 2 * m;
    for (unsigned e = threadIdx.x; e < SEG_TILE; e += blockDim.x) {       // word k of the tiles is fin[k - 1]
        const size_t k = base + e;
        if (k > 0 && k < m) {
            const unsigned at = (e / FRI_SEG) * SEG_ROW + e % FRI_SEG;
            gl_t x0 = 0, x1 = 0;
            for (unsigned q = 0; q < p.nb; q++) {
                x0 = gl_add(x0, sc_lds[(size_t)(2 * q) * 64 * SEG_ROW + at]);
                x1 = gl_add(x1, sc_lds[(size_t)(2 * q + 1) * 64 * SEG_ROW + at]);
            }
            f0[k - 1] = x0;
            f1[k - 1] = x1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { f0[m - 1] = 0; f1[m - 1] = 0; }
}
// fin[k - 1] = sum_b w_b S_b[k] (the dropped remainders are the S_b[0]), fin[m - 1] = 0      (suf: [batch][2][m], bottom level)
__global__ __launch_bounds__(256) void k_seg_combine(seg_batches p, size_t m, const gl_t* __restrict__ suf, gl_t* __restrict__ f0,
                                                     gl_t* __restrict__ f1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    suf += (size_t)blockIdx.z * 2 * p.nb * m;
    f0 += (size_t)blockIdx.z * 2 * m;
    f1 += (size_t)blockIdx.z * 2 * m;
    gl2_t f{0, 0};
    if (i + 1 < m)
        for (unsigned b = 0; b < p.nb; b++) f = gl2_add(f, gl2_mul(gl2_t{suf[(2 * b) * m + i + 1], suf[(2 * b + 1) * m + i + 1]}, seg_weight(p, b)));
    f0[i] = f.c0;
    f1[i] = f.c1;
}

struct fri_composite;
// fin = sum over the batches (in order) of: fin * shift_b + (comp_b(X) - comp_b(z_b)) / (X - z_b), re-padded to n coefficients.
// fin is written completely (no zero-fill needed).  Stacked: comps[s] = segment s's composites (pointers: segment 0's, the others'
// comp_seg words further per segment), fin of segment s at f0 / f1 + s * 2 n.
static void divide_accumulate_all(zkm_ctx* c, const std::vector<std::vector<fri_composite>>& comps, size_t comp_seg, size_t n, gl_t* f0, gl_t* f1);

// ------------------------------------------------------------------ K13: FRI fold
// c'_j = sum_{i < arity} beta^i c_{arity j + i}   (reduce_with_powers per chunk)
// A thread's 16 coefficients are one 128-byte line of each array: read with one 8-byte load per step they were fetched from HBM again
// and again (lanes 128 B apart, the lines evicted between steps: 850 MB of traffic for 76 MB of data in config 4, profiles/r04_e).  The
// workgroup's 64 x arity coefficients therefore come in with lane-contiguous loads into an LDS tile (row stride arity + 1: the walk
// over one's own row is bank-conflict free) and are read from there.
// (tile rows of arity + 1 words in dynamic LDS: any arity_bits the config check admits, 2 .. 6 -- ADVICE r04)
// (blockIdx.z = segment of a stack: its own beta; its arrays [2][total] in, [2][nout] out follow the previous segment's)
__global__ __launch_bounds__(64) void k_fri_fold(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1, size_t nout, unsigned arity,
                                                 seg_gl2 betas, gl_t* __restrict__ o0, gl_t* __restrict__ o1) {
    extern __shared__ __attribute__((aligned(16))) gl_t fold_lds[];   // two tiles of 64 rows of arity + 1 words
    gl_t *const t0 = fold_lds, *const t1 = fold_lds + 64 * (arity + 1);
    const gl2_t beta{betas.v[2 * blockIdx.z], betas.v[2 * blockIdx.z + 1]};
    const size_t j0 = (size_t)blockIdx.x * 64, total = nout * arity, base = j0 * arity;
    c0 += (size_t)blockIdx.z * 2 * total; c1 += (size_t)blockIdx.z * 2 * total;
    o0 += (size_t)blockIdx.z * 2 * nout; o1 += (size_t)blockIdx.z * 2 * nout;
    const unsigned tid = threadIdx.x, rs = arity + 1;
    for (unsigned e = tid; e < 64 * arity; e += 64) {
        const size_t k = base + e;
        const unsigned at = (e / arity) * rs + e % arity;
        t0[at] = k < total ? c0[k] : 0;
        t1[at] = k < total ? c1[k] : 0;
    }
    __syncthreads();
    const size_t j = j0 + tid;
    if (j >= nout) return;
    gl2_t acc{0, 0};
    for (unsigned i = arity; i-- > 0;) acc = gl2_add(gl2_mul(acc, beta), gl2_t{t0[tid * rs + i], t1[tid * rs + i]});
    o0[j] = acc.c0;
    o1[j] = acc.c1;
}

// ------------------------------------------------------------------ K14: proof of work
// Smallest candidate w such that Poseidon(state with w written at `pos`)[7] has >= pow_bits leading zeros.  One launch: thread t tries
// base + t, base + t + stride, ... below `limit` and stops as soon as a smaller hit is known (device-scope atomic on `best`), so a
// launch costs about one permutation's latency per 2^pow_bits / stride-th of the expected number of trials -- no host round trip
// between the rounds, no candidates tried beyond the round of the first hit.
struct pow_state { uint64_t s[12]; };
// The launch delivers its result ITSELF: the workgroup that finishes last (a ticket) writes `best` into pinned host memory, then the
// sequence number the host polls for, and leaves `best` = all ones and the ticket counter = 0 for the next search -- no memset launch in
// front, no download launch behind (two launches and their gaps per table proof).
// Stacked searches (blockIdx.y = segment): sts[s] = segment s's transcript state (device array), pos.v[s] the word its candidate goes
// to, best[s] / host_out[s] its result; a segment whose witness is already known (done.v[s] != 0: found by an earlier launch) is skipped.
struct seg_u32 { uint32_t v[ZKM_MAX_SEG]; };
__global__ __launch_bounds__(256) void k_pow_search(const pow_state* __restrict__ sts, seg_u32 pos_v, seg_u32 done, unsigned pow_bits, uint64_t base,
                                                    uint64_t limit, unsigned long long* best, unsigned* counter, uint64_t* host_out, uint64_t* flag,
                                                    uint64_t seq) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const unsigned pos = pos_v.v[blockIdx.y];
    best += blockIdx.y;
    if (done.v[blockIdx.y]) limit = 0;
    pow_state st;
#pragma unroll
    for (int i = 0; i < 12; i++) st.s[i] = sts[blockIdx.y].s[i];
#pragma unroll 1
    for (uint64_t w = base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < limit; w += stride) {
        if (w > __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        uint64_t s[12];
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = st.s[i];
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((unsigned)i == pos) s[i] = w;
        poseidon_permute(s);
        if ((s[7] >> (64 - pow_bits)) == 0) atomicMin(best, (unsigned long long)w);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = atomicAdd(counter, 1u);
        if (ticket == gridDim.x * gridDim.y - 1) {
            __threadfence();
            best -= blockIdx.y;
            for (unsigned sg = 0; sg < gridDim.y; sg++) {
                host_out[sg] = __hip_atomic_load(best + sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(best + sg, ~0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *counter = 0;
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------ query gather
// Everything the gather needs travels in the kernel-argument segment (no descriptor / index uploads per table): level l of a tree
// with 2^log_leaves leaf digests starts at word 4 (2^(log_leaves + 1) - 2^(log_leaves - l + 1)) of its digest array
// (zkm_merkle_layout), and the query indices are at most ZKM_FRI_MAX_QUERIES words.
struct gather_oracle { const gl_t* lde; const gl_t* digests; uint32_t ncols, nsib; };
struct gather_layer { const gl_t *c0, *c1, *digests; uint32_t nsib, log_leaves; };
#define ZKM_FRI_MAX_ORACLES 8
#define ZKM_FRI_MAX_QUERIES 128
struct gather_args {
    gather_oracle o[ZKM_FRI_MAX_ORACLES];
    gather_layer l[8];
    uint32_t noracles, nlayers, arity_bits, lde_bits;
    uint64_t query_words;
    // stacked proofs (blockIdx.y = segment): oracle k's matrix / digests of segment s start lde_seg[k] / dig_seg[k] words after segment
    // s - 1's, layer l's values / digests val_seg[l] / ldig_seg[l] words
    size_t lde_seg[ZKM_FRI_MAX_ORACLES], dig_seg[ZKM_FRI_MAX_ORACLES], val_seg[8], ldig_seg[8];
};
struct gather_queries { uint32_t x[ZKM_FRI_MAX_QUERIES]; };   // x < 2^lde_bits <= 2^32
__device__ __forceinline__ size_t merkle_level_offset(unsigned log_leaves, unsigned lvl) {
    return (((size_t)2 << log_leaves) - ((size_t)2 << (log_leaves - lvl))) * 4;
}
__global__ void k_gather_queries(gather_args g, const uint32_t* __restrict__ qs /* [segment][queries] */, gl_t* __restrict__ out) {
    const size_t sg = blockIdx.y;
    uint64_t x = qs[sg * gridDim.x + blockIdx.x];
    const uint64_t N = (uint64_t)1 << g.lde_bits;
    gl_t* o = out + (sg * gridDim.x + blockIdx.x) * g.query_words;
    for (uint32_t k = 0; k < g.noracles; k++) {
        const gather_oracle& r = g.o[k];
        const gl_t* lde = r.lde + sg * g.lde_seg[k];
        const gl_t* dig = r.digests + sg * g.dig_seg[k];
        for (uint32_t c = threadIdx.x; c < r.ncols; c += blockDim.x) o[c] = lde[(size_t)c * N + x];
        o += r.ncols;
        for (uint32_t e = threadIdx.x; e < r.nsib * 4; e += blockDim.x) {
            uint32_t lvl = e >> 2;
            o[e] = dig[merkle_level_offset(g.lde_bits, lvl) + 4 * ((x >> lvl) ^ 1) + (e & 3)];
        }
        o += r.nsib * 4;
    }
    uint32_t arity = 1u << g.arity_bits;
    for (uint32_t l = 0; l < g.nlayers; l++) {
        const gather_layer& r = g.l[l];
        const gl_t *c0 = r.c0 + sg * g.val_seg[l], *c1 = r.c1 + sg * g.val_seg[l], *dig = r.digests + sg * g.ldig_seg[l];
        x >>= g.arity_bits;
        for (uint32_t e = threadIdx.x; e < 2 * arity; e += blockDim.x) o[e] = (e & 1) ? c1[x * arity + (e >> 1)] : c0[x * arity + (e >> 1)];
        o += 2 * arity;
        for (uint32_t e = threadIdx.x; e < r.nsib * 4; e += blockDim.x) {
            uint32_t lvl = e >> 2;
            o[e] = dig[merkle_level_offset(r.log_leaves, lvl) + 4 * ((x >> lvl) ^ 1) + (e & 3)];
        }
        o += r.nsib * 4;
    }
}

// ------------------------------------------------------------------ host helpers
static gl2_t challenger_get_ext(zkm_challenger* ch) {
    gl_t a = zkm_challenger_get(ch), b = zkm_challenger_get(ch);
    return gl2_t{a, b};
}

struct fri_layer {
    gl_t* values = nullptr;   // [segment][2][len] bit-reversed
    gl_t* digests = nullptr;  // [segment][dig_words]
    std::vector<size_t> level_off;
    size_t len = 0, dig_words = 0;
    unsigned log_leaves = 0;
};

// ------------------------------------------------------------------ FRI: everything after the alpha-combination
// One composite polynomial per FriBatchInfo (coefficients, F2 as two arrays), its opening point and the factor alpha^(#polys of
// the batch) of ReducingFactor::shift_poly.  fri_finish divides each by (X - point), accumulates the final
// polynomial, runs the commit phase (LDE, Merkle tree, cap -> transcript, beta, fold), the proof of work and the query rounds over
// `noracles` initial oracles.  Shared by prove_single_table (the STARK instance of stark.rs:91-148) and zkm_fri_prove (any instance).
struct fri_composite {
    const gl_t *c0, *c1;
    gl2_t point, shift;
};

static void divide_accumulate_all(zkm_ctx* c, const std::vector<std::vector<fri_composite>>& comps, size_t comp_seg, size_t n, gl_t* f0, gl_t* f1) {
    const size_t nseg = comps.size();
    if (nseg == 0 || nseg > ZKM_MAX_SEG) throw std::runtime_error("FRI: bad stack size");
    const unsigned nb = (unsigned)comps[0].size(), z = (unsigned)nseg;
    if (nb == 0 || nb > FRI_MAX_BATCHES) throw std::runtime_error("FRI: 1..8 opening batches");
    for (const auto& k : comps)
        if (k.size() != nb) throw std::runtime_error("internal: FRI instances of a stack differ");
    zkm_prof_scope ps(c, "fri_divide_linear");
    // level sizes: n, ceil(n / FRI_SEG), ... down to <= FRI_SEG
    std::vector<size_t> m{n};
    while (m.back() > FRI_SEG) m.push_back((m.back() + FRI_SEG - 1) / FRI_SEG);
    const size_t L = m.size();
    std::vector<seg_batches> lv(L);
    std::vector<gl_t*> tot(L, nullptr), suf(L, nullptr);   // tot[l]: [segment][nb][2][m[l]] totals feeding level l (l >= 1); suf[l]: suffix values of level l
    std::vector<void*> tmp;
    // points and weights of every level and segment: zw[l][s][b] = (z_b^(FRI_SEG^l), w_b), w_b = product of the shifts of the batches after b
    const size_t per_level = nseg * nb * 2;
    std::vector<gl2_t> zw(L * per_level);
    for (size_t sg = 0; sg < nseg; sg++) {
        gl2_t wacc{1, 0};
        for (unsigned b = nb; b-- > 0;) {
            gl2_t zp = comps[sg][b].point;
            for (size_t l = 0; l < L; l++) {
                zw[l * per_level + (sg * nb + b) * 2] = zp;
                zw[l * per_level + (sg * nb + b) * 2 + 1] = wacc;
                zp = gl2_pow(zp, FRI_SEG);
            }
            wacc = gl2_mul(wacc, comps[sg][b].shift);
        }
    }
    gl2_t* d_zw = (gl2_t*)c->alloc(zw.size() * sizeof(gl2_t));
    tmp.push_back(d_zw);
    c->upload(d_zw, zw.data(), zw.size() * sizeof(gl2_t));
    for (unsigned b = 0; b < nb; b++) { lv[0].a0[b] = comps[0][b].c0; lv[0].a1[b] = comps[0][b].c1; }
    lv[0].nb = nb;
    lv[0].a_seg = comp_seg;
    lv[0].zw = d_zw;
    for (size_t l = 1; l < L; l++) {
        tot[l] = (gl_t*)c->alloc(nseg * 2 * nb * m[l] * sizeof(gl_t));
        tmp.push_back(tot[l]);
        hipLaunchKernelGGL(k_seg_totals, dim3((unsigned)((m[l] + 63) / 64), nb, z), dim3(64), 0, c->stream, lv[l - 1], m[l - 1], tot[l]);
        lv[l] = lv[0];
        for (unsigned b = 0; b < nb; b++) {
            lv[l].a0[b] = tot[l] + (2 * b) * m[l];
            lv[l].a1[b] = tot[l] + (2 * b + 1) * m[l];
        }
        lv[l].a_seg = 2 * nb * m[l];
        lv[l].zw = d_zw + l * per_level;
    }
    // top-down: suffix values of each level, then the weighted sum of the bottom level's
    for (size_t l = L; l-- > 0;) {
        const size_t nsegm = (m[l] + FRI_SEG - 1) / FRI_SEG;
        const gl_t* upper = l + 1 < L ? suf[l + 1] : nullptr;
        const size_t nupper = l + 1 < L ? m[l + 1] : 0;
        if (l == 0 && n >= c->fri_fused_division_min) {
            const size_t lds = 4 * 64 * SEG_ROW * sizeof(gl_t);
            if (lds > 64 * 1024) {                                  // (only a build with longer segments needs the raised limit -- ADVICE r04)
                static std::atomic<uint64_t> lds_ok{0};             // (hipFuncSetAttribute is per device; setting it twice is harmless)
                const uint64_t bit = (uint64_t)1 << (c->device & 63);
                if (!(lds_ok.load(std::memory_order_acquire) & bit)) {
                    ZKM_HIP_CHECK(hipFuncSetAttribute((const void*)k_seg_scan_final, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    lds_ok.fetch_or(bit, std::memory_order_release);
                }
            }
            hipLaunchKernelGGL(k_seg_scan_final, dim3((unsigned)((nsegm + 63) / 64), 1, z), dim3(64), lds, c->stream, lv[0], n, upper, nupper, f0, f1);
            break;
        }
        if (l == 0 && c->fri_scan_combine) {   // bottom level: scan + weighted sum of the batches in one launch (no suffix arrays in HBM)
            // two tiles per batch: 9,216 B per batch with segments of 8 -- eight batches (FRI_MAX_BATCHES) are 73,728 B, over the 64 KB a
            // launch gets without asking; the limit is raised once per device like k_fri_fold's
            const size_t lds = (size_t)2 * nb * 64 * SEG_ROW * sizeof(gl_t);
            if (lds > 64 * 1024) {
                static std::atomic<uint64_t> sc_lds_ok{0};
                const uint64_t bit = (uint64_t)1 << (c->device & 63);
                if (!(sc_lds_ok.load(std::memory_order_acquire) & bit)) {
                    ZKM_HIP_CHECK(hipFuncSetAttribute((const void*)k_seg_scan_combine, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    sc_lds_ok.fetch_or(bit, std::memory_order_release);
                }
            }
            hipLaunchKernelGGL(k_seg_scan_combine, dim3((unsigned)((nsegm + 63) / 64), 1, z), dim3(64 * nb), lds, c->stream, lv[0], n, upper, nupper, f0,
                               f1);
            break;
        }
        suf[l] = (gl_t*)c->alloc(nseg * 2 * nb * m[l] * sizeof(gl_t));
        tmp.push_back(suf[l]);
        hipLaunchKernelGGL(k_seg_scan, dim3((unsigned)((nsegm + 63) / 64), nb, z), dim3(64), 0, c->stream, lv[l], m[l], upper, nupper, suf[l]);
        if (l == 0) hipLaunchKernelGGL(k_seg_combine, dim3((unsigned)((n + 255) / 256), 1, z), dim3(256), 0, c->stream, lv[0], n, suf[0], f0, f1);
    }
    ZKM_HIP_CHECK(hipGetLastError());
    // no host sync: released blocks are only reused by later work on this stream
    for (void* q : tmp) c->release(q);
}

// Everything of a FRI proof after the alpha-combination, for a STACK of independent proofs of the same shape (one proof: vectors of
// length 1): comps[s] = proof s's composites (pointers: proof 0's; proof s's are s * comp_seg words further), orc = the stacked initial
// oracles, chs[s] = proof s's transcript, caps_out[s] etc. its output fields.  Every stage is ONE launch (or group of launches) for all
// proofs, and every transcript round trip brings the words of all of them down together.
static void fri_finish(zkm_ctx* c, const zkm_stark_config* cfg, unsigned log_n, const std::vector<std::vector<fri_composite>>& comps, size_t comp_seg,
                       const zkm_batch* const* orc, size_t noracles, const std::vector<zkm_challenger*>& chs, unsigned L, size_t F, size_t nq,
                       size_t query_words, const std::vector<uint64_t*>& caps_out, const std::vector<uint64_t*>& final_out,
                       const std::vector<uint64_t*>& pow_out, const std::vector<uint64_t*>& queries_out) {
    if (noracles == 0 || noracles > ZKM_FRI_MAX_ORACLES) throw std::runtime_error("FRI: 1..8 initial oracles");
    if (L > 8) throw std::runtime_error("too many FRI layers");
    const size_t nseg = comps.size();
    if (nseg == 0 || nseg > ZKM_MAX_SEG || chs.size() != nseg) throw std::runtime_error("FRI: bad stack size");
    const unsigned z = (unsigned)nseg;
    const size_t n = (size_t)1 << log_n, C4 = (size_t)4 << cfg->cap_height;
    const unsigned lde_bits = log_n + cfg->rate_bits;
    const size_t N = (size_t)1 << lde_bits;
    std::vector<fri_layer> layers(L);
    std::vector<void*> scratch;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        for (auto& l : layers) { c->release(l.values); c->release(l.digests); }
        for (void* p : scratch) c->release(p);
    };
    try {
        gl_t* d_fin = (gl_t*)c->alloc(nseg * 2 * n * sizeof(gl_t));  // final poly coefficients [segment][2][n]
        scratch.push_back(d_fin);
        divide_accumulate_all(c, comps, comp_seg, n, d_fin, d_fin + n);

        // commit phase: coefficients stay in d_fin (length clen, implicitly zero-padded x4)
        size_t clen = n;
        unsigned clog = log_n;
        gl_t shift = GL_GENERATOR;
        unsigned arity = 1u << cfg->arity_bits;
        gl_t* d_coef = d_fin;       // [segment][2][clen]: c0 array, then c1
        std::vector<uint64_t> capbuf(nseg * C4);
        for (unsigned l = 0; l < L; l++) {
            fri_layer& fl = layers[l];
            fl.len = clen << cfg->rate_bits;
            fl.values = (gl_t*)c->alloc(nseg * 2 * fl.len * sizeof(gl_t));
            // values = coset_fft(shift) of the zero-padded coefficients, bit-reversed: two base-field columns per proof
            // (the coefficient arrays are contiguous: [c0 | c1] per proof, column stride clen)
            zkm_lde_bitrev(c, d_coef, fl.values, 2 * nseg, clog, cfg->rate_bits, shift);
            fl.log_leaves = clog + cfg->rate_bits - cfg->arity_bits;
            size_t dwords = zkm_merkle_layout(fl.log_leaves, cfg->cap_height, fl.level_off);
            fl.dig_words = dwords;
            fl.digests = (gl_t*)c->alloc(nseg * dwords * sizeof(gl_t));
            zkm_launch_merkle_leaves_ext(c, fl.values, fl.values + fl.len, (size_t)1 << fl.log_leaves, arity, fl.digests, nseg, 2 * fl.len, dwords);
            zkm_merkle_build_inner_cap(c, fl.digests, fl.level_off, fl.log_leaves, cfg->cap_height, capbuf.data(), nseg, dwords);
            seg_gl2 betas{};
            for (size_t sg = 0; sg < nseg; sg++) {
                uint64_t* capo = caps_out[sg] + l * C4;
                memcpy(capo, capbuf.data() + sg * C4, C4 * sizeof(uint64_t));
                zkm_challenger_observe(chs[sg], capo, C4);
                gl2_t beta = challenger_get_ext(chs[sg]);
                betas.v[2 * sg] = beta.c0; betas.v[2 * sg + 1] = beta.c1;
            }
            size_t nout = clen >> cfg->arity_bits;
            gl_t* d_new = (gl_t*)c->alloc(nseg * 2 * nout * sizeof(gl_t));
            scratch.push_back(d_new);
            {
                zkm_prof_scope ps(c, "fri_fold");
                const size_t lds = 2 * 64 * ((size_t)arity + 1) * sizeof(gl_t);
                if (lds > 64 * 1024) {
                    static std::atomic<uint64_t> lds_ok{0};
                    const uint64_t bit = (uint64_t)1 << (c->device & 63);
                    if (!(lds_ok.load(std::memory_order_acquire) & bit)) {
                        ZKM_HIP_CHECK(hipFuncSetAttribute((const void*)k_fri_fold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                        lds_ok.fetch_or(bit, std::memory_order_release);
                    }
                }
                hipLaunchKernelGGL(k_fri_fold, dim3((unsigned)((nout + 63) / 64), 1, z), dim3(64), lds, c->stream, d_coef, d_coef + clen, nout, arity, betas,
                                   d_new, d_new + nout);
                ZKM_HIP_CHECK(hipGetLastError());
            }
            d_coef = d_new;
            clen = nout;
            clog -= cfg->arity_bits;
            shift = gl_pow(shift, arity);
        }
        if (clen != F) throw std::runtime_error("internal: final polynomial length mismatch");
        {
            std::vector<gl_t> f(nseg * 2 * clen);
            c->download(f.data(), d_coef, nseg * 2 * clen * 8);
            for (size_t sg = 0; sg < nseg; sg++) {
                uint64_t* fp = final_out[sg];
                const gl_t* fs = f.data() + sg * 2 * clen;
                for (size_t i = 0; i < clen; i++) { fp[2 * i] = fs[i]; fp[2 * i + 1] = fs[clen + i]; }
                zkm_challenger_observe(chs[sg], fp, 2 * clen);
            }
        }

        // proof of work (App. A.9), smallest witness of every proof
        {
            std::vector<pow_state> sts(nseg);
            seg_u32 pos{}, done{};
            for (size_t sg = 0; sg < nseg; sg++) {
                memcpy(sts[sg].s, chs[sg]->state, sizeof sts[sg].s);
                for (uint32_t i = 0; i < chs[sg]->n_in; i++) sts[sg].s[i] = chs[sg]->in_buf[i];
                pos.v[sg] = chs[sg]->n_in;
            }
            pow_state* d_sts = (pow_state*)c->alloc(nseg * sizeof(pow_state));
            scratch.push_back(d_sts);
            c->upload(d_sts, sts.data(), nseg * sizeof(pow_state));
            unsigned long long* d_best = c->pow_best();      // ZKM_MAX_SEG words, all ones between searches (the kernel leaves them that way)
            std::vector<unsigned long long> best(nseg, ~0ULL), got(nseg);
            // 2^pow_round_log candidates per round and proof (17 for a proof alone: two waves per SIMD -- a wave alone issues at half the
            // SIMD's rate, so the round is barely longer than with one, and it holds the hit with probability 0.86 instead of 0.63 at 16
            // bits; the searches of a stack share the machine: the round of each shrinks with their number, down to 2^12, so that the
            // stack's round is still one launch-filling set of waves and few candidates are tried beyond the round of a proof's first hit),
            // up to 2^6 rounds per launch (a launch without a hit: probability e^-128 for a proof alone)
            unsigned round_log = c->pow_round_log;
            for (size_t k = 1; k < nseg && round_log > 12; k <<= 1) round_log--;
            const uint64_t stride = (uint64_t)1 << round_log, span = stride << 6;
            size_t open = nseg;
            for (uint64_t base = 0; open; base += span) {
                if (base > ((uint64_t)1 << 40)) throw std::runtime_error("Proof of work failed. This is highly unlikely!");
                uint64_t *host_slot, *flag;
                unsigned* counter;
                const uint64_t seq = c->xfer_begin(8 * nseg, &host_slot, &flag, &counter);
                {
                    zkm_prof_scope ps(c, "fri_pow_search");
                    hipLaunchKernelGGL(k_pow_search, dim3(stride / 256, z), dim3(256), 0, c->stream, d_sts, pos, done, cfg->pow_bits, base, base + span,
                                       d_best, counter, host_slot, flag, seq);
                    ZKM_HIP_CHECK(hipGetLastError());
                }
                c->xfer_finish(seq, got.data(), 8 * nseg);
                for (size_t sg = 0; sg < nseg; sg++)
                    if (!done.v[sg] && got[sg] != ~0ULL) { best[sg] = got[sg]; done.v[sg] = 1; open--; }
            }
            for (size_t sg = 0; sg < nseg; sg++) {
                uint64_t w = best[sg];
                *pow_out[sg] = w;
                zkm_challenger_observe(chs[sg], &w, 1);
                uint64_t resp = zkm_challenger_get(chs[sg]);
                if ((resp >> (64 - cfg->pow_bits)) != 0) throw std::runtime_error("internal: proof-of-work response check failed");
            }
        }

        // query rounds
        {
            if (nq > ZKM_FRI_MAX_QUERIES || lde_bits > 32) throw std::runtime_error("FRI: at most 128 query rounds on domains of at most 2^32 points");
            std::vector<uint32_t> qs(nseg * nq);
            for (size_t sg = 0; sg < nseg; sg++)
                for (size_t q = 0; q < nq; q++) qs[sg * nq + q] = (uint32_t)(zkm_challenger_get(chs[sg]) % N);
            uint32_t* d_qs = (uint32_t*)c->alloc(qs.size() * sizeof(uint32_t));
            scratch.push_back(d_qs);
            c->upload(d_qs, qs.data(), qs.size() * sizeof(uint32_t));
            gather_args ga{};
            ga.noracles = (uint32_t)noracles;
            for (size_t k = 0; k < noracles; k++) {
                if (orc[k]->lde_bits() != lde_bits || orc[k]->nseg != nseg) throw std::runtime_error("internal: oracle of another domain size in the query gather");
                ga.o[k].lde = orc[k]->lde; ga.o[k].digests = orc[k]->digests; ga.o[k].ncols = (uint32_t)orc[k]->ncols;
                ga.o[k].nsib = lde_bits - cfg->cap_height;
                ga.lde_seg[k] = orc[k]->lde_seg(); ga.dig_seg[k] = orc[k]->dig_words;
            }
            for (unsigned l = 0; l < L; l++) {
                ga.l[l].c0 = layers[l].values; ga.l[l].c1 = layers[l].values + layers[l].len; ga.l[l].digests = layers[l].digests;
                ga.l[l].nsib = layers[l].log_leaves - cfg->cap_height;
                ga.l[l].log_leaves = layers[l].log_leaves;
                ga.val_seg[l] = 2 * layers[l].len; ga.ldig_seg[l] = layers[l].dig_words;
            }
            ga.nlayers = L; ga.arity_bits = cfg->arity_bits; ga.lde_bits = lde_bits; ga.query_words = query_words;
            const size_t qwords = nq * query_words;
            gl_t* d_q = (gl_t*)c->alloc(nseg * qwords * 8);
            scratch.push_back(d_q);
            {
                zkm_prof_scope ps(c, "fri_gather_queries");
                hipLaunchKernelGGL(k_gather_queries, dim3(nq, z), dim3(256), 0, c->stream, ga, d_qs, d_q);
                ZKM_HIP_CHECK(hipGetLastError());
            }
            if (nseg == 1) {
                c->download(queries_out[0], d_q, qwords * 8);
            } else {
                std::vector<gl_t> all(nseg * qwords);
                c->download(all.data(), d_q, nseg * qwords * 8);
                for (size_t sg = 0; sg < nseg; sg++) memcpy(queries_out[sg], all.data() + sg * qwords, qwords * 8);
            }
        }

    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

// prove_single_table for a STACK of nseg = chs.size() independent proofs of the same table at the same height (one proof: vectors of
// length 1, the single-table entry points): the commitments are stacked batches (zkm_internal.h), zs = nseg lists of CtlZData,
// lookup_challenges = nseg x num_challenges, chs[s] / proofs[s] = proof s's transcript and output blob.  Every stage is one launch (or
// group of launches) for all proofs; the transcripts advance side by side on the host between the round trips.
static void prove_single_table(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                               const zkm_batch* trace_batch, const uint64_t* aux, size_t A_ctl, const zkm_ctl_table* ctl_table,
                               const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t Z, const uint64_t* lookup_challenges,
                               const std::vector<zkm_challenger*>& chs, const std::vector<uint64_t*>& proofs, const zkm_batch* aux_batch_in = nullptr,
                               const zkm_batch* quot_batch_in = nullptr) {
    // openings-only mode (zkm_prove_openings, BASELINE config 4): the three commitments exist already; the transcript
    // is compact -> zeta -> openings -> prove_openings.  aux_given (zkm_prove_with_traces): the auxiliary commitment was built ahead of
    // the transcript (it only depends on the CTL challenges); everything from observing its cap on runs here.
    const bool openings_only = quot_batch_in != nullptr;
    const bool aux_given = aux_batch_in != nullptr && !openings_only;
    const size_t nseg = chs.size();
    if (nseg == 0 || nseg > ZKM_MAX_SEG || proofs.size() != nseg) throw std::runtime_error("prove_single_table: bad stack size");
    if (nseg > 1 && !(aux_given && trace_batch)) throw std::runtime_error("internal: a stack of proofs needs its stacked commitments");
    validate_config(cfg, log_n);  // before zkm_num_lookup_columns reads it; make_layout checks again
    // the table's own lookup helper columns come first among the auxiliary polynomials (prover.rs:467-508)
    const size_t NL = openings_only ? 0 : zkm_num_lookup_columns(table_id, cfg);
    if (NL && !lookup_challenges) throw std::runtime_error("this table has lookups: lookup challenges are required");
    if (NL && !trace && !aux_given) throw std::runtime_error("this table has lookups: the trace values are required to build their helper columns");
    if (!NL) lookup_challenges = nullptr;
    const size_t A = NL + A_ctl;
    proof_layout y;
    make_layout(y, cfg, log_n, W, A, Z);
    if (y.L > 8) throw std::runtime_error("too many FRI layers");
    size_t n = (size_t)1 << log_n;
    if (openings_only) {
        if (Z > A) throw std::runtime_error("zkm_prove_openings: more CTL Zs than auxiliary polynomials");
    } else {
        size_t ctl_helpers = 0;
        for (size_t i = 0; i < Z; i++) ctl_helpers += zs[i].num_helpers;
        if (ctl_helpers + Z != A_ctl) throw std::runtime_error("No CTL? aux column count does not match the CTL description");
    }
    if (A_ctl == 0) throw std::runtime_error("No CTL? aux column count does not match the CTL description");  // prover.rs:509
    const size_t total_helpers = A - Z;  // index of the first CTL Z among the auxiliary polynomials
    ctl_dev_owner own;
    if (!openings_only) own.upload(c, ctl_table, zs, colset_ids, Z, false, W, nseg);
    if (ctl_table && !openings_only)
        for (size_t i = 0; i < ctl_table->nterms; i++)
            if (ctl_table->term_col[i] >= W) throw std::runtime_error("CTL description: trace column index out of range");

    // (the query rounds -- nine tenths of the blob, 2 MB for the Keccak table -- are written in full by the download at the end of
    // fri_finish: zeroing them here was 0.2 ms of host time per segment in front of the table's first launch)
    for (uint64_t* proof : proofs) {
        memset(proof, 0, y.o_queries * sizeof(uint64_t));
        proof[0] = ZKM_PROOF_MAGIC; proof[1] = log_n; proof[2] = W; proof[3] = A; proof[4] = y.Q; proof[5] = Z; proof[6] = y.cap;
        proof[7] = y.L; proof[8] = y.F; proof[9] = y.nq; proof[10] = cfg->rate_bits; proof[11] = cfg->arity_bits;
    }

    zkm_batch* own_trace = nullptr;
    zkm_batch *ab = nullptr, *qb = nullptr;
    std::vector<void*> scratch;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        zkm_batch_free(own_trace);
        zkm_batch_free(ab);
        zkm_batch_free(qb);
        for (void* p : scratch) c->release(p);
    };
    try {
        const zkm_batch* tb = trace_batch;
        if (!tb) {
            own_trace = new zkm_batch();
            own_trace->ctx = c; own_trace->ncols = W; own_trace->log_n = log_n; own_trace->rate_bits = cfg->rate_bits;
            own_trace->cap_height = cfg->cap_height;
            zkm_prof_scope st(c, "stage/compute trace commitment");  // prover.rs:146-163 (done by the caller in the reference)
            zkm_batch_build(own_trace, trace, true);
            tb = own_trace;
        }
        if (tb->ncols != W || tb->log_n != log_n || tb->rate_bits != cfg->rate_bits || tb->cap_height != cfg->cap_height || tb->nseg != nseg)
            throw std::runtime_error("trace commitment does not match the table shape / config");

        const size_t C4 = y.C * 4;
        for (size_t sg = 0; sg < nseg; sg++) {
            zkm_challenger_compact(chs[sg], proofs[sg] + y.o_init);  // :466
            memcpy(proofs[sg] + y.o_caps, tb->cap.data() + sg * C4, C4 * 8);
        }
        const zkm_batch *abp = aux_batch_in, *qbp = quot_batch_in;
        if (aux_given) {
            if (abp->ncols != A || abp->log_n != log_n || abp->rate_bits != cfg->rate_bits || abp->cap_height != cfg->cap_height || abp->nseg != nseg)
                throw std::runtime_error("auxiliary commitment does not match the table shape / config");
        } else if (!openings_only) {
            // auxiliary commitment :511-522
            ab = new zkm_batch();
            ab->ctx = c; ab->ncols = A; ab->log_n = log_n; ab->rate_bits = cfg->rate_bits; ab->cap_height = cfg->cap_height;
            if (NL) {
                // "compute lookup helper columns" :475-493, then the CTL columns behind them
                auto st = std::make_unique<zkm_prof_scope>(c, "stage/compute lookup helper columns");
                gl_t* d_all = (gl_t*)c->alloc(A * n * sizeof(gl_t));
                scratch.push_back(d_all);
                const gl_t* d_trace = trace;
                if (!zkm_is_device_ptr(trace)) {
                    gl_t* d = (gl_t*)c->alloc(W * n * sizeof(gl_t));
                    scratch.push_back(d);
                    c->upload(d, trace, W * n * sizeof(gl_t));
                    zkm_launch_canon(c, d, W * n);   // (host words may be any representative; the lookup kernels want canonical ones)
                    d_trace = d;
                }
                zkm_table_lookup_columns_device(c, table_id, lookup_challenges, cfg->num_challenges, d_trace, n, d_all);
                ZKM_HIP_CHECK(hipMemcpyAsync(d_all + NL * n, aux, A_ctl * n * sizeof(gl_t), hipMemcpyDefault, c->stream));
                st.reset();
                zkm_prof_scope st2(c, "stage/compute auxiliary polynomials commitment");  // :511-522
                zkm_batch_build(ab, d_all, true);
            } else {
                zkm_prof_scope st2(c, "stage/compute auxiliary polynomials commitment");
                zkm_batch_build(ab, aux, true);
            }
            abp = ab;
        }
        if (!openings_only) {
            std::vector<gl_t> alphas(nseg * cfg->num_challenges);
            for (size_t sg = 0; sg < nseg; sg++) {
                uint64_t* caps = proofs[sg] + y.o_caps;
                memcpy(caps + C4, abp->cap.data() + sg * C4, C4 * 8);
                zkm_challenger_observe(chs[sg], caps + C4, C4);  // :525
                for (unsigned i = 0; i < cfg->num_challenges; i++) alphas[sg * cfg->num_challenges + i] = zkm_challenger_get(chs[sg]);  // :527
            }

            // quotient :543-587
            gl_t* d_quot = (gl_t*)c->alloc(nseg * cfg->num_challenges * 2 * n * sizeof(gl_t));
            scratch.push_back(d_quot);
            {
                zkm_prof_scope st(c, "stage/compute quotient polys");  // :543-559
                quotient_device(c, table_id, tb, abp, own, lookup_challenges, alphas.data(), cfg->num_challenges, d_quot);
            }
            qb = new zkm_batch();
            qb->ctx = c; qb->ncols = y.Q; qb->nseg = nseg; qb->log_n = log_n; qb->rate_bits = cfg->rate_bits; qb->cap_height = cfg->cap_height;
            {
                zkm_prof_scope st(c, "stage/compute quotient commitment");  // :576-587
                zkm_batch_build(qb, d_quot, false);  // chunks [q0_lo, q0_hi, q1_lo, q1_hi] == d_quot viewed as Q columns of n, proof after proof
            }
            for (size_t sg = 0; sg < nseg; sg++) {
                uint64_t* caps = proofs[sg] + y.o_caps;
                memcpy(caps + 2 * C4, qb->cap.data() + sg * C4, C4 * 8);
                zkm_challenger_observe(chs[sg], caps + 2 * C4, C4);  // :589
            }
            qbp = qb;
        } else {
            for (const zkm_batch* b : {abp, qbp})
                if (!b || b->log_n != log_n || b->rate_bits != cfg->rate_bits || b->cap_height != cfg->cap_height || b->nseg != nseg)
                    throw std::runtime_error("zkm_prove_openings: commitments do not match the table shape / config");
            if (abp->ncols != A || qbp->ncols != y.Q) throw std::runtime_error("zkm_prove_openings: unexpected number of polynomials");
            for (size_t sg = 0; sg < nseg; sg++) {
                uint64_t* caps = proofs[sg] + y.o_caps;
                memcpy(caps + C4, abp->cap.data() + sg * C4, C4 * 8);
                memcpy(caps + 2 * C4, qbp->cap.data() + sg * C4, C4 * 8);
            }
        }

        std::vector<gl2_t> zeta(nseg), zeta_next(nseg);
        gl_t g = gl_root_of_unity(log_n);
        for (size_t sg = 0; sg < nseg; sg++) {
            zeta[sg] = challenger_get_ext(chs[sg]);  // :591
            if (gl2_eq(gl2_exp_pow2(zeta[sg], log_n), gl2_t{1, 0})) throw std::runtime_error("Opening point is in the subgroup.");  // :596-599
            zeta_next[sg] = gl2_scalar_mul(zeta[sg], g);
        }

        // openings proof.rs:299-334
        {
            zkm_prof_scope st(c, "stage/openings (StarkOpeningSet::new)");  // proof.rs:299-334, between two timed! scopes in the reference
            auto ev = eval_batches(c, {tb, abp, qbp}, zeta.data(), zeta_next.data());
            for (size_t sg = 0; sg < nseg; sg++) {
                uint64_t* op = proofs[sg] + y.o_open;
                uint64_t *o_local = op, *o_next = op + 2 * W, *o_aux = op + 4 * W, *o_auxn = o_aux + 2 * A, *o_ctl = o_auxn + 2 * A, *o_quot = o_ctl + Z;
                const auto &tv = ev[sg][0], &av = ev[sg][1], &qv = ev[sg][2];
                for (size_t i = 0; i < W; i++) {
                    o_local[2 * i] = tv[i].at_z0.c0; o_local[2 * i + 1] = tv[i].at_z0.c1;
                    o_next[2 * i] = tv[i].at_z1.c0; o_next[2 * i + 1] = tv[i].at_z1.c1;
                }
                for (size_t i = 0; i < A; i++) {
                    o_aux[2 * i] = av[i].at_z0.c0; o_aux[2 * i + 1] = av[i].at_z0.c1;
                    o_auxn[2 * i] = av[i].at_z1.c0; o_auxn[2 * i + 1] = av[i].at_z1.c1;
                    if (i >= total_helpers) o_ctl[i - total_helpers] = av[i].at_one;
                }
                for (size_t i = 0; i < y.Q; i++) { o_quot[2 * i] = qv[i].at_z0.c0; o_quot[2 * i + 1] = qv[i].at_z0.c1; }
            }
        }
        // ---- prove_openings (App. A.8)
        zkm_prof_scope st_fri(c, "stage/compute openings proof");  // prover.rs:618-628
        const size_t np0 = W + A + y.Q, np1 = W + A, np2 = Z, apow_seg = 2 * (np0 + 1);
        std::vector<gl_t> apow(nseg * apow_seg);
        for (size_t sg = 0; sg < nseg; sg++) {
            zkm_challenger* ch = chs[sg];
            uint64_t* op = proofs[sg] + y.o_open;
            uint64_t *o_local = op, *o_next = op + 2 * W, *o_aux = op + 4 * W, *o_auxn = o_aux + 2 * A, *o_ctl = o_auxn + 2 * A, *o_quot = o_ctl + Z;
            // observe_openings(to_fri_openings) proof.rs:336-367
            zkm_challenger_observe(ch, o_local, 2 * W);
            zkm_challenger_observe(ch, o_aux, 2 * A);
            zkm_challenger_observe(ch, o_quot, 2 * y.Q);
            zkm_challenger_observe(ch, o_next, 2 * W);
            zkm_challenger_observe(ch, o_auxn, 2 * A);
            for (size_t i = 0; i < Z; i++) { uint64_t e[2] = {o_ctl[i], 0}; zkm_challenger_observe(ch, e, 2); }
            gl2_t alpha = challenger_get_ext(ch);
            gl2_t p{1, 0};
            gl_t* ap = apow.data() + sg * apow_seg;
            for (size_t j = 0; j <= np0; j++) { ap[2 * j] = p.c0; ap[2 * j + 1] = p.c1; p = gl2_mul(p, alpha); }
        }
        gl_t* d_apow = (gl_t*)c->alloc(apow.size() * sizeof(gl_t));
        scratch.push_back(d_apow);
        c->upload(d_apow, apow.data(), apow.size() * sizeof(gl_t));
        gl_t* d_comp = (gl_t*)c->alloc(nseg * 6 * n * sizeof(gl_t));
        scratch.push_back(d_comp);
        if (tb->coeff_s1 != abp->coeff_s1 || tb->coeff_s1 != qbp->coeff_s1) throw std::runtime_error("internal: coefficient layouts of the three oracles differ");
        {
            zkm_prof_scope ps(c, "fri_combine");
            const size_t npoly = W + A + y.Q;
            const unsigned z = (unsigned)nseg;
            // slices of >= 32 polynomials while the launch stays below ~2^17 threads (two waves per SIMD)
            size_t slices = npoly / 32 < 1 ? 1 : npoly / 32;
            while (slices > 1 && slices * n * nseg > ((size_t)1 << 17)) slices >>= 1;
            if (slices >= 4) {
                const size_t per = (npoly + slices - 1) / slices;
                slices = (npoly + per - 1) / per;
                gl_t* d_part = (gl_t*)c->alloc(nseg * slices * 6 * n * sizeof(gl_t));
                scratch.push_back(d_part);
                hipLaunchKernelGGL(k_fri_combine_slice, dim3((n + 255) / 256, (unsigned)slices, z), dim3(256), 0, c->stream, tb->coeffs, W, abp->coeffs, A,
                                   qbp->coeffs, y.Q, total_helpers, d_apow, n, per, d_part, apow_seg);
                hipLaunchKernelGGL(k_fri_combine_sum, dim3((n + 255) / 256, 1, z), dim3(256), 0, c->stream, d_part, (unsigned)slices, n, d_comp);
            } else {
                hipLaunchKernelGGL(k_fri_combine, dim3((n + 255) / 256, 1, z), dim3(256), 0, c->stream, tb->coeffs, W, abp->coeffs, A, qbp->coeffs, y.Q,
                                   total_helpers, d_apow, n, d_comp, apow_seg);
            }
            ZKM_HIP_CHECK(hipGetLastError());
        }
        if (tb->coeff_s1) {
            // the combination is position-wise, so the six composite arrays come out in the batches' coefficient layout; the division by
            // (X - point) walks exponents in order
            gl_t* d_nat = (gl_t*)c->alloc(nseg * 6 * n * sizeof(gl_t));
            scratch.push_back(d_nat);
            zkm_coeff_layout_convert(c, d_comp, n, d_nat, n, 6 * nseg, log_n, /*to_natural=*/true);
            d_comp = d_nat;
        }
        // divide by (X - point), accumulate, commit phase, proof of work, query rounds: shared with zkm_fri_prove
        {
            std::vector<std::vector<fri_composite>> comps(nseg);
            std::vector<uint64_t*> caps_o(nseg), final_o(nseg), pow_o(nseg), queries_o(nseg);
            for (size_t sg = 0; sg < nseg; sg++) {
                const gl_t* ap = apow.data() + sg * apow_seg;
                comps[sg] = {{d_comp, d_comp + n, zeta[sg], gl2_t{ap[2 * np0], ap[2 * np0 + 1]}},
                             {d_comp + 2 * n, d_comp + 3 * n, zeta_next[sg], gl2_t{ap[2 * np1], ap[2 * np1 + 1]}},
                             {d_comp + 4 * n, d_comp + 5 * n, gl2_t{1, 0}, gl2_t{ap[2 * np2], ap[2 * np2 + 1]}}};
                caps_o[sg] = proofs[sg] + y.o_fri_caps; final_o[sg] = proofs[sg] + y.o_final; pow_o[sg] = proofs[sg] + y.o_pow;
                queries_o[sg] = proofs[sg] + y.o_queries;
            }
            const zkm_batch* orc[3] = {tb, abp, qbp};
            fri_finish(c, cfg, log_n, comps, 6 * n, orc, 3, chs, y.L, y.F, y.nq, y.query_words, caps_o, final_o, pow_o, queries_o);
        }
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

// prove_single_table on existing (stacked) trace AND auxiliary commitments (zkm_prove_with_traces builds the auxiliary commitments of all
// tables ahead of the transcript, side by side); zs = chs.size() lists of CtlZData, lookup_challenges = chs.size() x num_challenges; throws
void zkm_prove_single_table_aux(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, size_t ncols, unsigned log_n, const zkm_batch* trace_batch,
                                const zkm_batch* aux_batch, size_t naux_ctl, const zkm_ctl_table* table, const zkm_ctl_z* zs,
                                const uint32_t* colset_ids, size_t nzs, const uint64_t* lookup_challenges, const std::vector<zkm_challenger*>& chs,
                                const std::vector<uint64_t*>& proofs) {
    if (!trace_batch || !aux_batch) throw std::runtime_error("prove_single_table: commitments are required");
    prove_single_table(c, table_id, cfg, nullptr, ncols, log_n, trace_batch, nullptr, naux_ctl, table, zs, colset_ids, nzs, lookup_challenges, chs,
                       proofs, aux_batch, nullptr);
}

// ------------------------------------------------------------------ C ABI
static int fail(char** err, const std::string& msg) {
    if (err) {
        *err = (char*)malloc(msg.size() + 1);
        if (*err) memcpy(*err, msg.c_str(), msg.size() + 1);
    }
    return 1;
}

// ---- PolynomialBatch::prove_openings for an arbitrary FriInstanceInfo (plonky2 fri/oracle.rs; instance of the STARKs: stark.rs:91-148)
// composite of one batch: sum_j alpha^j p_j, polynomials named by pointer (any oracle, any column)
__global__ __launch_bounds__(256) void k_fri_combine_generic(const gl_t* const* __restrict__ polys, size_t npolys, const gl_t* __restrict__ apow,
                                                             size_t n, gl_t* __restrict__ c0, gl_t* __restrict__ c1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl_t a0 = 0, a1 = 0;
    for (size_t j = 0; j < npolys; j++) {
        gl_t v = polys[j][i];
        a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
        a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
    }
    c0[i] = a0;
    c1[i] = a1;
}

struct fri_blob_layout {
    unsigned L;
    size_t F, C4, o_caps, o_final, o_pow, o_queries, query_words, total;
};
static void fri_blob_make(fri_blob_layout& y, const zkm_stark_config* cfg, unsigned log_n, const size_t* cols, size_t noracles) {
    validate_config(cfg, log_n);
    if (noracles == 0 || noracles > ZKM_FRI_MAX_ORACLES) throw std::runtime_error("zkm_fri_prove: 1..8 oracles");
    y.L = fri_num_layers(cfg, log_n);
    y.F = (size_t)1 << (log_n - y.L * cfg->arity_bits);
    y.C4 = (size_t)4 << cfg->cap_height;
    const unsigned lde_bits = log_n + cfg->rate_bits;
    size_t o = 24;
    y.o_caps = o; o += y.L * y.C4;
    y.o_final = o; o += 2 * y.F;
    y.o_pow = o; o += 1;
    y.o_queries = o;
    size_t q = 0;
    for (size_t k = 0; k < noracles; k++) q += cols[k] + (size_t)(lde_bits - cfg->cap_height) * 4;
    for (unsigned i = 0; i < y.L; i++) q += 2 * ((size_t)1 << cfg->arity_bits) + (size_t)(lde_bits - cfg->arity_bits * (i + 1) - cfg->cap_height) * 4;
    y.query_words = q;
    y.total = o + q * cfg->num_queries;
}

// parity / debug: a b mod p through THIS translation unit's gl_mul_loose (GL_REDUCE_BRANCHFREE), for zkm_field_selftest
__global__ void k_mul_selftest_branchfree(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, size_t n, uint64_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl_canon(gl_mul_loose(a[i], b[i]));
}
void zkm_launch_mul_selftest_branchfree(zkm_ctx* c, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    hipLaunchKernelGGL(k_mul_selftest_branchfree, dim3((n + 255) / 256), dim3(256), 0, c->stream, a, b, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}

extern "C" {

size_t zkm_fri_proof_words(const zkm_stark_config* cfg, unsigned log_n, const size_t* oracle_cols, size_t noracles) {
    try {
        fri_blob_layout y;
        fri_blob_make(y, cfg, log_n, oracle_cols, noracles);
        return y.total;
    } catch (...) {
        return 0;
    }
}

int zkm_fri_prove(zkm_ctx* c, const zkm_stark_config* cfg, const zkm_batch* const* oracles, size_t noracles, const zkm_fri_batch* batches,
                  size_t nbatches, zkm_challenger* ch_io, uint64_t* proof, char** err) {
    std::vector<void*> scratch;
    if (!c || !cfg || !oracles || !batches || !nbatches || !ch_io || !proof) return fail(err, "zkm_fri_prove: null argument");
    // The transcript is advanced on a copy and handed back only when the whole proof exists: a failed call leaves the caller's
    // challenger where it was (the reference's prove_openings cannot fail half way; a C ABI call can).
    zkm_challenger local = *ch_io;
    zkm_challenger* const ch = &local;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (noracles == 0 || noracles > ZKM_FRI_MAX_ORACLES) throw std::runtime_error("zkm_fri_prove: 1..8 oracles");
        const unsigned log_n = oracles[0]->log_n;
        size_t cols[ZKM_FRI_MAX_ORACLES];
        for (size_t k = 0; k < noracles; k++) {
            if (!oracles[k] || oracles[k]->log_n != log_n || oracles[k]->rate_bits != cfg->rate_bits || oracles[k]->cap_height != cfg->cap_height ||
                oracles[k]->coeff_s1 != oracles[0]->coeff_s1)
                throw std::runtime_error("zkm_fri_prove: oracles must share degree, rate and cap height with the config");
            cols[k] = oracles[k]->ncols;
        }
        fri_blob_layout y;
        fri_blob_make(y, cfg, log_n, cols, noracles);
        const size_t n = (size_t)1 << log_n;
        size_t maxp = 0;
        for (size_t b = 0; b < nbatches; b++) {
            if (batches[b].npolys == 0 || !batches[b].polys) throw std::runtime_error("zkm_fri_prove: empty batch");
            if (batches[b].point[0] >= GL_P || batches[b].point[1] >= GL_P) throw std::runtime_error("zkm_fri_prove: non-canonical opening point");
            for (size_t j = 0; j < batches[b].npolys; j++)
                if (batches[b].polys[j].oracle >= noracles || batches[b].polys[j].poly >= cols[batches[b].polys[j].oracle])
                    throw std::runtime_error("zkm_fri_prove: polynomial index out of range");
            maxp = std::max(maxp, batches[b].npolys);
        }
        memset(proof, 0, y.total * sizeof(uint64_t));
        proof[0] = ZKM_FRI_PROOF_MAGIC; proof[1] = log_n; proof[2] = noracles; proof[3] = cfg->cap_height; proof[4] = y.L; proof[5] = y.F;
        proof[6] = cfg->num_queries; proof[7] = cfg->rate_bits; proof[8] = cfg->arity_bits;
        for (size_t k = 0; k < noracles; k++) proof[16 + k] = cols[k];

        gl2_t alpha = challenger_get_ext(ch);
        std::vector<gl_t> apow(2 * (maxp + 1));
        {
            gl2_t pw{1, 0};
            for (size_t j = 0; j <= maxp; j++) { apow[2 * j] = pw.c0; apow[2 * j + 1] = pw.c1; pw = gl2_mul(pw, alpha); }
        }
        gl_t* d_apow = (gl_t*)c->alloc(apow.size() * sizeof(gl_t));
        scratch.push_back(d_apow);
        c->upload(d_apow, apow.data(), apow.size() * sizeof(gl_t));
        gl_t* d_comp = (gl_t*)c->alloc(2 * nbatches * n * sizeof(gl_t));
        scratch.push_back(d_comp);
        std::vector<const gl_t*> ptrs;
        std::vector<size_t> first(nbatches);
        for (size_t b = 0; b < nbatches; b++) {
            first[b] = ptrs.size();
            for (size_t j = 0; j < batches[b].npolys; j++) ptrs.push_back(oracles[batches[b].polys[j].oracle]->coeffs + (size_t)batches[b].polys[j].poly * n);
        }
        const gl_t** d_ptrs = (const gl_t**)c->alloc(ptrs.size() * sizeof(gl_t*));
        scratch.push_back((void*)d_ptrs);
        c->upload((void*)d_ptrs, ptrs.data(), ptrs.size() * sizeof(gl_t*));
        std::vector<std::vector<fri_composite>> comps_v(1);
        std::vector<fri_composite>& comps = comps_v[0];
        for (size_t b = 0; b < nbatches; b++) {
            zkm_prof_scope ps(c, "fri_combine");
            hipLaunchKernelGGL(k_fri_combine_generic, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_ptrs + first[b], batches[b].npolys, d_apow, n,
                               d_comp + 2 * b * n, d_comp + (2 * b + 1) * n);
            ZKM_HIP_CHECK(hipGetLastError());
            const size_t np = batches[b].npolys;
            comps.push_back({d_comp + 2 * b * n, d_comp + (2 * b + 1) * n, gl2_t{batches[b].point[0], batches[b].point[1]},
                             gl2_t{apow[2 * np], apow[2 * np + 1]}});
        }
        if (oracles[0]->coeff_s1) {   // (the layout is a function of the height: the same for every oracle; see prove_single_table)
            gl_t* d_nat = (gl_t*)c->alloc(2 * nbatches * n * sizeof(gl_t));
            scratch.push_back(d_nat);
            zkm_coeff_layout_convert(c, d_comp, n, d_nat, n, 2 * nbatches, log_n, /*to_natural=*/true);
            for (auto& k : comps) { k.c0 = d_nat + (k.c0 - d_comp); k.c1 = d_nat + (k.c1 - d_comp); }
        }
        ZKM_HIP_CHECK(hipStreamSynchronize(c->stream));  // host vectors (ptrs, apow) are consumed
        for (size_t k = 0; k < noracles; k++)
            if (oracles[k]->nseg != 1) throw std::runtime_error("zkm_fri_prove: stacked batches are internal");
        fri_finish(c, cfg, log_n, comps_v, 0, oracles, noracles, {ch}, y.L, y.F, cfg->num_queries, y.query_words, {proof + y.o_caps}, {proof + y.o_final},
                   {proof + y.o_pow}, {proof + y.o_queries});
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* q : scratch) c->release(q);
        return fail(err, e.what());
    } catch (...) {
        (void)hipStreamSynchronize(c->stream);
        for (void* q : scratch) c->release(q);
        return fail(err, "zkm_fri_prove: unknown error");
    }
    for (void* q : scratch) c->release(q);
    *ch_io = local;
    return 0;
}

}  // extern "C"

extern "C" {

size_t zkm_proof_words(const zkm_stark_config* cfg, unsigned log_n, size_t ncols, size_t naux, size_t nctl_zs) {
    try {  // 0 = unsupported configuration (the prove entry points report which field)
        proof_layout y;
        make_layout(y, cfg, log_n, ncols, naux, nctl_zs);
        return y.total;
    } catch (...) {
        return 0;
    }
}

// CtlZData of the benchmark's fake CTL shape: helper columns, no column sets (poseidon_stark.rs:786-799)
static std::vector<zkm_ctl_z> fake_zs(const uint32_t* num_helpers, size_t n) {
    std::vector<zkm_ctl_z> zs(n);
    for (size_t i = 0; i < n; i++) {
        if (num_helpers[i] == 0) throw std::runtime_error("CTLs without helper columns need column sets: use zkm_prove_single_table_ctl");
        zs[i] = zkm_ctl_z{0, 0, num_helpers[i], 0, 0, 0};
    }
    return zs;
}

int zkm_prove_single_table_ctl(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                               const zkm_batch* trace_batch, const uint64_t* aux, size_t naux, const zkm_ctl_table* table,
                               const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs, const uint64_t* lookup_challenges,
                               zkm_challenger* challenger, uint64_t* proof_out, char** err) {
    if (!c || !cfg || !challenger || !proof_out) return fail(err, "zkm_prove_single_table: null argument");
    zkm_challenger local = *challenger;   // (handed back on success only: a failed call leaves the caller's transcript untouched)
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!trace && !trace_batch) throw std::runtime_error("zkm_prove_single_table: need trace values or a trace commitment");
        prove_single_table(c, table_id, cfg, trace, ncols, log_n, trace_batch, aux, naux, table, zs, colset_ids, nzs, lookup_challenges,
                           {&local}, {proof_out});
    } catch (const std::exception& e) {
        return fail(err, e.what());
    } catch (...) {
        return fail(err, "zkm_prove_single_table: unknown error");
    }
    *challenger = local;
    return 0;
}

int zkm_prove_openings(zkm_ctx* c, const zkm_stark_config* cfg, const zkm_batch* trace_batch, const zkm_batch* aux_batch,
                       const zkm_batch* quot_batch, size_t nctl_zs, zkm_challenger* challenger, uint64_t* proof_out, char** err) {
    if (!c || !cfg || !challenger || !proof_out) return fail(err, "zkm_prove_openings: null argument");
    zkm_challenger local = *challenger;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!trace_batch || !aux_batch || !quot_batch) throw std::runtime_error("zkm_prove_openings: three commitments are required");
        prove_single_table(c, -1, cfg, nullptr, trace_batch->ncols, trace_batch->log_n, trace_batch, nullptr, aux_batch->ncols, nullptr,
                           nullptr, nullptr, nctl_zs, nullptr, {&local}, {proof_out}, aux_batch, quot_batch);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    } catch (...) {
        return fail(err, "zkm_prove_openings: unknown error");
    }
    *challenger = local;
    return 0;
}

int zkm_prove_single_table(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                           const zkm_batch* trace_batch, const uint64_t* aux, size_t naux, const uint32_t* num_helpers, size_t nctl_zs,
                           zkm_challenger* challenger, uint64_t* proof_out, char** err) {
    try {
        auto zs = fake_zs(num_helpers, nctl_zs);
        return zkm_prove_single_table_ctl(c, table_id, cfg, trace, ncols, log_n, trace_batch, aux, naux, nullptr, zs.data(), nullptr,
                                          nctl_zs, nullptr, challenger, proof_out, err);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    } catch (...) {
        return fail(err, "zkm_prove_single_table: unknown error");
    }
}

// K proofs of the same table at the same height in LOCK-STEP (the benchmark shape: CtlData given as auxiliary columns, no column sets):
// the K trace and auxiliary commitments are stacked batches, every stage is one launch for all K, each transcript round trip serves
// all K.  Tables with lookups of their own need zkm_prove_segments.
int zkm_prove_single_tables(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, size_t nproofs, const uint64_t* const* traces, size_t ncols,
                            unsigned log_n, const uint64_t* const* aux, size_t naux, const uint32_t* num_helpers, size_t nctl_zs,
                            zkm_challenger* const* challengers, uint64_t* const* proofs_out, char** err) {
    if (!c || !cfg || !traces || !aux || !challengers || !proofs_out) return fail(err, "zkm_prove_single_tables: null argument");
    if (nproofs == 0) return 0;
    zkm_batch *tb = nullptr, *ab = nullptr;
    std::vector<zkm_challenger> local(nproofs);
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (nproofs > ZKM_MAX_SEG) throw std::runtime_error("zkm_prove_single_tables: at most 32 proofs per call");
        validate_config(cfg, log_n);
        if (zkm_num_lookup_columns(table_id, cfg)) throw std::runtime_error("zkm_prove_single_tables: a table with lookups of its own needs zkm_prove_segments");
        auto zs1 = fake_zs(num_helpers, nctl_zs);
        std::vector<zkm_ctl_z> zs;
        for (size_t k = 0; k < nproofs; k++) zs.insert(zs.end(), zs1.begin(), zs1.end());
        std::vector<zkm_challenger*> chs(nproofs);
        std::vector<uint64_t*> proofs(nproofs);
        for (size_t k = 0; k < nproofs; k++) {
            if (!traces[k] || !aux[k] || !challengers[k] || !proofs_out[k]) throw std::runtime_error("zkm_prove_single_tables: null proof argument");
            local[k] = *challengers[k];
            chs[k] = &local[k];
            proofs[k] = proofs_out[k];
        }
        auto stacked = [&](size_t cols, const uint64_t* const* srcs) {
            zkm_batch* b = new zkm_batch();
            b->ctx = c; b->ncols = cols; b->nseg = nproofs; b->log_n = log_n; b->rate_bits = cfg->rate_bits; b->cap_height = cfg->cap_height;
            try {
                if (nproofs == 1) zkm_batch_build(b, srcs[0], true);
                else zkm_batch_build(b, nullptr, true, nullptr, nullptr, srcs);   // (device matrices are transformed where they lie)
            } catch (...) {
                zkm_batch_free(b);
                throw;
            }
            return b;
        };
        {
            zkm_prof_scope st(c, "stage/compute trace commitment");
            tb = stacked(ncols, traces);
        }
        {
            zkm_prof_scope st(c, "stage/compute auxiliary polynomials commitment");
            ab = stacked(naux, aux);
        }
        prove_single_table(c, table_id, cfg, nullptr, ncols, log_n, tb, nullptr, naux, nullptr, zs.data(), nullptr, nctl_zs, nullptr, chs, proofs, ab,
                           nullptr);
    } catch (const std::exception& e) {
        zkm_batch_free(tb);
        zkm_batch_free(ab);
        return fail(err, e.what());
    } catch (...) {
        zkm_batch_free(tb);
        zkm_batch_free(ab);
        return fail(err, "zkm_prove_single_tables: unknown error");
    }
    zkm_batch_free(tb);
    zkm_batch_free(ab);
    for (size_t k = 0; k < nproofs; k++) *challengers[k] = local[k];
    return 0;
}

int zkm_quotient(zkm_ctx* c, int table_id, const zkm_batch* trace, const zkm_batch* aux, const uint32_t* num_helpers, size_t nctl_zs,
                 const uint64_t* alphas, size_t nalphas, uint64_t* out_coeffs, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        auto zs = fake_zs(num_helpers, nctl_zs);
        ctl_dev_owner own;
        own.upload(c, nullptr, zs.data(), nullptr, nctl_zs);
        size_t words = nalphas * 2 * trace->n();
        bool dev = zkm_is_device_ptr(out_coeffs);
        zkm_scratch tmp(c, dev ? 8 : words * 8);   // released on every exit path
        gl_t* d = dev ? out_coeffs : tmp.as<gl_t>();
        quotient_device(c, table_id, trace, aux, own, nullptr, alphas, nalphas, d);
        if (!dev) {
            c->download(out_coeffs, d, words * 8);
        }
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
    return 0;
}

// first row (of n) at which any of the nalphas accumulators is nonzero, into *first (left untouched if none): one pass
__global__ __launch_bounds__(256) void k_first_nonzero_row(const gl_t* __restrict__ acc, size_t n, unsigned nalphas, unsigned long long* first) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool bad = false;
    for (unsigned a = 0; a < nalphas; a++) bad = bad || gl_canon(acc[(size_t)a * n + i]) != 0;
    if (bad) atomicMin(first, (unsigned long long)i);
}

int zkm_check_constraints(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                          const uint64_t* aux, size_t naux, const zkm_ctl_table* table, const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs,
                          const uint64_t* lookup_challenges, const uint64_t* alphas, size_t nalphas, uint64_t* first_failing_row, char** err) {
    std::vector<void*> tmp;
    if (!c || !cfg || !trace || !aux || !alphas || !first_failing_row) return fail(err, "zkm_check_constraints: null argument");
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        validate_config(cfg, log_n);
        if (table_id < 0 || table_id >= 12) throw std::runtime_error("zkm_check_constraints: unknown table id");
        // a table with lookups of its own (Memory, Arithmetic) takes its challenges from the caller: none given is an error here, not a
        // null dereference on the host
        if (!lookup_challenges && zkm_num_lookup_columns(table_id, cfg)) throw std::runtime_error("zkm_check_constraints: this table has lookups of its own: lookup challenges are required");
        if (nalphas < 1 || nalphas > 2) throw std::runtime_error("zkm_check_constraints: 1 or 2 challenges supported");
        for (size_t a = 0; a < nalphas; a++)
            if (alphas[a] >= GL_P) throw std::runtime_error("zkm_check_constraints: non-canonical challenge");
        const size_t n = (size_t)1 << log_n;
        // the values stand where the quotient kernels expect an LDE: "batches" of rate 0 over device copies of the caller's columns
        zkm_batch tb, ab;
        tb.ctx = ab.ctx = c; tb.ncols = ncols; ab.ncols = naux; tb.log_n = ab.log_n = log_n; tb.rate_bits = ab.rate_bits = 0;
        auto on_device = [&](const uint64_t* p, size_t words) -> gl_t* {
            gl_t* d = (gl_t*)c->alloc(words * sizeof(gl_t));
            tmp.push_back(d);
            ZKM_HIP_CHECK(hipMemcpyAsync(d, p, words * sizeof(gl_t), hipMemcpyDefault, c->stream));
            zkm_launch_canon(c, d, words);
            return d;
        };
        tb.lde = on_device(trace, ncols * n);
        ab.lde = on_device(aux, naux * n);
        ctl_dev_owner own;
        own.upload(c, table, zs, colset_ids, nzs, false, ncols);
        gl_t* d_acc = (gl_t*)c->alloc(nalphas * n * sizeof(gl_t));
        tmp.push_back(d_acc);
        quotient_device(c, table_id, &tb, &ab, own, lookup_challenges, alphas, nalphas, d_acc, /*check=*/true);
        unsigned long long* d_first = (unsigned long long*)c->alloc(8);
        tmp.push_back(d_first);
        ZKM_HIP_CHECK(hipMemsetAsync(d_first, 0xff, 8, c->stream));
        hipLaunchKernelGGL(k_first_nonzero_row, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_acc, n, (unsigned)nalphas, d_first);
        ZKM_HIP_CHECK(hipGetLastError());
        unsigned long long first = ~0ULL;
        c->download(&first, d_first, 8);
        for (void* q : tmp) c->release(q);
        tmp.clear();
        *first_failing_row = first;
        if (first != ~0ULL) {
            static const char* const names[] = {"PoseidonStark", "LogicStark", "KeccakSpongeStark", "KeccakStark", "MemoryStark", "PoseidonSpongeStark",
                                                "ShaExtendStark", "ShaExtendSpongeStark", "ShaCompressStark", "ShaCompressSpongeStark", "ArithmeticStark",
                                                "CpuStark"};
            return fail(err, std::string("Constraint failed in ") + names[table_id] + " (first failing row " + std::to_string(first) + ")");   // prover.rs:903-908
        }
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* q : tmp) c->release(q);
        return fail(err, e.what());
    } catch (...) {
        (void)hipStreamSynchronize(c->stream);
        for (void* q : tmp) c->release(q);
        return fail(err, "zkm_check_constraints: unknown error");
    }
    return 0;
}

int zkm_eval_openings(zkm_ctx* c, const zkm_batch* b, const uint64_t zeta[2], uint64_t* out, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        gl2_t z{zeta[0], zeta[1]};
        if (b->nseg != 1) throw std::runtime_error("zkm_eval_openings: stacked batches are internal");
        auto v = eval_batches(c, {b}, &z, &z)[0][0];
        for (size_t i = 0; i < b->ncols; i++) { out[2 * i] = v[i].at_z0.c0; out[2 * i + 1] = v[i].at_z0.c1; }
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
    return 0;
}

}  // extern "C"
