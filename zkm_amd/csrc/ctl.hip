// ctl.hip -- cross-table lookup data on the GPU (K6 / a3) and the multi-table driver prove_with_traces (a1).
//
// Reference: cross_table_lookup_data cross_table_lookup.rs:634-703, get_helper_cols :709-797, partial_sums
// :841-872 (serial over CTLs and rows in the reference); prove_with_traces prover.rs:130-232.
// One lane per trace row: evaluate the filter and the challenge-combined column set, invert (Fermat; the
// reference batch-inverts, the inverse is unique so the bytes agree), accumulate helper columns, then an
// additive suffix scan produces the upside-down running sum Z.
#define GL_REDUCE_BRANCHFREE 1   // (gl_dev.h: these kernels interleave independent products at low occupancy)
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

#include "ctl_dev.h"
#include "all_stark_ctl.inc"

// ------------------------------------------------------------------ descriptor upload + validation
void ctl_dev_owner::upload(zkm_ctx* ctx, const zkm_ctl_table* t, const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs, bool lookup_mode,
                           size_t trace_ncols, size_t nseg) {
    c = ctx;
    if (nseg == 0 || nseg > ZKM_MAX_SEG) throw std::runtime_error("CTL description: bad segment count");
    // (a stacked description: zs holds nseg lists of nzs entries that differ in their challenges only -- segment s of a launch reads list s)
    for (size_t sg = 1; sg < nseg; sg++)
        for (size_t i = 0; i < nzs; i++) {
            const zkm_ctl_z &a = zs[i], &b = zs[sg * nzs + i];
            if (a.ncolsets != b.ncolsets || a.colset_off != b.colset_off || a.num_helpers != b.num_helpers)
                throw std::runtime_error("CTL description: the segments of a stack must share the lookup structure");
        }
    static const zkm_ctl_table empty{};
    if (!t) t = &empty;
    size_t nids = 0, th = 0;
    for (size_t i = 0; i < nzs; i++) {
        if ((size_t)zs[i].colset_off + zs[i].ncolsets > nids) nids = (size_t)zs[i].colset_off + zs[i].ncolsets;
        uint32_t want = (zs[i].ncolsets > 1 || lookup_mode) ? (zs[i].ncolsets + 1) / 2 : 0;
        if (zs[i].ncolsets == 0) {
            if (zs[i].num_helpers == 0) throw std::runtime_error("CTL description: a Z without column sets needs helper columns");
        } else if (zs[i].num_helpers != want) {
            throw std::runtime_error("CTL description: num_helpers must be ceil(ncolsets/2) (0 for a single column set)");
        }
        th += zs[i].num_helpers;
    }
    for (size_t i = 0; i < nids; i++)
        if (colset_ids[i] >= t->ncolsets) throw std::runtime_error("CTL description: column-set index out of range");
    for (size_t i = 0; i < t->ncolsets; i++) {
        const zkm_colset& s = t->colsets[i];
        if ((size_t)s.col_off + s.ncols > t->ncolumns) throw std::runtime_error("CTL description: column range out of bounds");
        if (s.has_filter && ((size_t)s.prod_off + 2 * s.nprod > t->nfilter_idx || (size_t)s.const_off + s.nconst > t->nfilter_idx))
            throw std::runtime_error("CTL description: filter index range out of bounds");
    }
    for (size_t i = 0; i < t->nfilter_idx; i++)
        if (t->filter_idx[i] >= t->ncolumns) throw std::runtime_error("CTL description: filter column index out of range");
    for (size_t i = 0; i < t->ncolumns; i++)
        if ((size_t)t->columns[i].term_off + t->columns[i].n_local + t->columns[i].n_next > t->nterms)
            throw std::runtime_error("CTL description: term range out of bounds");
    for (size_t i = 0; i < t->ncolumns; i++)
        if (t->columns[i].constant >= GL_P) throw std::runtime_error("CTL description: non-canonical constant");
    for (size_t i = 0; i < t->nterms; i++) {
        if (t->term_coeff[i] >= GL_P) throw std::runtime_error("CTL description: non-canonical coefficient");
        // every kernel that evaluates the description indexes trace columns with term_col: checked before anything is launched
        if (trace_ncols && t->term_col[i] >= trace_ncols) throw std::runtime_error("CTL description: trace column index out of range");
    }
    // one blob: [columns | term_coeff | colsets | zs | term_col | filter_idx | colset_ids]
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t o_cols = 0, o_coeff = al(o_cols + t->ncolumns * sizeof(zkm_column)), o_sets = al(o_coeff + t->nterms * 8);
    size_t o_zs = al(o_sets + t->ncolsets * sizeof(zkm_colset)), o_tc = al(o_zs + nseg * nzs * sizeof(zkm_ctl_z));
    size_t o_fi = al(o_tc + t->nterms * 4), o_ids = al(o_fi + t->nfilter_idx * 4), total = al(o_ids + nids * 4) + 16;
    std::vector<char> h(total, 0);
    if (t->ncolumns) memcpy(&h[o_cols], t->columns, t->ncolumns * sizeof(zkm_column));
    if (t->nterms) { memcpy(&h[o_coeff], t->term_coeff, t->nterms * 8); memcpy(&h[o_tc], t->term_col, t->nterms * 4); }
    if (t->ncolsets) memcpy(&h[o_sets], t->colsets, t->ncolsets * sizeof(zkm_colset));
    if (nzs) memcpy(&h[o_zs], zs, nseg * nzs * sizeof(zkm_ctl_z));
    if (t->nfilter_idx) memcpy(&h[o_fi], t->filter_idx, t->nfilter_idx * 4);
    if (nids) memcpy(&h[o_ids], colset_ids, nids * 4);
    blob = c->alloc(total);
    h_zs.assign(zs, zs + nzs);
    c->upload(blob, h.data(), total);   // (staged through the context's pinned ring or by the runtime: `h` may go)
    char* b = (char*)blob;
    d.columns = (const zkm_column*)(b + o_cols);
    d.term_coeff = (const uint64_t*)(b + o_coeff);
    d.colsets = (const zkm_colset*)(b + o_sets);
    d.zs = (const zkm_ctl_z*)(b + o_zs);
    d.term_col = (const uint32_t*)(b + o_tc);
    d.filter_idx = (const uint32_t*)(b + o_fi);
    d.colset_ids = (const uint32_t*)(b + o_ids);
    d.nzs = (uint32_t)nzs;
    d.total_helpers = (uint32_t)th;
    naux = th + nzs;
}
ctl_dev_owner::~ctl_dev_owner() {
    if (c && blob) {
        (void)hipStreamSynchronize(c->stream);
        c->release(blob);
    }
}

// ------------------------------------------------------------------ K6 kernels
// helper columns + per-row sum of all terms of CtlZData zi0 + blockIdx.y (a run of Zs in one launch): helper columns of Z zi start at
// aux + (helper columns of the Zs before it) * n, the row sums go to hsum_all + zi * n
// (blockIdx.z = segment of a stack: its trace / auxiliary columns start trace_seg / aux_seg words after the previous segment's, its row
// sums nzs * n words, and it reads its own list of CtlZData -- its own challenges)
__global__ __launch_bounds__(256) void k_ctl_terms(ctl_dev d, uint32_t zi0, const gl_t* __restrict__ trace, size_t n,
                                                   gl_t* __restrict__ aux, gl_t* __restrict__ hsum_all, int* __restrict__ bad, size_t trace_seg,
                                                   size_t aux_seg) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    trace += (size_t)blockIdx.z * trace_seg;
    aux += (size_t)blockIdx.z * aux_seg;
    hsum_all += (size_t)blockIdx.z * d.nzs * n;
    d.zs += (size_t)blockIdx.z * d.nzs;
    const uint32_t zi = zi0 + blockIdx.y;
    const zkm_ctl_z z = d.zs[zi];
    size_t hstart = 0;
    for (uint32_t k = 0; k < zi; k++) hstart += d.zs[k].num_helpers;     // (wave-uniform scalar loads; a table has at most a few dozen Zs)
    gl_t* const helpers = z.num_helpers ? aux + hstart * n : nullptr;
    gl_t* const hsum = hsum_all + (size_t)zi * n;
    const uint32_t* ids = d.colset_ids + z.colset_off;
    const gl_t* lv = trace + row;
    bool next_ok = row + 1 < n;
    gl_t total = 0;
    // The reference batch-inverts the combined columns (batch_multiplicative_inverse, cross_table_lookup.rs:748, 781); so does a lane
    // here, eight column sets (four helper columns) at a time: prefix products, ONE inversion, back-substitution -- 3 products per
    // set instead of an inversion each (a logUp lookup of the Arithmetic table has 18 looking columns per row).  Inverses are unique:
    // the same words as one inversion per set; a set whose filter is 0 (or whose combination is 0) contributes 0 as before.
    for (uint32_t j0 = 0; 2 * j0 < z.ncolsets; j0 += 4) {
        gl_t v[8], pre[8];
        bool on[8];
        gl_t acc = 1;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t ci = 2 * j0 + e;
            on[e] = false;
            v[e] = 0;
            if (ci < z.ncolsets) {
                const zkm_colset cs = d.colsets[ids[ci]];
                gl_t f = ctl_eval_filter(d, cs, lv, n, 1, next_ok);
                if (f == 1) {
                    v[e] = gl_canon(ctl_combine(d, cs, z.beta, z.gamma, lv, n, 1, next_ok));
                    on[e] = v[e] != 0;
                } else if (f != 0) {
                    bad[blockIdx.z] = 1;  // "Non-binary filter?" (cross_table_lookup.rs:741) -- per segment of a stack
                }
            }
            pre[e] = acc;
            if (on[e]) acc = gl_mul(acc, v[e]);
        }
        gl_t inv = gl_inv(acc);
#pragma unroll
        for (int e = 7; e >= 0; e--) {
            if (on[e]) {
                const gl_t r = gl_mul(inv, pre[e]);
                inv = gl_mul(inv, v[e]);
                v[e] = r;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t j = j0 + q;
            if (2 * j < z.ncolsets) {
                const gl_t h = gl_add(v[2 * q], v[2 * q + 1]);
                if (helpers) helpers[(size_t)j * n + row] = h;
                total = gl_add(total, h);
            }
        }
    }
    hsum[row] = total;
}

// Tables with many looking column sets and few rows (KeccakSponge: 136 + 34 sets, a few thousand rows) are latency-bound in
// k_ctl_terms (one thread walks every set of its row); here blockIdx.y picks the helper column, and k_ctl_rowsum adds them up.
__global__ __launch_bounds__(256) void k_ctl_helper(ctl_dev d, uint32_t zi, const gl_t* __restrict__ trace, size_t n, gl_t* __restrict__ helpers,
                                                    int* __restrict__ bad, size_t trace_seg, size_t aux_seg) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    trace += (size_t)blockIdx.z * trace_seg;
    helpers += (size_t)blockIdx.z * aux_seg;
    const zkm_ctl_z z = d.zs[(size_t)blockIdx.z * d.nzs + zi];
    const uint32_t* ids = d.colset_ids + z.colset_off;
    const uint32_t j = blockIdx.y;
    const gl_t* lv = trace + row;
    bool next_ok = row + 1 < n;
    // (the two column sets of a helper column share one inversion: 1 / a + 1 / b = (a + b) / (a b))
    gl_t v[2] = {0, 0};
    bool on[2] = {false, false};
#pragma unroll
    for (int e = 0; e < 2; e++) {
        if (2 * j + e < z.ncolsets) {
            const zkm_colset cs = d.colsets[ids[2 * j + e]];
            gl_t f = ctl_eval_filter(d, cs, lv, n, 1, next_ok);
            if (f == 1) {
                v[e] = gl_canon(ctl_combine(d, cs, z.beta, z.gamma, lv, n, 1, next_ok));
                on[e] = v[e] != 0;
            } else if (f != 0) {
                bad[blockIdx.z] = 1;  // "Non-binary filter?" (cross_table_lookup.rs:741)
            }
        }
    }
    gl_t h = 0;
    if (on[0] && on[1]) h = gl_mul(gl_add(v[0], v[1]), gl_inv(gl_mul(v[0], v[1])));
    else if (on[0]) h = gl_inv(v[0]);
    else if (on[1]) h = gl_inv(v[1]);
    helpers[(size_t)j * n + row] = gl_canon(h);
}
__global__ __launch_bounds__(256) void k_ctl_rowsum(const gl_t* __restrict__ helpers, uint32_t nh, size_t n, gl_t* __restrict__ hsum, size_t aux_seg,
                                                    size_t hsum_seg) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    helpers += (size_t)blockIdx.z * aux_seg;
    hsum += (size_t)blockIdx.z * hsum_seg;
    gl_t total = 0;
    for (uint32_t j = 0; j < nh; j++) total = gl_add(total, helpers[(size_t)j * n + row]);
    hsum[row] = total;
}

// additive suffix scans of nb arrays of length m at stride `stride` (blockIdx.y = array), segments of 64:  totals[b][s] = sum of segment s
// (blockIdx.z = segment of a stack: a += z * a_seg; the totals of all arrays of a segment are contiguous)
__global__ void k_sum_totals(const gl_t* __restrict__ a, size_t stride, size_t m, gl_t* __restrict__ out, size_t a_seg) {
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nseg = (m + 63) / 64;
    if (s >= nseg) return;
    a += (size_t)blockIdx.y * stride + (size_t)blockIdx.z * a_seg;
    out += (size_t)blockIdx.z * gridDim.y * nseg;
    size_t end = (s + 1) * 64 < m ? (s + 1) * 64 : m;
    gl_t acc = 0;
    for (size_t k = s * 64; k < end; k++) acc = gl_add(acc, a[k]);
    out[(size_t)blockIdx.y * nseg + s] = acc;
}
// S[k] = a[k] + S[k+1] inside each segment, carry-in = upper[s+1]   (upper: [array][nupper])
__global__ void k_sum_scan(const gl_t* __restrict__ a, size_t stride, size_t m, const gl_t* __restrict__ upper, size_t nupper,
                           gl_t* __restrict__ out, size_t out_stride, size_t a_seg, size_t out_seg) {
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nseg = (m + 63) / 64;
    if (s >= nseg) return;
    a += (size_t)blockIdx.y * stride + (size_t)blockIdx.z * a_seg;
    out += (size_t)blockIdx.y * out_stride + (size_t)blockIdx.z * out_seg;
    if (upper) upper += (size_t)blockIdx.z * gridDim.y * nupper;
    gl_t acc = (upper && s + 1 < nupper) ? upper[(size_t)blockIdx.y * nupper + s + 1] : 0;
    size_t end = (s + 1) * 64 < m ? (s + 1) * 64 : m;
    for (size_t k = end; k-- > s * 64;) {
        acc = gl_add(acc, a[k]);
        out[k] = acc;
    }
}

// out[b][k] = sum_{m >= k} a[b][m] for nb arrays (a[b] = a + b * a_stride, out[b] = out + b * out_stride; out may alias a): one launch per
// level for all arrays -- the Z columns of a table are built together (a table of the reference has up to 28 of them)
static void suffix_sum(zkm_ctx* c, const gl_t* a, size_t a_stride, size_t n, size_t nb, gl_t* out, size_t out_stride, size_t nstack = 1,
                       size_t a_seg = 0, size_t out_seg = 0) {
    if (!nb) return;
    const unsigned z = (unsigned)nstack;
    struct level { const gl_t* t; size_t stride, m, seg; };
    std::vector<level> lv{{a, a_stride, n, a_seg}};
    std::vector<void*> tmp;
    while (lv.back().m > 64) {
        size_t nseg = (lv.back().m + 63) / 64;
        gl_t* t = (gl_t*)c->alloc(nstack * nb * nseg * sizeof(gl_t));
        tmp.push_back(t);
        hipLaunchKernelGGL(k_sum_totals, dim3((unsigned)((nseg + 63) / 64), (unsigned)nb, z), dim3(64), 0, c->stream, lv.back().t, lv.back().stride,
                           lv.back().m, t, lv.back().seg);
        lv.push_back({t, nseg, nseg, nb * nseg});
    }
    std::vector<gl_t*> S(lv.size(), nullptr);
    for (size_t l = lv.size(); l-- > 0;) {
        size_t nseg = (lv[l].m + 63) / 64;
        const gl_t* upper = l + 1 < lv.size() ? S[l + 1] : nullptr;
        size_t nupper = l + 1 < lv.size() ? lv[l + 1].m : 0;
        if (l == 0) S[l] = out;
        else { S[l] = (gl_t*)c->alloc(nstack * nb * lv[l].m * sizeof(gl_t)); tmp.push_back(S[l]); }
        hipLaunchKernelGGL(k_sum_scan, dim3((unsigned)((nseg + 63) / 64), (unsigned)nb, z), dim3(64), 0, c->stream, lv[l].t, lv[l].stride, lv[l].m,
                           upper, nupper, S[l], l == 0 ? out_stride : lv[l].m, lv[l].seg, l == 0 ? out_seg : nb * lv[l].m);
    }
    ZKM_HIP_CHECK(hipGetLastError());
    // no host sync: the temporaries go back to the caching allocator, whose blocks are only ever reused by later work on this stream
    for (void* q : tmp) c->release(q);
}

// logUp: x[i] = hsum[i] - frequencies[i] / (challenge + table[i])   (lookup.rs:100-116)
// (segment blockIdx.z of a stack: its challenge is the gamma of its CtlZData list's only entry)
__global__ __launch_bounds__(256) void k_lookup_x(ctl_dev d, uint32_t table_col, uint32_t freq_col,
                                                  const gl_t* __restrict__ trace, size_t n, const gl_t* __restrict__ hsum,
                                                  gl_t* __restrict__ x, size_t trace_seg) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    trace += (size_t)blockIdx.z * trace_seg;
    hsum += (size_t)blockIdx.z * n;
    x += (size_t)blockIdx.z * n;
    const gl_t challenge = d.zs[(size_t)blockIdx.z * d.nzs].gamma;
    bool next_ok = row + 1 < n;
    gl_t tinv = gl_inv(gl_add(challenge, ctl_eval_column(d, table_col, trace + row, n, 1, next_ok)));
    x[row] = gl_sub(hsum[row], gl_mul(ctl_eval_column(d, freq_col, trace + row, n, 1, next_ok), tinv));
}
// exclusive prefix sum from the suffix sums: z[k] = S[0] - S[k]
__global__ __launch_bounds__(256) void k_prefix_from_suffix(const gl_t* __restrict__ S, size_t n, gl_t* __restrict__ z, size_t z_seg) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    S += (size_t)blockIdx.z * n;
    z += (size_t)blockIdx.z * z_seg;
    if (k < n) z[k] = gl_sub(S[0], S[k]);
}

// aux (device, naux x n) = helper columns (zs order) ++ Z columns.  A stacked description (own: nseg lists of CtlZData): segment s reads
// the trace at d_trace + s * trace_seg and writes its columns at d_aux + s * aux_seg.
void zkm_ctl_data_device(zkm_ctx* c, const ctl_dev_owner& own, const gl_t* d_trace, unsigned log_n, gl_t* d_aux, size_t nseg, size_t trace_seg,
                         size_t aux_seg) {
    size_t n = (size_t)1 << log_n;
    const std::vector<zkm_ctl_z>& zs = own.h_zs;             // (host copy kept by upload(): no round trip for the launch planning)
    const size_t nzs = zs.size();
    const unsigned z = (unsigned)nseg;
    gl_t* const d_hsum_all = (gl_t*)c->alloc(nseg * (nzs ? nzs : 1) * n * sizeof(gl_t));   // the per-row sums of every Z, scanned together below
    int* d_bad = (int*)c->alloc(ZKM_MAX_SEG * sizeof(int));     // one flag per segment of the stack
    ZKM_HIP_CHECK(hipMemsetAsync(d_bad, 0, ZKM_MAX_SEG * sizeof(int), c->stream));
    // Zs with many helper columns on a short table spread their column sets over blockIdx.y (few workgroups per launch otherwise);
    // every other run of consecutive Zs is ONE launch (blockIdx.y = Z)
    auto wide = [&](uint32_t i) { return zs[i].num_helpers >= 4 && ((n * nseg) >> 8) < 4096; };
    size_t hstart = 0;
    for (uint32_t i = 0; i < nzs;) {
        zkm_prof_scope ps(c, "ctl_terms");
        if (wide(i)) {
            gl_t* helpers = d_aux + hstart * n;
            const uint32_t nh = zs[i].num_helpers;
            hipLaunchKernelGGL(k_ctl_helper, dim3((n + 255) / 256, nh, z), dim3(256), 0, c->stream, own.d, i, d_trace, n, helpers, d_bad, trace_seg, aux_seg);
            hipLaunchKernelGGL(k_ctl_rowsum, dim3((n + 255) / 256, 1, z), dim3(256), 0, c->stream, helpers, nh, n, d_hsum_all + (size_t)i * n, aux_seg,
                               nzs * n);
            hstart += nh;
            i++;
        } else {
            uint32_t run = 0;
            while (i + run < nzs && !wide(i + run)) { hstart += zs[i + run].num_helpers; run++; }
            hipLaunchKernelGGL(k_ctl_terms, dim3((n + 255) / 256, run, z), dim3(256), 0, c->stream, own.d, i, d_trace, n, d_aux, d_hsum_all, d_bad,
                               trace_seg, aux_seg);
            i += run;
        }
        ZKM_HIP_CHECK(hipGetLastError());
    }
    {
        zkm_prof_scope ps(c, "ctl_suffix_sum");
        suffix_sum(c, d_hsum_all, n, n, nzs, d_aux + (size_t)own.d.total_helpers * n, n, nseg, nzs * n, aux_seg);
    }
    int bad[ZKM_MAX_SEG];
    c->download(bad, d_bad, ZKM_MAX_SEG * sizeof(int));
    c->release(d_hsum_all);
    c->release(d_bad);
    for (size_t sg = 0; sg < nseg; sg++)
        if (bad[sg]) throw zkm_segment_error(sg, "Non-binary filter?");
}

// ------------------------------------------------------------------ per-table CtlZData lists (cross_table_lookup_data order)
struct table_zs {
    std::vector<zkm_ctl_z> zs;
    std::vector<uint32_t> ids;
    size_t naux = 0;
};
static std::vector<table_zs> derive_zs(size_t ntables, const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides, size_t nctls,
                                       size_t nch, const uint64_t* challenges) {
    std::vector<table_zs> out(ntables);
    for (size_t c = 0; c < nctls; c++) {
        const zkm_ctl_side* lk = sides + ctls[c].looking_off;
        if (ctls[c].looked.table >= ntables) throw std::runtime_error("CTL: looked table index out of range");
        // a table's looking entries must be consecutive (the reference groups consecutive runs, :807, but collects
        // columns over all entries of the table, :656-671 -- the two only agree for consecutive entries)
        for (uint32_t i = 0; i < ctls[c].nlooking; i++) {
            if (lk[i].table >= ntables) throw std::runtime_error("CTL: looking table index out of range");
            for (uint32_t j = i + 1; j < ctls[c].nlooking; j++)
                if (lk[j].table == lk[i].table && lk[j - 1].table != lk[i].table)
                    throw std::runtime_error("CTL: looking entries of one table must be consecutive");
        }
        for (size_t ch = 0; ch < nch; ch++) {
            uint64_t beta = challenges ? challenges[2 * ch] : 0, gamma = challenges ? challenges[2 * ch + 1] : 0;
            for (uint32_t i = 0; i < ctls[c].nlooking;) {
                uint32_t j = i;
                while (j < ctls[c].nlooking && lk[j].table == lk[i].table) j++;
                table_zs& o = out[lk[i].table];
                zkm_ctl_z z{};
                z.ncolsets = j - i; z.colset_off = (uint32_t)o.ids.size(); z.beta = beta; z.gamma = gamma;
                z.num_helpers = (j - i) > 1 ? (j - i + 1) / 2 : 0;
                for (uint32_t k = i; k < j; k++) o.ids.push_back(lk[k].colset);
                o.zs.push_back(z);
                o.naux += z.num_helpers + 1;
                i = j;
            }
            table_zs& o = out[ctls[c].looked.table];
            zkm_ctl_z z{};
            z.ncolsets = 1; z.colset_off = (uint32_t)o.ids.size(); z.beta = beta; z.gamma = gamma;
            o.ids.push_back(ctls[c].looked.colset);
            o.zs.push_back(z);
            o.naux += 1;
        }
    }
    return out;
}

static int fail(char** err, const std::string& msg) {
    if (err) {
        *err = (char*)malloc(msg.size() + 1);
        if (*err) memcpy(*err, msg.c_str(), msg.size() + 1);
    }
    return 1;
}

extern "C" {

int zkm_ctl_data(zkm_ctx* c, const zkm_ctl_table* table, const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs,
                 const uint64_t* trace, size_t ncols, unsigned log_n, uint64_t* aux_out, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        size_t n = (size_t)1 << log_n;
        ctl_dev_owner own;
        own.upload(c, table, zs, colset_ids, nzs, false, ncols);
        for (size_t i = 0; i < (table ? table->nterms : 0); i++)
            if (table->term_col[i] >= ncols) throw std::runtime_error("CTL description: trace column index out of range");
        bool tdev = zkm_is_device_ptr(trace), adev = zkm_is_device_ptr(aux_out);
        gl_t* d_trace = tdev ? const_cast<gl_t*>(trace) : (gl_t*)c->alloc(ncols * n * 8);
        if (!tdev) ZKM_HIP_CHECK(hipMemcpyAsync(d_trace, trace, ncols * n * 8, hipMemcpyHostToDevice, c->stream));
        gl_t* d_aux = adev ? aux_out : (gl_t*)c->alloc(own.naux * n * 8);
        try {
            zkm_ctl_data_device(c, own, d_trace, log_n, d_aux, 1, 0, 0);
            if (!adev) ZKM_HIP_CHECK(hipMemcpyAsync(aux_out, d_aux, own.naux * n * 8, hipMemcpyDeviceToHost, c->stream));
            c->sync();
        } catch (...) {
            (void)hipStreamSynchronize(c->stream);
            if (!tdev) c->release(d_trace);
            if (!adev) c->release(d_aux);
            throw;
        }
        if (!tdev) c->release(d_trace);
        if (!adev) c->release(d_aux);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
    return 0;
}

}  // extern "C"

// (ceil(nlookup / 2) + 1) columns of n into d_out (device): helper columns then Z (lookup.rs:46-124).  A stack of nseg segments:
// challenges[s] is segment s's, its trace starts trace_seg words and its output out_seg words after the previous segment's.
static void lookup_helper_columns_device(zkm_ctx* c, const zkm_ctl_table* table, const uint32_t* colset_ids, size_t nlookup, uint32_t table_col,
                                         uint32_t freq_col, const uint64_t* challenges, const gl_t* d_trace, size_t n, gl_t* d_out,
                                         size_t trace_ncols = 0, size_t nseg = 1, size_t trace_seg = 0, size_t out_seg = 0) {
    std::vector<void*> tmp;
    size_t nh = (nlookup + 1) / 2;
    std::vector<zkm_ctl_z> zl(nseg);
    for (size_t sg = 0; sg < nseg; sg++) zl[sg] = zkm_ctl_z{(uint32_t)nlookup, 0, (uint32_t)nh, 0, 1, challenges[sg]};  // GrandProductChallenge{beta: 1, gamma: challenge}
    ctl_dev_owner own;
    own.upload(c, table, zl.data(), colset_ids, 1, /*lookup_mode=*/true, trace_ncols, nseg);
    const unsigned z = (unsigned)nseg;
    try {
        gl_t* d_hsum = (gl_t*)c->alloc(nseg * n * 8);
        tmp.push_back(d_hsum);
        gl_t* d_x = (gl_t*)c->alloc(nseg * n * 8);
        tmp.push_back(d_x);
        int* d_bad = (int*)c->alloc(ZKM_MAX_SEG * sizeof(int));
        tmp.push_back(d_bad);
        ZKM_HIP_CHECK(hipMemsetAsync(d_bad, 0, ZKM_MAX_SEG * sizeof(int), c->stream));
        {
            zkm_prof_scope ps(c, "lookup_terms");
            hipLaunchKernelGGL(k_ctl_terms, dim3((n + 255) / 256, 1, z), dim3(256), 0, c->stream, own.d, 0u, d_trace, n, d_out, d_hsum, d_bad, trace_seg,
                               out_seg);
            hipLaunchKernelGGL(k_lookup_x, dim3((n + 255) / 256, 1, z), dim3(256), 0, c->stream, own.d, table_col, freq_col, d_trace, n, d_hsum, d_x,
                               trace_seg);
            ZKM_HIP_CHECK(hipGetLastError());
        }
        suffix_sum(c, d_x, n, n, 1, d_hsum, n, nseg, n, n);
        hipLaunchKernelGGL(k_prefix_from_suffix, dim3((n + 255) / 256, 1, z), dim3(256), 0, c->stream, d_hsum, n, d_out + nh * n, out_seg);
        ZKM_HIP_CHECK(hipGetLastError());
        int bad[ZKM_MAX_SEG];
        c->download(bad, d_bad, ZKM_MAX_SEG * sizeof(int));
        for (void* p : tmp) c->release(p);
        tmp.clear();
        for (size_t sg = 0; sg < nseg; sg++)
            if (bad[sg]) throw zkm_segment_error(sg, "Non-binary filter?");
    } catch (...) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        throw;
    }
}

// A table's own logUp lookups (Stark::lookups()): every use in the reference is Column::single columns without filters
// (memory_stark.rs:476-483, arithmetic_stark.rs:269-276).  Writes zkm_num_lookup_columns() columns of n into d_out:
// per lookup, per challenge: helper columns then Z (prover.rs:475-493).
// Stacked: challenges = nseg x nch (segment-major); segment s reads d_trace + s * trace_seg and writes d_out + s * out_seg.
void zkm_table_lookup_columns_device(zkm_ctx* c, int table_id, const uint64_t* challenges, size_t nch, const gl_t* d_trace, size_t n,
                                     gl_t* d_out, size_t nseg, size_t trace_seg, size_t out_seg) {
    size_t nl = 0, off = 0;
    const zkm_table_lookup* defs = zkm_table_lookups(table_id, &nl);
    for (size_t l = 0; l < nl; l++) {
        const zkm_table_lookup& d = defs[l];
        // describe the lookup with the CTL column machinery: ncols single-column sets, then the table / frequencies columns
        std::vector<zkm_column> cols(d.ncols + 2);
        std::vector<uint32_t> tc(d.ncols + 2);
        std::vector<uint64_t> tf(d.ncols + 2, 1);
        std::vector<zkm_colset> sets(d.ncols);
        std::vector<uint32_t> ids(d.ncols);
        for (uint32_t i = 0; i < d.ncols + 2; i++) {
            tc[i] = i < d.ncols ? d.cols[i] : i == d.ncols ? d.table_col : d.freq_col;
            cols[i] = zkm_column{1, 0, i, 0, 0};
            if (i < d.ncols) { sets[i] = zkm_colset{1, i, 0, 0, 0, 0, 0, 0}; ids[i] = i; }
        }
        zkm_ctl_table t{cols.data(), cols.size(), tc.data(), tf.data(), tc.size(), sets.data(), sets.size(), nullptr, 0};
        for (size_t k = 0; k < nch; k++) {
            std::vector<uint64_t> chk(nseg);
            for (size_t sg = 0; sg < nseg; sg++) chk[sg] = challenges[sg * nch + k];
            lookup_helper_columns_device(c, &t, ids.data(), d.ncols, d.ncols, d.ncols + 1, chk.data(), d_trace, n, d_out + off * n, 0, nseg, trace_seg,
                                         out_seg);
            off += (d.ncols + 1) / 2 + 1;
        }
    }
}

extern "C" {

int zkm_lookup_helper_columns(zkm_ctx* c, const zkm_ctl_table* table, const uint32_t* colset_ids, size_t nlookup, uint32_t table_col,
                              uint32_t freq_col, uint64_t challenge, const uint64_t* trace, size_t ncols, unsigned log_n,
                              uint64_t* out, char** err) {
    std::vector<void*> tmp;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!table || nlookup == 0) throw std::runtime_error("zkm_lookup_helper_columns: empty lookup");
        if (table_col >= table->ncolumns || freq_col >= table->ncolumns) throw std::runtime_error("lookup: column index out of range");
        if (challenge >= GL_P) throw std::runtime_error("lookup: non-canonical challenge");
        for (size_t i = 0; i < table->nterms; i++)
            if (table->term_col[i] >= ncols) throw std::runtime_error("CTL description: trace column index out of range");
        size_t n = (size_t)1 << log_n, nh = (nlookup + 1) / 2;
        for (size_t i = 0; i < nlookup; i++)
            if (colset_ids[i] < table->ncolsets && table->colsets[colset_ids[i]].ncols != 1)
                throw std::runtime_error("lookup: every looking entry must be a single-column set");
        bool tdev = zkm_is_device_ptr(trace), odev = zkm_is_device_ptr(out);
        gl_t* d_trace = tdev ? const_cast<gl_t*>(trace) : (gl_t*)c->alloc(ncols * n * 8);
        if (!tdev) { tmp.push_back(d_trace); ZKM_HIP_CHECK(hipMemcpyAsync(d_trace, trace, ncols * n * 8, hipMemcpyHostToDevice, c->stream)); }
        gl_t* d_out = odev ? out : (gl_t*)c->alloc((nh + 1) * n * 8);
        if (!odev) tmp.push_back(d_out);
        lookup_helper_columns_device(c, table, colset_ids, nlookup, table_col, freq_col, &challenge, d_trace, n, d_out, ncols);
        if (!odev) ZKM_HIP_CHECK(hipMemcpyAsync(out, d_out, (nh + 1) * n * 8, hipMemcpyDeviceToHost, c->stream));
        c->sync();
        for (void* p : tmp) c->release(p);
        tmp.clear();
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}

// ---- the AllStark description shipped with the library (all_stark_ctl.inc, generated from zkm_amd/tables.py)
int zkm_all_stark_ctls(const zkm_cross_table_lookup** ctls_out, size_t* nctls_out, const zkm_ctl_side** sides_out, size_t* nsides_out) {
    if (ctls_out) *ctls_out = AS_CTLS;
    if (nctls_out) *nctls_out = AS_NCTLS;
    if (sides_out) *sides_out = AS_SIDES;
    if (nsides_out) *nsides_out = AS_NSIDES;
    return 0;
}
const zkm_ctl_table* zkm_all_stark_ctl_table(int table_id) {
    int e = zkm_table_enum_index(table_id);
    return e < 0 ? nullptr : &AS_CTL_TABLES[e];
}
int zkm_prove_segment(zkm_ctx* c, const zkm_stark_config* cfg, const uint64_t* const* traces, const unsigned* log_n,
                      const uint64_t* pub, size_t npub, uint64_t* proofs, size_t* offsets_out, uint64_t* challenges, char** err) {
    if (!traces || !log_n) return fail(err, "zkm_prove_segment: null argument");
    zkm_table_input tables[12];
    for (int t = 0; t < 12; t++) tables[t] = zkm_table_input{AS_TABLE_IDS[t], traces[t], AS_TABLE_WIDTH[t], log_n[t], &AS_CTL_TABLES[t], nullptr};
    size_t offs[13];
    size_t total = zkm_all_proof_words(cfg, tables, 12, AS_CTLS, AS_SIDES, AS_NCTLS, offs);
    if (!total) return fail(err, "zkm_prove_segment: unsupported configuration or table size");
    if (offsets_out) memcpy(offsets_out, offs, sizeof offs);
    if (!proofs) return 0;  // sizing pass
    if (!c || !challenges) return fail(err, "zkm_prove_segment: null argument");
    return zkm_prove_with_traces(c, cfg, tables, 12, AS_CTLS, AS_SIDES, AS_NCTLS, pub, npub, proofs, challenges, err);
}

int zkm_prove_segment_columns(zkm_ctx* c, const zkm_stark_config* cfg, const uint64_t* const* const* columns, const unsigned* log_n,
                              const uint64_t* pub, size_t npub, uint64_t* proofs, size_t* offsets_out, uint64_t* challenges, char** err) {
    if (!columns || !log_n) return fail(err, "zkm_prove_segment_columns: null argument");
    zkm_table_input tables[12];
    for (int t = 0; t < 12; t++) {
        if (!columns[t]) return fail(err, "zkm_prove_segment_columns: null table");
        tables[t] = zkm_table_input{AS_TABLE_IDS[t], nullptr, AS_TABLE_WIDTH[t], log_n[t], &AS_CTL_TABLES[t], columns[t]};
    }
    size_t offs[13];
    size_t total = zkm_all_proof_words(cfg, tables, 12, AS_CTLS, AS_SIDES, AS_NCTLS, offs);
    if (!total) return fail(err, "zkm_prove_segment_columns: unsupported configuration or table size");
    if (offsets_out) memcpy(offsets_out, offs, sizeof offs);
    if (!proofs) return 0;  // sizing pass
    if (!c || !challenges) return fail(err, "zkm_prove_segment_columns: null argument");
    return zkm_prove_with_traces(c, cfg, tables, 12, AS_CTLS, AS_SIDES, AS_NCTLS, pub, npub, proofs, challenges, err);
}

int zkm_table_enum_index(int table_id) {
    static const int order[12] = {ZKM_TABLE_ARITHMETIC, ZKM_TABLE_CPU, ZKM_TABLE_POSEIDON, ZKM_TABLE_POSEIDON_SPONGE, ZKM_TABLE_KECCAK,
                                  ZKM_TABLE_KECCAK_SPONGE, ZKM_TABLE_SHA_EXTEND, ZKM_TABLE_SHA_EXTEND_SPONGE, ZKM_TABLE_SHA_COMPRESS,
                                  ZKM_TABLE_SHA_COMPRESS_SPONGE, ZKM_TABLE_LOGIC, ZKM_TABLE_MEMORY};
    for (int i = 0; i < 12; i++)
        if (order[i] == table_id) return i;
    return -1;
}

size_t zkm_all_proof_words(const zkm_stark_config* cfg, const zkm_table_input* tables, size_t ntables,
                           const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides, size_t nctls, size_t* offs) {
    try {
        auto tz = derive_zs(ntables, ctls, sides, nctls, cfg->num_challenges, nullptr);
        size_t total = 0;
        for (size_t t = 0; t < ntables; t++) {
            if (offs) offs[t] = total;
            size_t w = zkm_proof_words(cfg, tables[t].log_n, tables[t].ncols, zkm_num_lookup_columns(tables[t].table_id, cfg) + tz[t].naux,
                                       tz[t].zs.size());
            if (!w) return 0;  // unsupported configuration / table height
            total += w;
        }
        if (offs) offs[ntables] = total;
        return total;
    } catch (...) {
        return 0;
    }
}

}  // extern "C"

// Work that is independent per table -- the trace commitments, and after the CTL challenges the CTL data + auxiliary commitments --
// runs side by side on the context's COMMIT LANES (sub-contexts with their own stream, allocator and tables, one host thread
// each; the calling thread works on the context itself).  A segment of the reference's default size has nine tables of 2^6 .. 2^13
// rows whose launches cannot fill the GPU and are chains of latency-bound steps (small Merkle levels, cap downloads): `small`
// tables are pulled from one queue, largest first; `big` ones (LDE over 1 GiB: they fill the machine on their own) stay on the
// context.  fn(worker context, table index); the first exception stops the queue and is rethrown on the caller.
// Estimated time of committing `ncols` columns of 2^log_n rows (seconds; rate 4): every 8 columns are one absorb step of the leaf
// sponge, and a step costs the larger of the permutation's latency in the form the matrix gets (hash.hip: 16 lanes per leaf up to
// wide_max_hashes rows, four lanes up to quad_max_hashes, one lane beyond) and the time the whole machine needs for that many hashes.
static double commit_cost_estimate(const zkm_ctx* c, size_t ncols, unsigned log_n, unsigned rate_bits, size_t nstack = 1) {
    const double rows = (double)((size_t)1 << (log_n + rate_bits)) * (double)nstack;   // (the rows of a stack are hashed in one launch)
    const double full_rate = 3.3e9;                        // one-lane permutations per second of the whole GPU
    double lat, rate;
    if (rows <= (double)c->wide_max_hashes) { lat = 13e-6; rate = full_rate / 4.2; }
    else if (rows <= (double)c->quad_max_hashes) { lat = 16e-6; rate = full_rate / 1.45; }
    else { lat = 40e-6; rate = full_rate; }
    const double step = rows / rate > lat ? rows / rate : lat;
    return (double)((ncols + 7) / 8) * step + rows * (double)ncols * 2.5e-11 + 2e-4;    // + transforms (~25 ps per LDE word) + launches
}

// Independent per-table jobs on the context and its commit lanes (one host thread each).  `big` tables run first, on the context
// itself; the others are assigned to the workers AHEAD of time, longest first onto the least loaded worker (cost[t] = estimated
// seconds) -- not pulled from a queue: a worker that gets the same tables on every call finds every block it needs in its own
// exact-size allocator cache from the second segment on (a dynamic queue kept hitting hipMalloc for ten or more calls: 37 -- 53 ms
// per 2^16-cycle segment depending on who had grabbed what), and the memory the lanes cache stays that of ONE assignment.
struct no_prep { void operator()(zkm_ctx*, const std::vector<size_t>&) const {} };
// prep(worker context, the jobs that worker will run, in order): called once on the worker's thread before its first job
template <class F, class P = no_prep>
static void run_on_lanes(zkm_ctx* c, const std::vector<size_t>& big, const std::vector<size_t>& small, const std::vector<double>& cost,
                         F&& fn, P&& prep = P()) {
    const size_t nlanes = small.size() >= 2 ? std::min<size_t>(std::max<size_t>(1, c->commit_lanes), small.size()) - 1 : 0;
    c->ensure_lanes(nlanes);
    std::vector<std::vector<size_t>> mine(nlanes + 1);
    {
        std::vector<double> load(nlanes + 1, 0.0);
        for (size_t t : big) load[0] += cost[t];
        std::vector<size_t> order(small);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
        for (size_t t : order) {
            size_t w = 0;
            for (size_t k = 1; k <= nlanes; k++)
                if (load[k] < load[w]) w = k;
            mine[w].push_back(t);
            load[w] += cost[t];
        }
    }
    std::atomic<bool> failed{false};
    std::vector<std::exception_ptr> errs(nlanes + 1);
    auto work = [&](zkm_ctx* w, size_t slot, bool take_big) {
        try {
            ZKM_HIP_CHECK(hipSetDevice(c->device));
            {
                std::vector<size_t> jobs;
                if (take_big) jobs = big;
                jobs.insert(jobs.end(), mine[slot].begin(), mine[slot].end());
                prep(w, jobs);
            }
            if (take_big)
                for (size_t t : big) {
                    if (failed.load()) break;                // a lane threw: do not commit the remaining big tables before reporting it
                    fn(w, t);
                }
            for (size_t t : mine[slot]) {
                if (failed.load()) break;                    // another worker threw: stop starting new tables
                fn(w, t);
            }
        } catch (...) {
            errs[slot] = std::current_exception();
            failed.store(true);
        }
    };
    std::vector<std::thread> threads;
    for (size_t k = 0; k < nlanes; k++) threads.emplace_back(work, c->lanes[k], k + 1, false);
    work(c, 0, true);
    for (auto& th : threads) th.join();
    for (auto& e : errs)
        if (e) std::rethrow_exception(e);
}

// ------------------------------------------------------------------ prove_with_traces for K segments in lock-step
// The segments of one call share the table list (ids, widths, column-set descriptions) and the cross-table lookups; heights may differ.
// For every table the segments are partitioned by height into GROUPS of at most ZKM_MAX_SEG: a group's trace, auxiliary and quotient
// commitments are stacked batches (zkm_internal.h), its CTL data / quotient / openings / FRI stages are one launch (or group of launches)
// for all its segments, and each of its transcript round trips brings all its segments' words down together.  Per segment the
// transcript is exactly prove_with_traces' (prover.rs:130-232, 234-438): all trace caps in table order, the public values, the CTL
// challenges, then the tables in order, each continuing the segment's challenger.  One segment = one group of one per table.
struct seg_io {
    const zkm_table_input* tables;   // ntables entries
    const uint64_t* pub;
    size_t npub;
    uint64_t* proofs;                // per-table blobs concatenated
    uint64_t* challenges;            // num_challenges (beta, gamma) pairs out
};
struct table_group {
    size_t t = 0;                    // table
    std::vector<size_t> segs;        // its segments (call order)
    unsigned log_n = 0;
    zkm_batch* commit = nullptr;
    zkm_batch* aux = nullptr;
    gl_t* d_traces = nullptr;        // stacked device copy of the trace values (segs.size() x ncols x n), or null: read in place
    zkm_ctx* d_owner = nullptr;      // ... from the allocator of the context (commit lane) that made it
    bool host = false, keep = false;
    hipEvent_t uploaded = nullptr;   // host-resident traces of a stack: recorded on the worker's copy stream behind their upload
    zkm_ctx* uploaded_on = nullptr;
};

// (seg_base: position of io[0] in the caller's call -- the waves of an oversized call; only error messages use it)
static void prove_segments_impl(zkm_ctx* c, const zkm_stark_config* cfg, size_t nseg, const seg_io* io, size_t ntables,
                                const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides, size_t nctls, size_t seg_base = 0) {
    std::vector<table_group> groups;
    auto drop_traces = [&]() {
        (void)hipStreamSynchronize(c->stream);
        for (table_group& g : groups) {
            if (g.d_traces) (g.d_owner ? g.d_owner : c)->release(g.d_traces);
            g.d_traces = nullptr;
        }
    };
    auto drop_all = [&]() {
        (void)hipStreamSynchronize(c->stream);
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        for (zkm_ctx* l : c->lanes) {
            (void)hipStreamSynchronize(l->stream);
            if (l->copy_stream) (void)hipStreamSynchronize(l->copy_stream);
        }
        for (table_group& g : groups)
            if (g.uploaded) {
                g.uploaded_on->event_pool.push_back(g.uploaded);
                g.uploaded = nullptr;
            }
        for (table_group& g : groups) {
            zkm_batch_free(g.commit);
            zkm_batch_free(g.aux);
            g.commit = g.aux = nullptr;
        }
        drop_traces();
    };
    struct background {            // lanes building auxiliary commitments behind the table proofs (below)
        std::mutex mu;
        std::condition_variable cv;
        std::vector<char> ready;
        std::vector<std::exception_ptr> errs;
        std::vector<std::thread> threads;
        std::atomic<bool> failed{false};
        void join() {
            for (auto& th : threads)
                if (th.joinable()) th.join();
            threads.clear();
        }
    } bg;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (nseg == 0) return;
        // The allocator's free lists are exact-size, and the sizes of a call's blocks are those of its stacks: a call of another
        // height would leave the blocks of BOTH heights cached (8 segments per call: 14.5 GB per context -- twice that after one
        // call of 7).  The cache is returned to the device when the number of segments per call changes (a few hipFree / hipMalloc
        // on the next call; calls of one height -- the steady state -- never get here).
        // TWO heights may stay cached side by side: the waves of one oversized call have floor and ceil(nseg / nwaves) segments, and a
        // caller that cuts k segments into calls (bench.py: 20 steps = 4, 4, 3, 3, 3, 3) alternates between two heights as well -- with
        // one remembered height every such call trimmed and re-hipMalloc'ed its blocks.  A third height trims.
        if (nseg != c->last_stack && nseg != c->prev_stack) {
            if (c->last_stack && c->prev_stack) {
                c->trim();
                c->last_stack = 0;
            }
            c->prev_stack = c->last_stack;
        } else if (nseg == c->prev_stack) {
            c->prev_stack = c->last_stack;
        }
        c->last_stack = nseg;
        const zkm_table_input* T0 = io[0].tables;
        for (size_t s = 0; s < nseg; s++) {
            if (!io[s].tables || !io[s].proofs || !io[s].challenges) throw std::runtime_error("zkm_prove_with_traces: null argument");
            for (size_t t = 0; t < ntables; t++) {
                const zkm_table_input& a = io[s].tables[t];
                if (a.table_id != T0[t].table_id || a.ncols != T0[t].ncols || a.ctl != T0[t].ctl)
                    throw std::runtime_error("zkm_prove_segments: the segments of one call must share the table list and its lookup description");
                if (a.ncols == 0 || a.log_n > 30) throw std::runtime_error("zkm_prove_with_traces: bad table shape");
                if (!a.trace && !a.columns) throw std::runtime_error("zkm_prove_with_traces: table without a trace");
            }
        }
        if (ntables == 12) {  // a whole AllStark segment: the transcript only matches the reference's in Table::all() order
            bool all = true, ordered = true;
            for (size_t t = 0; t < 12; t++) {
                int e = zkm_table_enum_index(T0[t].table_id);
                all = all && e >= 0;
                ordered = ordered && e == (int)t;
            }
            bool distinct = all;
            for (size_t a = 0; a < 12 && distinct; a++)
                for (size_t b = a + 1; b < 12; b++)
                    if (T0[a].table_id == T0[b].table_id) distinct = false;
            if (distinct && !ordered)
                throw std::runtime_error("zkm_prove_with_traces: the twelve tables must be given in the reference's Table enum order "
                                         "(all_stark.rs:96-110; see zkm_table_enum_index)");
        }
        std::vector<std::vector<size_t>> offs(nseg, std::vector<size_t>(ntables + 1));
        for (size_t s = 0; s < nseg; s++)
            if (!zkm_all_proof_words(cfg, io[s].tables, ntables, ctls, sides, nctls, offs[s].data()) && ntables)
                throw std::runtime_error("zkm_prove_with_traces: malformed cross-table lookups");
        // the structure of every table's CtlZData list (no challenges yet): the number of auxiliary columns bounds the stack height
        const auto tz0 = derive_zs(ntables, ctls, sides, nctls, cfg->num_challenges, nullptr);
        // groups: per table, segments of equal height, at most max_stack (and 65535 stacked columns: a grid dimension of the transforms)
        std::vector<std::vector<size_t>> where(ntables, std::vector<size_t>(nseg));     // group of (table, segment)
        std::vector<std::vector<size_t>> pos(ntables, std::vector<size_t>(nseg));       // ... and the segment's position in it
        for (size_t t = 0; t < ntables; t++) {
            const size_t widest = std::max<size_t>(T0[t].ncols, tz0[t].naux + zkm_num_lookup_columns(T0[t].table_id, cfg));
            size_t cap = std::min<size_t>(std::max<size_t>(1, c->max_stack), ZKM_MAX_SEG);
            cap = std::max<size_t>(1, std::min<size_t>(cap, 65535 / std::max<size_t>(1, widest)));
            std::vector<char> done(nseg, 0);
            for (size_t s = 0; s < nseg; s++) {
                if (done[s]) continue;
                table_group g;
                g.t = t;
                g.log_n = io[s].tables[t].log_n;
                for (size_t r = s; r < nseg && g.segs.size() < cap; r++)
                    if (!done[r] && io[r].tables[t].log_n == g.log_n) {
                        done[r] = 1;
                        where[t][r] = groups.size();
                        pos[t][r] = g.segs.size();
                        g.segs.push_back(r);
                    }
                groups.push_back(std::move(g));
            }
        }
        const size_t njobs = groups.size();
        // "compute all trace commitments" prover.rs:144-167
        std::vector<zkm_challenger> ch(nseg);
        for (auto& x : ch) zkm_challenger_init(&x);
        // Host-resident traces are uploaded once, with their commitment, and the device copy is reused for the table's CTL / lookup
        // columns -- as long as the copies kept this way stay within a quarter of the memory that is free now (all twelve commitments
        // -- coefficients + 4x LDE + digests -- are alive at the same time, and a full-size Keccak table alone is 20 GB of values).
        // A trace beyond that budget is dropped after its commitment and uploaded again when its table is proven.  (A group of several
        // segments always keeps its stacked copy: its kernels read ONE block.)
        size_t kept = 0, keep_limit = 0;
        bool have_limit = false;                       // (hipMemGetInfo is a driver round trip: only asked when a trace is host-resident)
        std::vector<size_t> big, small;
        {
            zkm_prof_scope st(c, "stage/compute all trace commitments");
            // The commitments do not depend on each other -- only the transcript does, and it starts after them (prover.rs:144-167 is a
            // plain loop; :182-185 observes the caps in table order): run_on_lanes.
            for (size_t j = 0; j < njobs; j++) {
                table_group& g = groups[j];
                const zkm_table_input& a = io[g.segs[0]].tables[g.t];
                const size_t bytes = (a.ncols << g.log_n) * sizeof(gl_t) * g.segs.size();
                g.host = false;
                for (size_t s : g.segs) {
                    const zkm_table_input& b = io[s].tables[g.t];
                    g.host = g.host || b.columns || !zkm_is_device_ptr(b.trace);   // (column pointers are gathered into one device block)
                }
                if (g.segs.size() > 1) {
                    g.keep = true;
                } else {
                    if (g.host && !have_limit) {
                        size_t free_b = 0, total_b = 0;
                        ZKM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
                        // (a quarter of what is free NOW -- shared with every other context of this process working on the same GPU, each of
                        // which asks the same question on its own: divided by their number)
                        keep_limit = free_b / 4 / (size_t)std::max(1, zkm_live_contexts());
                        have_limit = true;
                    }
                    g.keep = g.host && kept + bytes <= keep_limit;
                    if (g.keep) kept += bytes;
                }
                ((bytes << cfg->rate_bits) > ((size_t)1 << 30) ? big : small).push_back(j);
            }
            auto words = [&](size_t j) { return (T0[groups[j].t].ncols << groups[j].log_n) * groups[j].segs.size(); };
            std::sort(small.begin(), small.end(), [&](size_t a, size_t b) { return words(a) > words(b); });
            std::vector<double> cost(njobs, 0.0);
            for (size_t j = 0; j < njobs; j++)
                cost[j] = commit_cost_estimate(c, T0[groups[j].t].ncols, groups[j].log_n, cfg->rate_bits, groups[j].segs.size());
            run_on_lanes(c, big, small, cost, [&](zkm_ctx* w, size_t j) {
                table_group& g = groups[j];
                const size_t W = T0[g.t].ncols, n = (size_t)1 << g.log_n, G = g.segs.size();
                if (g.keep && !g.d_traces) {
                    g.d_traces = (gl_t*)w->alloc(G * W * n * sizeof(gl_t));
                    g.d_owner = w;
                }
                if (G == 1) {
                    const zkm_table_input& a = io[g.segs[0]].tables[g.t];
                    g.commit = zkm_batch_commit_values_keep(w, a.trace, W, g.log_n, cfg->rate_bits, cfg->cap_height, g.d_traces, a.columns);
                    return;
                }
                zkm_batch* b = new zkm_batch();
                b->ctx = w; b->ncols = W; b->nseg = G; b->log_n = g.log_n; b->rate_bits = cfg->rate_bits; b->cap_height = cfg->cap_height;
                g.commit = b;      // (owned from here on: freed on every exit path)
                if (g.uploaded) {  // its traces came up on the copy stream while the worker's earlier groups were transformed and hashed
                    ZKM_HIP_CHECK(hipStreamWaitEvent(w->stream, g.uploaded, 0));
                    zkm_launch_canon(w, g.d_traces, G * W * n);
                    zkm_batch_build(b, g.d_traces, true);
                    return;
                }
                bool any_cols = false;
                for (size_t s : g.segs) any_cols = any_cols || io[s].tables[g.t].columns;
                if (any_cols) {    // one pointer per column of every segment
                    std::vector<const uint64_t*> cols(G * W);
                    for (size_t k = 0; k < G; k++) {
                        const zkm_table_input& a = io[g.segs[k]].tables[g.t];
                        for (size_t i = 0; i < W; i++) cols[k * W + i] = a.columns ? a.columns[i] : a.trace + i * n;
                    }
                    zkm_batch_build(b, nullptr, true, g.d_traces, cols.data());
                } else {
                    std::vector<const uint64_t*> srcs(G);
                    for (size_t k = 0; k < G; k++) srcs[k] = io[g.segs[k]].tables[g.t].trace;
                    zkm_batch_build(b, nullptr, true, g.d_traces, nullptr, srcs.data());
                }
            }, [&](zkm_ctx* w, const std::vector<size_t>& jobs) {
                // Host-resident traces of the worker's stacks (the deployed input: generate_traces leaves them in host memory) go up on
                // the worker's COPY stream, all of them queued now, each followed by an event: the transforms and hashing of a group
                // overlap the uploads of the groups behind it (0.3 GB per 2^16-cycle segment over PCIe).  Pinned (zkm_host_alloc /
                // zkm_host_register) sources make the copies asynchronous; pageable ones are staged by the runtime inside the call.
                for (size_t j : jobs) {
                    table_group& g = groups[j];
                    if (!(g.host && g.keep && g.segs.size() > 1)) continue;
                    const size_t W = T0[g.t].ncols, n = (size_t)1 << g.log_n, G = g.segs.size();
                    if (!w->copy_stream) {
                        hipStream_t cs = nullptr;
                        ZKM_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                        std::lock_guard<std::mutex> lk(w->alloc_mu);
                        w->copy_stream = cs;
                    }
                    g.d_traces = (gl_t*)w->alloc(G * W * n * sizeof(gl_t));
                    g.d_owner = w;
                    for (size_t k = 0; k < G; k++) {
                        const zkm_table_input& a = io[g.segs[k]].tables[g.t];
                        gl_t* dst = g.d_traces + k * W * n;
                        if (a.columns)
                            for (size_t i = 0; i < W; i++)
                                ZKM_HIP_CHECK(hipMemcpyAsync(dst + i * n, a.columns[i], n * sizeof(gl_t), hipMemcpyDefault, w->copy_stream));
                        else
                            ZKM_HIP_CHECK(hipMemcpyAsync(dst, a.trace, W * n * sizeof(gl_t), hipMemcpyDefault, w->copy_stream));
                    }
                    g.uploaded = w->get_event();
                    g.uploaded_on = w;
                    ZKM_HIP_CHECK(hipEventRecord(g.uploaded, w->copy_stream));
                }
            });
            for (table_group& g : groups)      // (every upload has been waited for by its group's commitment: the events go back)
                if (g.uploaded) {
                    g.uploaded_on->event_pool.push_back(g.uploaded);
                    g.uploaded = nullptr;
                }
        }
        const size_t C4 = (size_t)4 << cfg->cap_height;
        const unsigned nch = cfg->num_challenges;
        for (size_t s = 0; s < nseg; s++) {
            for (size_t t = 0; t < ntables; t++)   // :182-185
                zkm_challenger_observe(&ch[s], groups[where[t][s]].commit->cap.data() + pos[t][s] * C4, C4);
            zkm_challenger_observe(&ch[s], io[s].pub, io[s].npub);  // :187 observe_public_values
            for (unsigned k = 0; k < nch; k++) {  // :190, beta then gamma (cross_table_lookup.rs:560-566)
                io[s].challenges[2 * k] = zkm_challenger_get(&ch[s]);
                io[s].challenges[2 * k + 1] = zkm_challenger_get(&ch[s]);
            }
        }
        std::vector<std::vector<table_zs>> tz(nseg);
        for (size_t s = 0; s < nseg; s++) tz[s] = derive_zs(ntables, ctls, sides, nctls, nch, io[s].challenges);
        // per group: its segments' CtlZData lists one after the other (same structure, own challenges), and the betas of the CTL
        // challenges as lookup challenges (prover.rs:468-474)
        std::vector<std::vector<zkm_ctl_z>> gzs(njobs);
        std::vector<std::vector<uint64_t>> glookup(njobs);
        for (size_t j = 0; j < njobs; j++)
            for (size_t s : groups[j].segs) {
                const table_zs& z = tz[s][groups[j].t];
                gzs[j].insert(gzs[j].end(), z.zs.begin(), z.zs.end());
                for (unsigned k = 0; k < nch; k++) glookup[j].push_back(io[s].challenges[2 * k]);
            }
        // "compute CTL data" :191-200 and each table's "compute auxiliary polynomials commitment" :511-522 (with its "compute lookup helper
        // columns" :475-493) depend on the CTL challenges only, not on the transcript of the table proofs: all tables' auxiliary
        // commitments are built now, side by side (run_on_lanes).  The transcript then observes them in table order, below.
        std::vector<double> aux_cost(njobs, 0.0);          // CTL data + lookup columns read the trace once; then an `naux`-column commitment
        for (size_t j = 0; j < njobs; j++) {
            const size_t t = groups[j].t;
            aux_cost[j] = commit_cost_estimate(c, tz0[t].naux + zkm_num_lookup_columns(T0[t].table_id, cfg), groups[j].log_n, cfg->rate_bits,
                                               groups[j].segs.size()) +
                          (double)((T0[t].ncols << groups[j].log_n) * groups[j].segs.size()) * 1e-10;
        }
        auto aux_job = [&](zkm_ctx* w, size_t j) {
            table_group& g = groups[j];
            const size_t t = g.t, G = g.segs.size();
            if (tz0[t].naux == 0) return;   // ("No CTL?" -- reported by prove_single_table in table order, prover.rs:509)
            const size_t n = (size_t)1 << g.log_n, W = T0[t].ncols;
            const size_t NL = zkm_num_lookup_columns(T0[t].table_id, cfg), A = NL + tz0[t].naux;
            ctl_dev_owner own;
            own.upload(w, T0[t].ctl, gzs[j].data(), tz0[t].ids.data(), tz0[t].zs.size(), false, W, G);
            const zkm_table_input& a0 = io[g.segs[0]].tables[t];
            const gl_t* d_trace = g.d_traces ? g.d_traces : a0.trace;
            zkm_scratch again(w, (!g.d_traces && g.host) ? W * n * sizeof(gl_t) : 8);
            if (!g.d_traces && g.host) {   // (one segment, over the keep budget: second upload)
                gl_t* d = again.as<gl_t>();
                if (a0.columns)
                    for (size_t i = 0; i < W; i++) ZKM_HIP_CHECK(hipMemcpyAsync(d + i * n, a0.columns[i], n * sizeof(gl_t), hipMemcpyDefault, w->stream));
                else
                    ZKM_HIP_CHECK(hipMemcpyAsync(d, a0.trace, W * n * sizeof(gl_t), hipMemcpyHostToDevice, w->stream));
                zkm_launch_canon(w, d, W * n);   // (as zkm_batch_build does for the copies it keeps)
                d_trace = d;
            }
            zkm_scratch d_all(w, G * A * n * sizeof(gl_t));   // per segment [lookup helper columns | CTL helper columns and Zs], prover.rs:497-508
            try {
                {
                    zkm_prof_scope st(w, "stage/compute CTL data");
                    zkm_ctl_data_device(w, own, d_trace, g.log_n, d_all.as<gl_t>() + NL * n, G, W * n, A * n);
                }
                if (NL) {
                    zkm_prof_scope st(w, "stage/compute lookup helper columns");
                    zkm_table_lookup_columns_device(w, T0[t].table_id, glookup[j].data(), nch, d_trace, n, d_all.as<gl_t>(), G, W * n, A * n);
                }
            } catch (const zkm_segment_error& e) {   // (a stack names the position in the group: the caller wants ITS segment and the table)
                if (nseg == 1 && seg_base == 0) throw std::runtime_error(e.what());
                throw std::runtime_error("segment " + std::to_string(seg_base + g.segs[e.seg]) + ", table " + std::to_string(t) + ": " + e.what());
            }
            zkm_prof_scope st(w, "stage/compute auxiliary polynomials commitment");
            zkm_batch* ab = new zkm_batch();
            ab->ctx = w; ab->ncols = A; ab->nseg = G; ab->log_n = g.log_n; ab->rate_bits = cfg->rate_bits; ab->cap_height = cfg->cap_height;
            g.aux = ab;            // (owned from here on: freed with the others on every exit path)
            zkm_batch_build(ab, d_all.as<uint64_t>(), true);
            w->sync();             // d_all / `again` go back to the lane's allocator when this returns
            // the device copy of the traces has served its purpose (commitment, CTL data, lookup columns): back to its owner's allocator
            // NOW, not after the last table proof (nothing reads it any more and w's stream is drained)
            if (g.d_traces) {
                (g.d_owner ? g.d_owner : c)->release(g.d_traces);
                g.d_traces = nullptr;
            }
        };
        auto prove_group = [&](size_t j) {   // "compute all proofs given commitments" :234-438: tables in order, one transcript per segment
            table_group& g = groups[j];
            const size_t t = g.t;
            try {
                if (!g.aux) throw std::runtime_error("No CTL? aux column count does not match the CTL description");  // prover.rs:509
                std::vector<zkm_challenger> local(g.segs.size());
                std::vector<zkm_challenger*> chs(g.segs.size());
                std::vector<uint64_t*> proofs(g.segs.size());
                for (size_t k = 0; k < g.segs.size(); k++) {
                    local[k] = ch[g.segs[k]];
                    chs[k] = &local[k];
                    proofs[k] = io[g.segs[k]].proofs + offs[g.segs[k]][t];
                }
                zkm_prove_single_table_aux(c, T0[t].table_id, cfg, T0[t].ncols, g.log_n, g.commit, g.aux, tz0[t].naux, T0[t].ctl, gzs[j].data(),
                                           tz0[t].ids.data(), tz0[t].zs.size(), glookup[j].data(), chs, proofs);
                for (size_t k = 0; k < g.segs.size(); k++) ch[g.segs[k]] = local[k];
            } catch (const zkm_segment_error& e) {
                throw std::runtime_error("segment " + std::to_string(seg_base + g.segs[e.seg]) + ", table " + std::to_string(t) + ": " + e.what());
            } catch (const std::exception& e) {
                if (nseg == 1 && seg_base == 0) throw std::runtime_error("table " + std::to_string(t) + ": " + e.what());
                // a stacked group fails as a whole: name its members (positions in the caller's call) next to the table
                std::string who = g.segs.size() == 1 ? "segment " : "segments ";
                for (size_t k = 0; k < g.segs.size(); k++) who += (k ? "," : "") + std::to_string(seg_base + g.segs[k]);
                throw std::runtime_error(who + ", table " + std::to_string(t) + ": " + e.what());
            }
        };
        // (groups are numbered table by table: job order is proof order)
        const size_t bg_lanes =
            (big.empty() && c->aux_pipeline && njobs >= 3) ? std::min<size_t>(std::max<size_t>(1, c->commit_lanes), njobs - 1) - 1 : 0;
        if (bg_lanes == 0) {
            run_on_lanes(c, big, small, aux_cost, aux_job);
            drop_traces();         // the device copies of the traces have served their purpose (commitment, CTL data, lookup columns)
            for (size_t j = 0; j < njobs; j++) prove_group(j);
        } else {
            // A segment of short tables (no table fills the GPU on its own): the table proofs are ONE chain of latency-bound steps on
            // the context's stream, and the auxiliary commitment of a LATER table is not needed before that table's turn -- the lanes
            // build them behind the proofs of the earlier tables (dealt to the lanes in proof order, the same deal every segment, so
            // that each lane finds its blocks in its own allocator cache; the context builds table 0's and starts proving).  The
            // transcript is what it was: every cap is observed inside its table's proof, in table order.
            c->ensure_lanes(bg_lanes);
            bg.ready.assign(njobs, 0);
            bg.errs.assign(bg_lanes, nullptr);
            struct stop_and_join {     // (an exception on this thread unwinds what the lanes refer to: they are stopped and joined first)
                background& b;
                ~stop_and_join() {
                    if (!b.threads.empty()) b.failed.store(true);
                    b.join();
                }
            } guard{bg};
            for (size_t k = 0; k < bg_lanes; k++)
                bg.threads.emplace_back([&, k]() {
                    zkm_ctx* w = c->lanes[k];
                    try {
                        ZKM_HIP_CHECK(hipSetDevice(c->device));
                        for (size_t j = 1 + k; j < njobs; j += bg_lanes) {
                            if (bg.failed.load()) break;
                            aux_job(w, j);
                            std::lock_guard<std::mutex> g(bg.mu);
                            bg.ready[j] = 1;
                            bg.cv.notify_all();
                        }
                    } catch (...) {
                        bg.errs[k] = std::current_exception();
                        bg.failed.store(true);
                    }
                    std::lock_guard<std::mutex> g(bg.mu);      // (whatever happened: nobody waits for this lane any more)
                    for (size_t j = 1 + k; j < njobs; j += bg_lanes) bg.ready[j] = 1;
                    bg.cv.notify_all();
                });
            aux_job(c, 0);
            for (size_t j = 0; j < njobs; j++) {
                if (j) {
                    zkm_prof_scope st(c, "stage/wait for auxiliary polynomials commitment");
                    std::unique_lock<std::mutex> g(bg.mu);
                    bg.cv.wait(g, [&] { return bg.ready[j] != 0; });
                }
                if (bg.failed.load()) break;
                prove_group(j);
            }
            bg.join();
            for (auto& e : bg.errs)
                if (e) std::rethrow_exception(e);
            drop_traces();
        }
    } catch (...) {
        bg.failed.store(true);
        bg.join();
        drop_all();
        throw;
    }
    drop_all();
}

// A call may hold more segments than fit in HBM together (every commitment of every segment of a lock-step call is alive until its
// table has been proven): the segments are proven in consecutive WAVES whose estimated footprint -- per table the trace values,
// coefficients, 4x LDE and digests of the trace, auxiliary and quotient batches -- stays within 80 % of the blocks its allocator has
// cached plus the memory that is free right now.  One segment always goes (a single
// segment that does not fit fails in the allocator, as it always did).
static void prove_segments_waves(zkm_ctx* c, const zkm_stark_config* cfg, size_t nseg, const seg_io* io, size_t ntables,
                                 const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides, size_t nctls, size_t seg_base = 0) {
    if (nseg <= 1) {
        prove_segments_impl(c, cfg, nseg, io, ntables, ctls, sides, nctls, seg_base);
        return;
    }
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    size_t free_b = 0, total_b = 0, live = 0, cached = 0;
    ZKM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    zkm_ctx_memory(c, &live, &cached);
    // (what this context may count on: the blocks its own allocator has cached -- after the first call they hold a whole wave -- and what
    // is free NOW.  Other contexts of the process look at the same free memory: the caller sizes contexts x segments per call for the
    // GPU (1.8 GB per 2^16-cycle segment in flight); this is the safety net of ONE oversized call, not an arbiter between contexts --
    // dividing the free memory among them starved the contexts that warmed up last and left their allocators with blocks of two wave
    // sizes: 12 x 8 fell from 107 to 72 segments/s.)
    const double budget = c->segments_memory_budget ? (double)c->segments_memory_budget : 0.8 * ((double)cached + (double)free_b);
    const auto tz0 = derive_zs(ntables, ctls, sides, nctls, cfg->num_challenges, nullptr);
    auto footprint = [&](const seg_io& s) {
        double words = 0;
        for (size_t t = 0; t < ntables && s.tables; t++) {
            const double n = (double)((size_t)1 << std::min<unsigned>(s.tables[t].log_n, 40)), W = (double)s.tables[t].ncols;
            const double A = (double)(tz0[t].naux + zkm_num_lookup_columns(s.tables[t].table_id, cfg)), Q = 2.0 * cfg->num_challenges;
            words += W * n + (W + A + Q) * n * (1.0 + (double)(1u << cfg->rate_bits)) + 3.0 * 8.0 * n * (double)(1u << cfg->rate_bits);
        }
        return words * 8.0;
    };
    double total = 0;
    for (size_t s = 0; s < nseg; s++) total += footprint(io[s]);
    // even waves: nseg mod nwaves of them hold ceil(nseg / nwaves) segments, the others floor -- two stack heights at most, which is
    // what the exact-size allocator keeps cached side by side (prove_segments_impl), the same ones call after call
    const size_t nwaves = (size_t)std::min<double>((double)nseg, std::max(1.0, std::ceil(total / std::max(budget, 1.0))));
    const size_t lo = nseg / nwaves, extra = nseg % nwaves;
    if (getenv("ZKM_DEBUG_WAVES"))
        fprintf(stderr, "zkm waves: %zu segments, %.2f GB in all, budget %.2f GB (free %.2f, cached %.2f) -> %zu wave(s): %zu of %zu, %zu of %zu\n", nseg,
                total / 1e9, budget / 1e9, free_b / 1e9, cached / 1e9, nwaves, extra, lo + 1, nwaves - extra, lo);
    for (size_t w = 0, s0 = 0; w < nwaves; w++) {
        const size_t k = lo + (w < extra ? 1 : 0);
        prove_segments_impl(c, cfg, k, io + s0, ntables, ctls, sides, nctls, seg_base + s0);   // (errors name positions in the CALL, not in the wave)
        s0 += k;
    }
}

extern "C" {

int zkm_prove_with_traces(zkm_ctx* c, const zkm_stark_config* cfg, const zkm_table_input* tables, size_t ntables,
                          const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides, size_t nctls, const uint64_t* pub,
                          size_t npub, uint64_t* proofs, uint64_t* challenges, char** err) {
    try {
        if (!c || !cfg || (!tables && ntables)) throw std::runtime_error("zkm_prove_with_traces: null argument");
        seg_io io{tables, pub, npub, proofs, challenges};
        prove_segments_impl(c, cfg, 1, &io, ntables, ctls, sides, nctls);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    } catch (...) {
        return fail(err, "zkm_prove_with_traces: unknown error");
    }
    return 0;
}

// traces[s][t] (one block per table) or columns[s][t][i] (one pointer per column) -- exactly one of the two is non-null
// (zkm_internal.h: the pool's workers call it with the position of their group in the pool call as seg_base -- error messages only)
int zkm_prove_segments_entry(const char* what, zkm_ctx* c, const zkm_stark_config* cfg, size_t nseg, const uint64_t* const* const* traces,
                             const uint64_t* const* const* const* columns, const unsigned* const* log_n, const uint64_t* const* pub,
                             const size_t* npub, uint64_t* const* proofs, uint64_t* const* challenges, char** err, size_t seg_base) {
    try {
        if (!c || !cfg || (!traces && !columns) || !log_n || !proofs || !challenges) throw std::runtime_error(std::string(what) + ": null argument");
        std::vector<std::vector<zkm_table_input>> tables(nseg, std::vector<zkm_table_input>(12));
        std::vector<seg_io> io(nseg);
        for (size_t s = 0; s < nseg; s++) {
            if ((traces && !traces[s]) || (columns && !columns[s]) || !log_n[s] || !proofs[s] || !challenges[s])
                throw std::runtime_error(std::string(what) + ": null segment");
            for (int t = 0; t < 12; t++) {
                if (columns && !columns[s][t]) throw std::runtime_error(std::string(what) + ": null table");
                tables[s][t] = zkm_table_input{AS_TABLE_IDS[t], traces ? traces[s][t] : nullptr, AS_TABLE_WIDTH[t], log_n[s][t], &AS_CTL_TABLES[t],
                                               columns ? columns[s][t] : nullptr};
            }
            io[s] = seg_io{tables[s].data(), pub ? pub[s] : nullptr, npub ? npub[s] : 0, proofs[s], challenges[s]};
            if (io[s].npub && !io[s].pub) throw std::runtime_error(std::string(what) + ": null public values");
        }
        prove_segments_waves(c, cfg, nseg, io.data(), 12, AS_CTLS, AS_SIDES, AS_NCTLS, seg_base);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    } catch (...) {
        return fail(err, std::string(what) + ": unknown error");
    }
    return 0;
}

int zkm_prove_segments(zkm_ctx* c, const zkm_stark_config* cfg, size_t nseg, const uint64_t* const* const* traces, const unsigned* const* log_n,
                       const uint64_t* const* pub, const size_t* npub, uint64_t* const* proofs, uint64_t* const* challenges, char** err) {
    return zkm_prove_segments_entry("zkm_prove_segments", c, cfg, nseg, traces, nullptr, log_n, pub, npub, proofs, challenges, err, 0);
}

int zkm_prove_segments_columns(zkm_ctx* c, const zkm_stark_config* cfg, size_t nseg, const uint64_t* const* const* const* columns,
                               const unsigned* const* log_n, const uint64_t* const* pub, const size_t* npub, uint64_t* const* proofs,
                               uint64_t* const* challenges, char** err) {
    return zkm_prove_segments_entry("zkm_prove_segments_columns", c, cfg, nseg, nullptr, columns, log_n, pub, npub, proofs, challenges, err, 0);
}

}  // extern "C"
