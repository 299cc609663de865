// ctl_dev.h -- device-side interpretation of the cross-table-lookup description (include/zkm_hip.h).
//
// Restates Column::eval_with_next / eval_table (cross_table_lookup.rs:292-333), Filter::eval_filter (:64-79),
// GrandProductChallenge::combine (:494-504) and eval_cross_table_lookup_checks (:1006-1150) for one lane = one row
// (CTL data generation) or one lane = one quotient-domain point (constraint checks).  The descriptor arrays are
// wave-uniform, so they come through the scalar cache; the per-lane work is the column gathers.
#pragma once
#include <vector>

#include "zkm_internal.h"

struct ctl_dev {
    const zkm_column* columns;
    const uint32_t* term_col;
    const uint64_t* term_coeff;
    const zkm_colset* colsets;
    const uint32_t* filter_idx;
    const zkm_ctl_z* zs;
    const uint32_t* colset_ids;
    uint32_t nzs, total_helpers;
};

// `lv` points at this lane's element of column 0, columns are `stride` words apart; the next-row element of a
// column sits `next_delta` words after the local one.  next_ok = false reproduces eval_table's last-row rule.
__device__ __forceinline__ gl_t ctl_eval_column(const ctl_dev& d, uint32_t ci, const gl_t* __restrict__ lv, size_t stride,
                                                ptrdiff_t next_delta, bool next_ok) {
    const zkm_column c = d.columns[ci];
    gl_t acc = 0;
    for (uint32_t k = 0; k < c.n_local; k++) acc = gl_add(acc, gl_mul(lv[(size_t)d.term_col[c.term_off + k] * stride], d.term_coeff[c.term_off + k]));
    if (next_ok)
        for (uint32_t k = 0; k < c.n_next; k++) {
            uint32_t o = c.term_off + c.n_local + k;
            acc = gl_add(acc, gl_mul(lv[(size_t)d.term_col[o] * stride + next_delta], d.term_coeff[o]));
        }
    return gl_add(acc, c.constant);
}
__device__ __forceinline__ gl_t ctl_eval_filter(const ctl_dev& d, const zkm_colset& cs, const gl_t* __restrict__ lv, size_t stride,
                                                ptrdiff_t next_delta, bool next_ok) {
    if (!cs.has_filter) return 1;
    gl_t acc = 0;
    for (uint32_t k = 0; k < cs.nprod; k++)
        acc = gl_add(acc, gl_mul(ctl_eval_column(d, d.filter_idx[cs.prod_off + 2 * k], lv, stride, next_delta, next_ok),
                                 ctl_eval_column(d, d.filter_idx[cs.prod_off + 2 * k + 1], lv, stride, next_delta, next_ok)));
    for (uint32_t k = 0; k < cs.nconst; k++) acc = gl_add(acc, ctl_eval_column(d, d.filter_idx[cs.const_off + k], lv, stride, next_delta, next_ok));
    return acc;
}
__device__ __forceinline__ gl_t ctl_combine(const ctl_dev& d, const zkm_colset& cs, gl_t beta, gl_t gamma, const gl_t* __restrict__ lv,
                                            size_t stride, ptrdiff_t next_delta, bool next_ok) {
    gl_t acc = 0;
    for (uint32_t k = cs.ncols; k-- > 0;) acc = gl_add(gl_mul(acc, beta), ctl_eval_column(d, cs.col_off + k, lv, stride, next_delta, next_ok));
    return gl_add(acc, gamma);
}

// host-side owner of the device copy of a description
struct ctl_dev_owner {
    zkm_ctx* c = nullptr;
    void* blob = nullptr;
    std::vector<zkm_ctl_z> h_zs;  // host copy of the CtlZData list (launch planning)
    ctl_dev d{};
    size_t naux = 0;
    // lookup_mode: logUp lookups keep a helper column even for a single looking column (lookup.rs:34-38)
    // nseg > 1: zs = nseg lists of nzs entries (the same structure, each segment's own challenges); kernels index list blockIdx.z
    void upload(zkm_ctx* ctx, const zkm_ctl_table* t, const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs, bool lookup_mode = false,
                size_t trace_ncols = 0, size_t nseg = 1);  // trace_ncols != 0: every term_col must be < trace_ncols
    ~ctl_dev_owner();
};
