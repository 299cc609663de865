/* oracle/ctl.c -- cross-table lookup data (a3), the multiset test utility, and prove_with_traces / verify_proof (a1).
 *
 * TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restates (all under /root/reference/prover/src):
 *   Column::eval_table             cross_table_lookup.rs:313-333 (next-row terms count as 0 on the last row)
 *   Filter::eval_table             cross_table_lookup.rs:106-117
 *   get_helper_cols                cross_table_lookup.rs:709-797 (dummy 1 before inversion, zeroed afterwards; filters must be 0/1)
 *   partial_sums                   cross_table_lookup.rs:841-872 (upside-down running sum; helpers dropped for a single column set)
 *   cross_table_lookup_data        cross_table_lookup.rs:634-703 (zs order: per CTL, per challenge, looking groups then looked)
 *   check_ctls                     cross_table_lookup.rs:1486-1581
 *   prove_with_traces              prover.rs:130-232; prove_with_commitments :234-438 (tables in order, one transcript)
 *   verify_proof                   verifier.rs:27-176; verify_cross_table_lookups cross_table_lookup.rs:1415-1452
 */
#include <stdlib.h>
#include <string.h>
#include "zkm_oracle.h"
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include "gl.h"

static gl_t col_eval_table(const zko_ctl_table* t, uint32_t ci, const uint64_t* trace, size_t n, size_t row) {
    const zko_column* c = &t->columns[ci];
    gl_t acc = 0;
    for (uint32_t k = 0; k < c->n_local; k++)
        acc = gl_add(acc, gl_mul(trace[(size_t)t->term_col[c->term_off + k] * n + row], t->term_coeff[c->term_off + k]));
    acc = gl_add(acc, c->constant);
    if (c->n_next && row + 1 < n)
        for (uint32_t k = 0; k < c->n_next; k++) {
            uint32_t o = c->term_off + c->n_local + k;
            acc = gl_add(acc, gl_mul(trace[(size_t)t->term_col[o] * n + row + 1], t->term_coeff[o]));
        }
    return acc;
}
static gl_t filter_eval_table(const zko_ctl_table* t, const zko_colset* cs, const uint64_t* trace, size_t n, size_t row) {
    if (!cs->has_filter) return 1;
    gl_t acc = 0;
    for (uint32_t k = 0; k < cs->nprod; k++)
        acc = gl_add(acc, gl_mul(col_eval_table(t, t->filter_idx[cs->prod_off + 2 * k], trace, n, row),
                                 col_eval_table(t, t->filter_idx[cs->prod_off + 2 * k + 1], trace, n, row)));
    for (uint32_t k = 0; k < cs->nconst; k++) acc = gl_add(acc, col_eval_table(t, t->filter_idx[cs->const_off + k], trace, n, row));
    return acc;
}

/* one term column: filter ? 1 / (sum_i eval_i beta^i + gamma) : 0 */
static int inverse_terms(const zko_ctl_table* t, const zko_colset* cs, gl_t beta, gl_t gamma, const uint64_t* trace, size_t n, gl_t* out) {
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (size_t d = 0; d < n; d++) {
        gl_t f = filter_eval_table(t, cs, trace, n, d);
        if (f == 1) {
            gl_t acc = 0;
            for (uint32_t k = cs->ncols; k-- > 0;) acc = gl_add(gl_mul(acc, beta), col_eval_table(t, cs->col_off + k, trace, n, d));
            out[d] = gl_inv(gl_add(acc, gamma));
        } else {
            if (f != 0) bad = 1; /* "Non-binary filter?" */
            out[d] = 0;
        }
    }
    return bad;
}

void zko_ctl_data(const zko_ctl_table* t, const zko_ctl_z* zs, const uint32_t* colset_ids, size_t nzs, const uint64_t* trace,
                  size_t ncols, unsigned log_n, uint64_t* aux) {
    (void)ncols;
    size_t n = (size_t)1 << log_n, total_helpers = 0, hstart = 0;
    for (size_t i = 0; i < nzs; i++) total_helpers += zs[i].num_helpers;
    gl_t* term = (gl_t*)malloc(sizeof(gl_t) * n);
    gl_t* hsum = (gl_t*)malloc(sizeof(gl_t) * n);
    for (size_t i = 0; i < nzs; i++) {
        const zko_ctl_z* z = &zs[i];
        const uint32_t* ids = colset_ids + z->colset_off;
        memset(hsum, 0, sizeof(gl_t) * n);
        for (uint32_t j = 0; 2 * j < z->ncolsets; j++) {
            gl_t* h = z->num_helpers ? aux + (hstart + j) * n : NULL;
            for (uint32_t e = 0; e < 2 && 2 * j + e < z->ncolsets; e++) {
                inverse_terms(t, &t->colsets[ids[2 * j + e]], z->beta, z->gamma, trace, n, term);
                for (size_t d = 0; d < n; d++) {
                    hsum[d] = gl_add(hsum[d], term[d]);
                    if (h) h[d] = e ? gl_add(h[d], term[d]) : term[d];
                }
            }
        }
        /* Z[n-1] = sum_h h[n-1]; Z[i] = Z[i+1] + sum_h h[i] */
        gl_t* zc = aux + (total_helpers + i) * n;
        gl_t acc = 0;
        for (size_t d = n; d-- > 0;) { acc = gl_add(acc, hsum[d]); zc[d] = acc; }
        hstart += z->num_helpers;
    }
    free(term);
    free(hsum);
}

/* LogicStark witness: Operation::into_row logic.rs:122-142, zero padding rows :155-173 */
void zko_logic_trace(const uint32_t* ops, size_t nops, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    memset(out, 0, sizeof(uint64_t) * 69 * n);
    for (size_t r = 0; r < nops && r < n; r++) {
        uint32_t op = ops[3 * r], a = ops[3 * r + 1], b = ops[3 * r + 2];
        uint32_t res = op == 0 ? (a & b) : op == 1 ? (a | b) : op == 2 ? (a ^ b) : ~(a | b);
        out[(size_t)op * n + r] = 1;
        for (int i = 0; i < 32; i++) {
            out[(size_t)(4 + i) * n + r] = (a >> i) & 1;
            out[(size_t)(36 + i) * n + r] = (b >> i) & 1;
        }
        out[(size_t)68 * n + r] = res;
    }
}

/* MemoryStark::generate_trace (memory/memory_stark.rs:123-248): sort by (context, segment, virt, timestamp), fill_gaps, pad
 * with filter-0 reads of the last operation, re-sort, first-change flags and range-check deltas, COUNTER, FREQUENCIES.
 * ops = nops x 6 words {context, segment, virt, timestamp, is_read, value}; the R0 write rule of into_row (:68-76) applies.
 * Returns the natural row count (power of two) or 0 when it exceeds 2^log_n or a range check fails. */
typedef struct { uint64_t ctx, seg, virt, ts, is_read, value, filter, seq; } mem_op;
static int mem_cmp(const void* a, const void* b) {
    const mem_op *x = (const mem_op*)a, *y = (const mem_op*)b;
    if (x->ctx != y->ctx) return x->ctx < y->ctx ? -1 : 1;
    if (x->seg != y->seg) return x->seg < y->seg ? -1 : 1;
    if (x->virt != y->virt) return x->virt < y->virt ? -1 : 1;
    if (x->ts != y->ts) return x->ts < y->ts ? -1 : 1;
    return x->seq < y->seq ? -1 : x->seq > y->seq; /* sort_by_key is stable */
}
static size_t next_pow2(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }
size_t zko_memory_trace(const uint64_t* ops_in, size_t nops, unsigned log_n, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    if (nops == 0) return 0; /* "No memory ops?" :225 */
    size_t cap = 2 * next_pow2(nops) + 16, cnt = nops;
    mem_op* ops = (mem_op*)malloc(sizeof(mem_op) * cap);
    for (size_t i = 0; i < nops; i++) {
        const uint64_t* o = ops_in + 6 * i;
        ops[i] = (mem_op){o[0], o[1], o[2], o[3], o[4] != 0, o[5], 1, i};
    }
    qsort(ops, cnt, sizeof(mem_op), mem_cmp);
    /* fill_gaps :180-221 iterates over a clone of the sorted list and appends dummy reads */
    size_t max_rc = next_pow2(cnt) - 1, base = cnt;
    for (size_t i = 0; i + 1 < base; i++) {
        mem_op curr = ops[i], next = ops[i + 1];
        if (curr.ctx != next.ctx || curr.seg != next.seg) continue;
        if (curr.virt != next.virt) {
            while (next.virt - curr.virt - 1 > max_rc) {
                curr.virt += max_rc + 1;
                curr.ts = 0; curr.value = 0; curr.is_read = 1; curr.filter = 0;
                if (cnt == cap) { cap *= 2; ops = (mem_op*)realloc(ops, sizeof(mem_op) * cap); }
                curr.seq = cnt;
                ops[cnt++] = curr;
            }
        } else {
            while (next.ts - curr.ts > max_rc) {
                curr.ts += max_rc;
                curr.is_read = 1; curr.filter = 0;
                if (cnt == cap) { cap *= 2; ops = (mem_op*)realloc(ops, sizeof(mem_op) * cap); }
                curr.seq = cnt;
                ops[cnt++] = curr;
            }
        }
    }
    /* pad_memory_ops :223-241: repeat the last operation of the list (filter 0, read) up to a power of two (and to 2^log_n) */
    size_t natural = next_pow2(cnt);
    if (natural > n) { free(ops); return 0; }
    if (n + 1 > cap) { cap = n + 1; ops = (mem_op*)realloc(ops, sizeof(mem_op) * cap); }
    mem_op pad = ops[cnt - 1];
    pad.filter = 0; pad.is_read = 1;
    while (cnt < n) { pad.seq = cnt; ops[cnt++] = pad; }
    qsort(ops, cnt, sizeof(mem_op), mem_cmp);
    memset(out, 0, sizeof(uint64_t) * 13 * n);
    int bad = 0;
    for (size_t i = 0; i < n; i++) {
        const mem_op* o = &ops[i];
        uint64_t value = (!o->is_read && o->ctx == 0 && o->seg == 4 && o->virt == 0) ? 0 : (uint32_t)o->value;
        out[0 * n + i] = o->filter; out[1 * n + i] = o->ts; out[2 * n + i] = o->is_read; out[3 * n + i] = o->ctx;
        out[4 * n + i] = o->seg; out[5 * n + i] = o->virt; out[6 * n + i] = value;
        if (i + 1 < n) { /* generate_first_change_flags_and_rc :83-130 */
            const mem_op* x = &ops[i + 1];
            int cfc = o->ctx != x->ctx, sfc = o->seg != x->seg && !cfc, vfc = o->virt != x->virt && !sfc && !cfc;
            out[7 * n + i] = cfc; out[8 * n + i] = sfc; out[9 * n + i] = vfc;
            gl_t rc = cfc ? gl_sub(gl_sub(x->ctx, o->ctx), 1) : sfc ? gl_sub(gl_sub(x->seg, o->seg), 1)
                      : vfc ? gl_sub(gl_sub(x->virt, o->virt), 1) : gl_sub(x->ts, o->ts);
            out[10 * n + i] = rc;
            if (rc >= n) bad = 1;
        }
        out[11 * n + i] = i; /* COUNTER :161 */
    }
    if (!bad)
        for (size_t i = 0; i < n; i++) out[12 * n + out[10 * n + i]] += 1; /* FREQUENCIES :163-166 */
    free(ops);
    return bad ? 0 : natural;
}

/* lookup_helper_columns lookup.rs:46-124: GrandProductChallenge{beta: 1, gamma: challenge} (:70-73), helper columns by
 * get_helper_cols, table inverse :100-105, forward running sum Z with Z[0] = 0 (:111-121) */
void zko_lookup_helper_columns(const zko_ctl_table* t, const uint32_t* colset_ids, size_t nlookup, uint32_t table_col, uint32_t freq_col,
                               uint64_t challenge, const uint64_t* trace, size_t ncols, unsigned log_n, uint64_t* out) {
    (void)ncols;
    size_t n = (size_t)1 << log_n, nh = (nlookup + 1) / 2;
    gl_t* term = (gl_t*)malloc(sizeof(gl_t) * n);
    gl_t* hsum = (gl_t*)calloc(n, sizeof(gl_t));
    for (size_t j = 0; j < nh; j++)
        for (size_t e = 0; e < 2 && 2 * j + e < nlookup; e++) {
            inverse_terms(t, &t->colsets[colset_ids[2 * j + e]], 1, challenge, trace, n, term);
            for (size_t d = 0; d < n; d++) {
                out[j * n + d] = e ? gl_add(out[j * n + d], term[d]) : term[d];
                hsum[d] = gl_add(hsum[d], term[d]);
            }
        }
    gl_t* z = out + nh * n;
    z[0] = 0;
    for (size_t i = 0; i + 1 < n; i++) {
        gl_t tinv = gl_inv(gl_add(challenge, col_eval_table(t, table_col, trace, n, i)));
        gl_t x = gl_sub(hsum[i], gl_mul(col_eval_table(t, freq_col, trace, n, i), tinv));
        z[i + 1] = gl_add(z[i], x);
    }
    free(term);
    free(hsum);
}

/* ---- per-table CtlZData lists of a set of cross-table lookups (cross_table_lookup_data order) ---- */
typedef struct { zko_ctl_z* zs; uint32_t* ids; size_t nzs, nids, naux; } table_zs_t;

static void derive_zs(size_t ntables, const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls, size_t nch,
                      const uint64_t* challenges /* nch x (beta,gamma) or NULL */, table_zs_t* out) {
    for (size_t t = 0; t < ntables; t++) {
        out[t].zs = (zko_ctl_z*)calloc(nctls * nch * 2 + 1, sizeof(zko_ctl_z));
        size_t cap = 1;
        for (size_t c = 0; c < nctls; c++) cap += (ctls[c].nlooking + 1) * nch;
        out[t].ids = (uint32_t*)calloc(cap, sizeof(uint32_t));
        out[t].nzs = out[t].nids = out[t].naux = 0;
    }
    for (size_t c = 0; c < nctls; c++) {
        const zko_ctl_side* lk = sides + ctls[c].looking_off;
        for (size_t ch = 0; ch < nch; ch++) {
            uint64_t beta = challenges ? challenges[2 * ch] : 0, gamma = challenges ? challenges[2 * ch + 1] : 0;
            for (uint32_t i = 0; i < ctls[c].nlooking;) { /* consecutive runs of the same table (itertools group_by) */
                uint32_t j = i;
                while (j < ctls[c].nlooking && lk[j].table == lk[i].table) j++;
                table_zs_t* o = &out[lk[i].table];
                zko_ctl_z* z = &o->zs[o->nzs++];
                z->ncolsets = j - i; z->colset_off = (uint32_t)o->nids; z->beta = beta; z->gamma = gamma;
                z->num_helpers = (j - i) > 1 ? (j - i + 1) / 2 : 0;
                for (uint32_t k = i; k < j; k++) o->ids[o->nids++] = lk[k].colset;
                o->naux += z->num_helpers + 1;
                i = j;
            }
            table_zs_t* o = &out[ctls[c].looked.table];
            zko_ctl_z* z = &o->zs[o->nzs++];
            z->ncolsets = 1; z->colset_off = (uint32_t)o->nids; z->num_helpers = 0; z->beta = beta; z->gamma = gamma;
            o->ids[o->nids++] = ctls[c].looked.colset;
            o->naux += 1;
        }
    }
}
static void free_zs(table_zs_t* z, size_t ntables) {
    for (size_t t = 0; t < ntables; t++) { free(z[t].zs); free(z[t].ids); }
}

/* ---- check_ctls: looking multiset == looked multiset ---- */
typedef struct { gl_t* v; size_t w; } rowset_t;
static size_t g_w;
static int row_cmp(const void* a, const void* b) {
    const gl_t *x = (const gl_t*)a, *y = (const gl_t*)b;
    for (size_t i = 0; i < g_w; i++) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}
static size_t collect_rows(const zko_table_input* tab, const zko_ctl_side* s, gl_t** buf, size_t* cap, size_t cnt, size_t w, int* bad) {
    const zko_ctl_table* t = tab[s->table].ctl;
    const zko_colset* cs = &t->colsets[s->colset];
    size_t n = (size_t)1 << tab[s->table].log_n;
    for (size_t d = 0; d < n; d++) {
        gl_t f = filter_eval_table(t, cs, tab[s->table].trace, n, d);
        if (f == 1) {
            if ((cnt + 1) * w > *cap) { *cap = (*cap) * 2 + w * 64; *buf = (gl_t*)realloc(*buf, sizeof(gl_t) * (*cap)); }
            for (size_t k = 0; k < w; k++) (*buf)[cnt * w + k] = col_eval_table(t, cs->col_off + (uint32_t)k, tab[s->table].trace, n, d);
            cnt++;
        } else if (f != 0) {
            *bad = 1;
        }
    }
    return cnt;
}
int zko_check_ctls(const zko_table_input* tables, size_t ntables, const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls) {
    (void)ntables;
    for (size_t c = 0; c < nctls; c++) {
        const zko_ctl_side* lk = sides + ctls[c].looking_off;
        size_t w = tables[ctls[c].looked.table].ctl->colsets[ctls[c].looked.colset].ncols;
        gl_t *a = NULL, *b = NULL;
        size_t ca = 0, cb = 0, na = 0, nb = 0;
        int bad = 0;
        for (uint32_t i = 0; i < ctls[c].nlooking; i++) {
            if (tables[lk[i].table].ctl->colsets[lk[i].colset].ncols != w) return 100 + (int)c;
            na = collect_rows(tables, &lk[i], &a, &ca, na, w, &bad);
        }
        nb = collect_rows(tables, &ctls[c].looked, &b, &cb, nb, w, &bad);
        int rc = 0;
        if (bad) rc = 200 + (int)c;
        else if (na != nb) rc = 300 + (int)c;
        else {
            g_w = w;
            qsort(a, na, sizeof(gl_t) * w, row_cmp);
            qsort(b, nb, sizeof(gl_t) * w, row_cmp);
            if (na && memcmp(a, b, sizeof(gl_t) * w * na)) rc = 300 + (int)c;
        }
        free(a);
        free(b);
        if (rc) return rc;
    }
    return 0;
}

/* ---- prove_with_traces / verify_proof ---- */
size_t zko_all_proof_words(const zko_stark_config* cfg, const zko_table_input* tables, size_t ntables,
                           const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls, size_t* offs) {
    table_zs_t* tz = (table_zs_t*)calloc(ntables, sizeof(table_zs_t));
    derive_zs(ntables, ctls, sides, nctls, cfg->num_challenges, NULL, tz);
    size_t total = 0;
    for (size_t t = 0; t < ntables; t++) {
        if (offs) offs[t] = total;
        total += zko_proof_words(cfg, tables[t].log_n, tables[t].ncols, zko_num_lookup_columns(tables[t].table_id, cfg) + tz[t].naux, tz[t].nzs);
    }
    if (offs) offs[ntables] = total;
    free_zs(tz, ntables);
    free(tz);
    return total;
}

static void seed_transcript(zko_challenger* ch, const zko_stark_config* cfg, const uint64_t* caps /* ntables x C*4 */, size_t ntables,
                            const uint64_t* pub, size_t npub, uint64_t* challenges) {
    size_t capw = (size_t)4 << cfg->cap_height;
    zko_challenger_init(ch);
    for (size_t t = 0; t < ntables; t++) zko_challenger_observe(ch, caps + t * capw, capw);   /* prover.rs:182-185 */
    zko_challenger_observe(ch, pub, npub);                                                     /* :187 observe_public_values */
    for (unsigned c = 0; c < cfg->num_challenges; c++) {                                        /* :190 beta then gamma */
        challenges[2 * c] = zko_challenger_get(ch);
        challenges[2 * c + 1] = zko_challenger_get(ch);
    }
}

int zko_prove_with_traces(const zko_stark_config* cfg, const zko_table_input* tables, size_t ntables,
                          const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls,
                          const uint64_t* pub, size_t npub, uint64_t* proofs, uint64_t* challenges) {
    size_t capw = (size_t)4 << cfg->cap_height;
    size_t* offs = (size_t*)calloc(ntables + 1, sizeof(size_t));
    zko_all_proof_words(cfg, tables, ntables, ctls, sides, nctls, offs);
    uint64_t* caps = (uint64_t*)malloc(sizeof(uint64_t) * capw * ntables);
    for (size_t t = 0; t < ntables; t++) { /* trace commitments; prove_generic recommits (the oracle favours clarity) */
        zko_batch* b = zko_batch_from_values(tables[t].trace, tables[t].ncols, tables[t].log_n, cfg->rate_bits, cfg->cap_height);
        zko_batch_cap(b, caps + t * capw);
        zko_batch_free(b);
    }
    zko_challenger ch;
    seed_transcript(&ch, cfg, caps, ntables, pub, npub, challenges);
    table_zs_t* tz = (table_zs_t*)calloc(ntables, sizeof(table_zs_t));
    derive_zs(ntables, ctls, sides, nctls, cfg->num_challenges, challenges, tz);
    int rc = 0;
    uint64_t lookup_ch[4]; /* the betas of the CTL challenges (prover.rs:468-474) */
    for (unsigned c = 0; c < cfg->num_challenges; c++) lookup_ch[c] = challenges[2 * c];
    for (size_t t = 0; t < ntables && !rc; t++) {
        size_t n = (size_t)1 << tables[t].log_n;
        uint64_t* aux = (uint64_t*)calloc(tz[t].naux * n + 1, sizeof(uint64_t));
        double t0 = omp_get_wtime();
        zko_ctl_data(tables[t].ctl, tz[t].zs, tz[t].ids, tz[t].nzs, tables[t].trace, tables[t].ncols, tables[t].log_n, aux);
        double t1 = omp_get_wtime();
        if (getenv("ZKO_TIMING")) fprintf(stderr, "[oracle] table %zu id %d: ctl_data %.2f s\n", t, tables[t].table_id, t1 - t0);
        rc = zko_prove_single_table_ctl(tables[t].table_id, cfg, tables[t].trace, tables[t].ncols, tables[t].log_n, aux, tz[t].naux,
                                        tables[t].ctl, tz[t].zs, tz[t].ids, tz[t].nzs, lookup_ch, &ch, proofs + offs[t]);
        if (getenv("ZKO_TIMING")) fprintf(stderr, "[oracle] table %zu id %d: prove %.2f s\n", t, tables[t].table_id, omp_get_wtime() - t1);
        free(aux);
    }
    free_zs(tz, ntables);
    free(tz);
    free(caps);
    free(offs);
    return rc;
}

int zko_verify_all(const zko_stark_config* cfg, const zko_table_input* tables, size_t ntables,
                   const zko_cross_table_lookup* ctls, const zko_ctl_side* sides, size_t nctls,
                   const uint64_t* pub, size_t npub, const uint64_t* proofs, const uint64_t* challenges_claimed) {
    size_t capw = (size_t)4 << cfg->cap_height;
    size_t* offs = (size_t*)calloc(ntables + 1, sizeof(size_t));
    zko_all_proof_words(cfg, tables, ntables, ctls, sides, nctls, offs);
    uint64_t* caps = (uint64_t*)malloc(sizeof(uint64_t) * capw * ntables);
    for (size_t t = 0; t < ntables; t++) memcpy(caps + t * capw, proofs + offs[t] + 16 + 12, sizeof(uint64_t) * capw); /* trace_cap */
    zko_challenger ch;
    uint64_t challenges[8];
    seed_transcript(&ch, cfg, caps, ntables, pub, npub, challenges);   /* AllProof::get_challenges get_challenges.rs:124-148 */
    int rc = 0;
    if (challenges_claimed && memcmp(challenges, challenges_claimed, sizeof(uint64_t) * 2 * cfg->num_challenges)) rc = 40;
    table_zs_t* tz = (table_zs_t*)calloc(ntables, sizeof(table_zs_t));
    derive_zs(ntables, ctls, sides, nctls, cfg->num_challenges, challenges, tz);
    uint64_t lookup_ch[4];
    for (unsigned c = 0; c < cfg->num_challenges; c++) lookup_ch[c] = challenges[2 * c];
    for (size_t t = 0; t < ntables && !rc; t++) {
        rc = zko_verify_single_table_ctl(tables[t].table_id, cfg, proofs + offs[t], tables[t].ncols, tz[t].naux, tables[t].ctl,
                                         tz[t].zs, tz[t].ids, tz[t].nzs, lookup_ch, &ch);
        if (rc) rc += 1000 * (int)(t + 1);
    }
    /* verify_cross_table_lookups: per CTL and challenge, sum of the looking tables' Z(1) == looked Z(1) */
    if (!rc) {
        size_t* cursor = (size_t*)calloc(ntables, sizeof(size_t));
        for (size_t c = 0; c < nctls && !rc; c++) {
            const zko_ctl_side* lk = sides + ctls[c].looking_off;
            for (unsigned chn = 0; chn < cfg->num_challenges && !rc; chn++) {
                gl_t sum = 0;
                for (uint32_t i = 0; i < ctls[c].nlooking;) {
                    uint32_t j = i;
                    while (j < ctls[c].nlooking && lk[j].table == lk[i].table) j++;
                    size_t t = lk[i].table;
                    size_t W = tables[t].ncols, A = zko_num_lookup_columns(tables[t].table_id, cfg) + tz[t].naux;
                    const uint64_t* o_ctl = proofs + offs[t] + 16 + 12 + 3 * capw + 4 * W + 4 * A;
                    sum = gl_add(sum, o_ctl[cursor[t]++]);
                    i = j;
                }
                size_t t = ctls[c].looked.table;
                size_t W = tables[t].ncols, A = zko_num_lookup_columns(tables[t].table_id, cfg) + tz[t].naux;
                const uint64_t* o_ctl = proofs + offs[t] + 16 + 12 + 3 * capw + 4 * W + 4 * A;
                if (sum != o_ctl[cursor[t]++]) rc = 50 + (int)c;
            }
        }
        free(cursor);
    }
    free_zs(tz, ntables);
    free(tz);
    free(caps);
    free(offs);
    return rc;
}
