// segment.hip -- the data formats either side of the proving path, host code only.
//
// N4  zkm_proof_get_layout / zkm_proof_get_query_layout: field offsets of a proof blob, derived from the blob's own header, for
//     rebuilding StarkProofWithMetadata / StarkOpeningSet / FriProof (reference prover/src/proof.rs:178-334) on the caller's side.
// N3  zkm_segment_image_*: one flat image of a segment's traces + cross-table-lookup description (what generate_traces and
//     all_cross_table_lookups hand to prove_with_traces, prover.rs:130-142), and zkm_prove_segment_image proving from it.
#include <cstring>
#include <string>
#include <vector>

#include "zkm_internal.h"

namespace {
constexpr uint64_t SEGMENT_MAGIC = 0x45434152544d4b5aULL;  // "ZKMTRACE" little-endian
constexpr uint64_t SEGMENT_VERSION = 1;
constexpr size_t HEADER_WORDS = 8, TABLE_WORDS = 8;

int fail_msg(char** err, const std::string& m) {
    if (err) *err = strdup(m.c_str());
    return 1;
}
size_t u32_words(size_t n) { return (n + 1) / 2; }
size_t desc_words(const zkm_ctl_table* t) {
    if (!t) return 0;
    return 3 * t->ncolumns + u32_words(t->nterms) + t->nterms + 4 * t->ncolsets + u32_words(t->nfilter_idx);
}
}  // namespace

extern "C" {

int zkm_proof_get_layout(const uint64_t* p, zkm_proof_layout* y) {
    if (!p || !y || p[0] != ZKM_PROOF_MAGIC) return 1;
    memset(y, 0, sizeof *y);
    y->degree_bits = p[1]; y->trace_cols = p[2]; y->aux_cols = p[3]; y->quotient_polys = p[4]; y->ctl_zs = p[5]; y->cap_height = p[6];
    y->fri_layers = p[7]; y->final_poly_len = p[8]; y->num_queries = p[9]; y->rate_bits = p[10]; y->arity_bits = p[11];
    if (y->fri_layers > 16 || y->cap_height > 32 || y->degree_bits > 40) return 1;
    const size_t W = y->trace_cols, A = y->aux_cols, Q = y->quotient_polys, Z = y->ctl_zs, C = (size_t)1 << y->cap_height;
    size_t o = 16;
    y->init_challenger_state = o; o += 12;
    y->trace_cap = o; o += C * 4;
    y->aux_cap = o; o += C * 4;
    y->quotient_cap = o; o += C * 4;
    y->local_values = o; o += 2 * W;
    y->next_values = o; o += 2 * W;
    y->aux_polys = o; o += 2 * A;
    y->aux_polys_next = o; o += 2 * A;
    y->ctl_zs_first = o; o += Z;
    y->quotient_polys_open = o; o += 2 * Q;
    y->commit_phase_merkle_caps = o; o += y->fri_layers * C * 4;
    y->final_poly = o; o += 2 * y->final_poly_len;
    y->pow_witness = o; o += 1;
    y->query_round_proofs = o;
    zkm_proof_query_layout q;
    if (zkm_proof_get_query_layout(p, &q)) return 1;
    const unsigned L = (unsigned)y->fri_layers;
    y->query_round_words = L ? q.layer_siblings[L - 1] + 4 * q.layer_siblings_count[L - 1] : q.oracle_siblings[2] + 4 * q.initial_siblings;
    y->total_words = o + y->query_round_words * y->num_queries;
    return 0;
}

int zkm_proof_get_query_layout(const uint64_t* p, zkm_proof_query_layout* q) {
    if (!p || !q || p[0] != ZKM_PROOF_MAGIC || p[7] > 16) return 1;
    memset(q, 0, sizeof *q);
    const size_t cols[3] = {(size_t)p[2], (size_t)p[3], (size_t)p[4]};
    const size_t lde_bits = p[1] + p[10], cap = p[6], arity_bits = p[11], L = p[7];
    if (lde_bits < cap) return 1;
    q->initial_siblings = lde_bits - cap;
    size_t o = 0;
    for (int k = 0; k < 3; k++) {
        q->oracle_cols[k] = cols[k];
        q->oracle_evals[k] = o; o += cols[k];
        q->oracle_siblings[k] = o; o += 4 * q->initial_siblings;
    }
    for (size_t i = 0; i < L; i++) {
        if (lde_bits < arity_bits * (i + 1) + cap) return 1;
        q->layer_evals[i] = o; o += 2 * ((size_t)1 << arity_bits);
        q->layer_siblings_count[i] = lde_bits - arity_bits * (i + 1) - cap;
        q->layer_siblings[i] = o; o += 4 * q->layer_siblings_count[i];
    }
    return 0;
}

size_t zkm_segment_image_words(const zkm_table_input* tables, size_t ntables, const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides,
                               size_t nctls, size_t npublic) {
    (void)ctls;
    size_t nsides = 0;
    for (size_t i = 0; i < nctls; i++) nsides += ctls[i].nlooking;
    (void)sides;
    size_t w = HEADER_WORDS + npublic + TABLE_WORDS * ntables + 2 * nctls + nsides;
    for (size_t t = 0; t < ntables; t++) w += desc_words(tables[t].ctl) + (tables[t].ncols << tables[t].log_n);
    return w;
}

int zkm_segment_image_write(const zkm_table_input* tables, size_t ntables, const zkm_cross_table_lookup* ctls, const zkm_ctl_side* sides,
                            size_t nctls, const uint64_t* pub, size_t npub, uint64_t* img, char** err) {
    if (!img || (ntables && !tables) || (nctls && (!ctls || !sides))) return fail_msg(err, "zkm_segment_image_write: null argument");
    size_t nsides = 0;
    for (size_t i = 0; i < nctls; i++) {
        if (ctls[i].looking_off != nsides) return fail_msg(err, "zkm_segment_image_write: looking sides must be stored in lookup order");
        nsides += ctls[i].nlooking;
    }
    for (size_t t = 0; t < ntables; t++) {
        if (!tables[t].trace && !tables[t].columns) return fail_msg(err, "zkm_segment_image_write: table without a trace");
        if (!tables[t].columns && zkm_is_device_ptr(tables[t].trace)) return fail_msg(err, "zkm_segment_image_write: traces must be host pointers");
        for (size_t i = 0; tables[t].columns && i < tables[t].ncols; i++)
            if (!tables[t].columns[i] || zkm_is_device_ptr(tables[t].columns[i]))
                return fail_msg(err, "zkm_segment_image_write: columns must be host pointers");
    }
    img[0] = SEGMENT_MAGIC; img[1] = SEGMENT_VERSION; img[2] = ntables; img[3] = npub; img[4] = nctls; img[5] = nsides; img[6] = img[7] = 0;
    size_t o = HEADER_WORDS;
    if (npub) memcpy(img + o, pub, npub * 8);
    o += npub;
    uint64_t* th = img + o;
    o += TABLE_WORDS * ntables;
    for (size_t t = 0; t < ntables; t++) {
        const zkm_ctl_table* c = tables[t].ctl;
        uint64_t* h = th + TABLE_WORDS * t;
        h[0] = (uint64_t)tables[t].table_id; h[1] = tables[t].ncols; h[2] = tables[t].log_n; h[3] = 0;
        h[4] = c ? c->ncolumns : 0; h[5] = c ? c->nterms : 0; h[6] = c ? c->ncolsets : 0; h[7] = c ? c->nfilter_idx : 0;
        if (!c) continue;
        memcpy(img + o, c->columns, c->ncolumns * sizeof(zkm_column)); o += 3 * c->ncolumns;
        img[o + u32_words(c->nterms) - (c->nterms ? 1 : 0)] = 0;
        memcpy(img + o, c->term_col, c->nterms * 4); o += u32_words(c->nterms);
        memcpy(img + o, c->term_coeff, c->nterms * 8); o += c->nterms;
        memcpy(img + o, c->colsets, c->ncolsets * sizeof(zkm_colset)); o += 4 * c->ncolsets;
        if (c->nfilter_idx) img[o + u32_words(c->nfilter_idx) - 1] = 0;
        memcpy(img + o, c->filter_idx, c->nfilter_idx * 4); o += u32_words(c->nfilter_idx);
    }
    memcpy(img + o, ctls, nctls * sizeof(zkm_cross_table_lookup)); o += 2 * nctls;
    memcpy(img + o, sides, nsides * sizeof(zkm_ctl_side)); o += nsides;
    for (size_t t = 0; t < ntables; t++) {
        size_t words = tables[t].ncols << tables[t].log_n;
        th[TABLE_WORDS * t + 3] = o;
        if (tables[t].columns)
            for (size_t i = 0; i < tables[t].ncols; i++) memcpy(img + o + (i << tables[t].log_n), tables[t].columns[i], (size_t)8 << tables[t].log_n);
        else
            memcpy(img + o, tables[t].trace, words * 8);
        o += words;
    }
    return 0;
}

int zkm_prove_segment_image(zkm_ctx* c, const zkm_stark_config* cfg, const uint64_t* img, size_t image_words, uint64_t* proofs,
                            size_t* proof_words_out, size_t* offsets_out, uint64_t* challenges, char** err) {
    if (!img || image_words < HEADER_WORDS || img[0] != SEGMENT_MAGIC) return fail_msg(err, "zkm_prove_segment_image: not a ZKMTRACE image");
    if (img[1] != SEGMENT_VERSION) return fail_msg(err, "zkm_prove_segment_image: unsupported image version");
    // Every size below comes from the image: all checks are subtractions against the words that remain (no sums of untrusted
    // 64-bit fields, which could wrap), and every field is bounded before it is multiplied or used as a shift count.
    const uint64_t ntables = img[2], npub = img[3], nctls = img[4], nsides = img[5];
    size_t o = HEADER_WORDS;
    auto take = [&](uint64_t words, size_t per) -> bool {  // reserve words * per image words at o; false if they do not fit
        if (per && words > (image_words - o) / per) return false;
        o += (size_t)words * per;
        return true;
    };
    if (ntables > 4096 || nctls > 65536) return fail_msg(err, "zkm_prove_segment_image: truncated header");
    const uint64_t* pub = img + o;
    if (!take(npub, 1)) return fail_msg(err, "zkm_prove_segment_image: truncated header");
    const uint64_t* th = img + o;
    if (!take(ntables, TABLE_WORDS)) return fail_msg(err, "zkm_prove_segment_image: truncated header");
    std::vector<zkm_ctl_table> descs(ntables);
    std::vector<zkm_table_input> tables(ntables);
    for (size_t t = 0; t < ntables; t++) {
        const uint64_t* h = th + TABLE_WORDS * t;
        zkm_ctl_table& d = descs[t];
        const char* trunc = "zkm_prove_segment_image: truncated table description";
        // (u32 arrays are stored two per word: counts are bounded by twice the remaining words before u32_words() rounds them)
        if (h[1] == 0 || h[1] > (1u << 20) || h[2] > 40) return fail_msg(err, "zkm_prove_segment_image: table width / height out of range");
        if (h[5] > 2 * (uint64_t)(image_words - o) || h[7] > 2 * (uint64_t)(image_words - o)) return fail_msg(err, trunc);
        d.ncolumns = h[4]; d.nterms = h[5]; d.ncolsets = h[6]; d.nfilter_idx = h[7];
        d.columns = (const zkm_column*)(img + o);
        if (!take(h[4], 3)) return fail_msg(err, trunc);
        d.term_col = (const uint32_t*)(img + o);
        if (!take(u32_words(d.nterms), 1)) return fail_msg(err, trunc);
        d.term_coeff = img + o;
        if (!take(h[5], 1)) return fail_msg(err, trunc);
        d.colsets = (const zkm_colset*)(img + o);
        if (!take(h[6], 4)) return fail_msg(err, trunc);
        d.filter_idx = (const uint32_t*)(img + o);
        if (!take(u32_words(d.nfilter_idx), 1)) return fail_msg(err, trunc);
        tables[t].table_id = (int)h[0]; tables[t].ncols = h[1]; tables[t].log_n = (unsigned)h[2]; tables[t].ctl = &d;
        // ncols <= 2^20 and log_n <= 40: the product cannot wrap
        const uint64_t words = h[1] << h[2];
        if (h[3] > image_words || words > image_words - h[3]) return fail_msg(err, "zkm_prove_segment_image: trace data out of bounds");
        tables[t].trace = img + h[3];
        tables[t].columns = nullptr;
    }
    const zkm_cross_table_lookup* ctls = (const zkm_cross_table_lookup*)(img + o);
    if (!take(nctls, 2)) return fail_msg(err, "zkm_prove_segment_image: truncated lookup description");
    const zkm_ctl_side* sides = (const zkm_ctl_side*)(img + o);
    if (!take(nsides, 1)) return fail_msg(err, "zkm_prove_segment_image: truncated lookup description");
    for (size_t i = 0; i < nctls; i++)
        if (ctls[i].looking_off > nsides || ctls[i].nlooking > nsides - ctls[i].looking_off)
            return fail_msg(err, "zkm_prove_segment_image: looking sides out of range");
    std::vector<size_t> offs(ntables + 1, 0);
    size_t total = zkm_all_proof_words(cfg, tables.data(), ntables, ctls, sides, nctls, offs.data());
    if (!total && ntables) return fail_msg(err, "zkm_prove_segment_image: malformed cross-table lookups");
    if (proof_words_out) *proof_words_out = total;
    if (offsets_out) memcpy(offsets_out, offs.data(), (ntables + 1) * sizeof(size_t));
    if (!proofs) return 0;
    return zkm_prove_with_traces(c, cfg, tables.data(), ntables, ctls, sides, nctls, pub, npub, proofs, challenges, err);
}

}  // extern "C"
