/* tools/abi_layout.c -- prints sizeof / alignment / offsetof of every struct of include/zkm_hip.h as one JSON object.
 *
 * The C ABI has three mirrors that no compiler checks against each other: this header, the ctypes Structures / numpy dtypes in
 * zkm_amd/, and the #[repr(C)] structs in integration/rust/zkm_hip_sys.rs (uncompiled here: no Rust in the image).
 * tests/test_abi.py builds this file with gcc, runs it, and compares all three -- a reordered or re-typed field fails on the CPU
 * instead of corrupting memory on a maintainer's first run.  Built by oracle/Makefile (target abi_layout) and by the test itself.
 */
#include <stddef.h>
#include <stdio.h>

#include "../include/zkm_hip.h"

#define ALIGN_OF(T) offsetof(struct { char c; T t; }, t)
#define FIELD(T, f) printf("%s[\"%s\", %zu, %zu]", first_field ? "" : ", ", #f, offsetof(T, f), sizeof(((T*)0)->f)), first_field = 0
#define BEGIN(T) printf("%s\n  \"%s\": {\"size\": %zu, \"align\": %zu, \"fields\": [", first_struct ? "" : ",", #T, sizeof(T), ALIGN_OF(T)), \
                 first_struct = 0, first_field = 1
#define END() printf("]}")

int main(void) {
    int first_struct = 1, first_field = 1;
    printf("{");
    BEGIN(zkm_challenger);
    FIELD(zkm_challenger, state); FIELD(zkm_challenger, in_buf); FIELD(zkm_challenger, out_buf); FIELD(zkm_challenger, n_in);
    FIELD(zkm_challenger, n_out);
    END();
    BEGIN(zkm_stark_config);
    FIELD(zkm_stark_config, rate_bits); FIELD(zkm_stark_config, cap_height); FIELD(zkm_stark_config, pow_bits);
    FIELD(zkm_stark_config, num_challenges); FIELD(zkm_stark_config, num_queries); FIELD(zkm_stark_config, arity_bits);
    FIELD(zkm_stark_config, final_poly_bits);
    END();
    BEGIN(zkm_proof_layout);
    FIELD(zkm_proof_layout, degree_bits); FIELD(zkm_proof_layout, trace_cols); FIELD(zkm_proof_layout, aux_cols);
    FIELD(zkm_proof_layout, quotient_polys); FIELD(zkm_proof_layout, ctl_zs); FIELD(zkm_proof_layout, cap_height);
    FIELD(zkm_proof_layout, fri_layers); FIELD(zkm_proof_layout, final_poly_len); FIELD(zkm_proof_layout, num_queries);
    FIELD(zkm_proof_layout, rate_bits); FIELD(zkm_proof_layout, arity_bits); FIELD(zkm_proof_layout, total_words);
    FIELD(zkm_proof_layout, init_challenger_state); FIELD(zkm_proof_layout, trace_cap); FIELD(zkm_proof_layout, aux_cap);
    FIELD(zkm_proof_layout, quotient_cap); FIELD(zkm_proof_layout, local_values); FIELD(zkm_proof_layout, next_values);
    FIELD(zkm_proof_layout, aux_polys); FIELD(zkm_proof_layout, aux_polys_next); FIELD(zkm_proof_layout, ctl_zs_first);
    FIELD(zkm_proof_layout, quotient_polys_open); FIELD(zkm_proof_layout, commit_phase_merkle_caps); FIELD(zkm_proof_layout, final_poly);
    FIELD(zkm_proof_layout, pow_witness); FIELD(zkm_proof_layout, query_round_proofs); FIELD(zkm_proof_layout, query_round_words);
    END();
    BEGIN(zkm_proof_query_layout);
    FIELD(zkm_proof_query_layout, oracle_evals); FIELD(zkm_proof_query_layout, oracle_cols); FIELD(zkm_proof_query_layout, oracle_siblings);
    FIELD(zkm_proof_query_layout, initial_siblings); FIELD(zkm_proof_query_layout, layer_evals); FIELD(zkm_proof_query_layout, layer_siblings);
    FIELD(zkm_proof_query_layout, layer_siblings_count);
    END();
    BEGIN(zkm_column);
    FIELD(zkm_column, n_local); FIELD(zkm_column, n_next); FIELD(zkm_column, term_off); FIELD(zkm_column, _pad); FIELD(zkm_column, constant);
    END();
    BEGIN(zkm_colset);
    FIELD(zkm_colset, ncols); FIELD(zkm_colset, col_off); FIELD(zkm_colset, has_filter); FIELD(zkm_colset, nprod); FIELD(zkm_colset, prod_off);
    FIELD(zkm_colset, nconst); FIELD(zkm_colset, const_off); FIELD(zkm_colset, _pad);
    END();
    BEGIN(zkm_ctl_table);
    FIELD(zkm_ctl_table, columns); FIELD(zkm_ctl_table, ncolumns); FIELD(zkm_ctl_table, term_col); FIELD(zkm_ctl_table, term_coeff);
    FIELD(zkm_ctl_table, nterms); FIELD(zkm_ctl_table, colsets); FIELD(zkm_ctl_table, ncolsets); FIELD(zkm_ctl_table, filter_idx);
    FIELD(zkm_ctl_table, nfilter_idx);
    END();
    BEGIN(zkm_ctl_z);
    FIELD(zkm_ctl_z, ncolsets); FIELD(zkm_ctl_z, colset_off); FIELD(zkm_ctl_z, num_helpers); FIELD(zkm_ctl_z, _pad); FIELD(zkm_ctl_z, beta);
    FIELD(zkm_ctl_z, gamma);
    END();
    BEGIN(zkm_ctl_side);
    FIELD(zkm_ctl_side, table); FIELD(zkm_ctl_side, colset);
    END();
    BEGIN(zkm_cross_table_lookup);
    FIELD(zkm_cross_table_lookup, nlooking); FIELD(zkm_cross_table_lookup, looking_off); FIELD(zkm_cross_table_lookup, looked);
    END();
    BEGIN(zkm_table_input);
    FIELD(zkm_table_input, table_id); FIELD(zkm_table_input, trace); FIELD(zkm_table_input, ncols); FIELD(zkm_table_input, log_n);
    FIELD(zkm_table_input, ctl); FIELD(zkm_table_input, columns);
    END();
    BEGIN(zkm_fri_poly);
    FIELD(zkm_fri_poly, oracle); FIELD(zkm_fri_poly, poly);
    END();
    BEGIN(zkm_fri_batch);
    FIELD(zkm_fri_batch, point); FIELD(zkm_fri_batch, polys); FIELD(zkm_fri_batch, npolys);
    END();
    printf("\n}\n");
    return 0;
}
