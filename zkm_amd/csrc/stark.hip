// stark.hip -- prove_single_table on the GPU (quotient, openings, FRI).  (stage stubs: filled in next)
#include "zkm_internal.h"

static int fail_ni(char** err, const char* what) {
    std::string msg = std::string(what) + ": not implemented yet";
    if (err) { *err = (char*)malloc(msg.size() + 1); if (*err) memcpy(*err, msg.c_str(), msg.size() + 1); }
    return 2;
}
extern "C" {
size_t zkm_proof_words(const zkm_stark_config*, unsigned, size_t, size_t, size_t) { return 0; }
int zkm_prove_single_table(zkm_ctx*, int, const zkm_stark_config*, const uint64_t*, size_t, unsigned, const zkm_batch*,
                           const uint64_t*, size_t, const uint32_t*, size_t, zkm_challenger*, uint64_t*, char** err) {
    return fail_ni(err, "zkm_prove_single_table");
}
int zkm_quotient(zkm_ctx*, int, const zkm_batch*, const zkm_batch*, const uint32_t*, size_t, const uint64_t*, size_t, uint64_t*,
                 char** err) {
    return fail_ni(err, "zkm_quotient");
}
int zkm_eval_openings(zkm_ctx*, const zkm_batch*, const uint64_t*, uint64_t*, char** err) { return fail_ni(err, "zkm_eval_openings"); }
}
