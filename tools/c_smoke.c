/* tools/c_smoke.c -- the C ABI used from plain C99, no Python anywhere: read a ZKMTRACE segment image from a file, prove it with
 * zkm_prove_segment_image (and, for a twelve-table image, twice more in one zkm_prove_segments call and three times through a zkm_pool of
 * two workers), write the proof blobs to a file.  The reference-side caller of this boundary is Rust over FFI
 * (its only existing FFI has this very shape: recursion/src/snark/snarks.rs:7-20, 39-59 -- int status, char** message
 * freed by the caller); this file is the proof that include/zkm_hip.h is usable as it stands by a C compiler in pedantic mode.
 *
 *   gcc -std=c99 -pedantic -Wall -Werror -Iinclude tools/c_smoke.c -Lzkm_amd/csrc -lzkmhip -Wl,-rpath,$PWD/zkm_amd/csrc \
 *       -Wl,-rpath-link,/opt/rocm/lib -o c_smoke
 *   ./c_smoke segment.zkmtrace proofs.bin        exit code 0 and "ok <words> words, <ntables> tables" on success
 *
 * tests/test_abi.py builds it (CPU suite: compiles, links, reports the missing GPU through the error channel) and
 * tests/test_segment.py runs it on the GPU box and compares the blob with the Python path's and the oracle's.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkm_hip.h"

static int die(const char* what, char* err) {
    fprintf(stderr, "c_smoke: %s: %s\n", what, err ? err : "(no message)");
    free(err); /* the library's messages are malloc'd: the caller frees */
    return 1;
}

int main(int argc, char** argv) {
    FILE* f;
    long bytes;
    uint64_t *image, *proofs, challenges[8];
    size_t words, proof_words = 0, offsets[65], ntables, k;
    zkm_ctx* ctx = NULL;
    zkm_stark_config cfg;
    char* err = NULL;

    if (argc != 3) {
        fprintf(stderr, "usage: %s <segment image> <proofs out>\n", argv[0]);
        return 2;
    }
    f = fopen(argv[1], "rb");
    if (!f) return die("cannot open the image", NULL);
    fseek(f, 0, SEEK_END);
    bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (bytes < 64 || bytes % 8) return die("not a segment image (size)", NULL);
    words = (size_t)bytes / 8;
    image = (uint64_t*)malloc((size_t)bytes);
    if (!image || fread(image, 8, words, f) != words) return die("short read", NULL);
    fclose(f);
    if (memcmp(image, "ZKMTRACE", 8) != 0) return die("not a segment image (magic)", NULL);
    ntables = (size_t)image[2];
    if (ntables > 64) return die("too many tables for this tool", NULL);

    printf("%s\n", zkm_version());
    zkm_standard_config(&cfg);
    if (zkm_ctx_create(0, &ctx, &err)) return die("zkm_ctx_create", err);
    /* first call: sizes only */
    if (zkm_prove_segment_image(ctx, &cfg, image, words, NULL, &proof_words, offsets, NULL, &err)) return die("sizing", err);
    proofs = (uint64_t*)malloc(proof_words * 8);
    if (!proofs) return die("out of host memory", NULL);
    if (zkm_prove_segment_image(ctx, &cfg, image, words, proofs, &proof_words, offsets, challenges, &err)) return die("proving", err);
    if (ntables == 12) {
        /* the same segment twice through zkm_prove_segments (K = 2 in lock-step: one launch per stage for both): each blob must equal the
         * single-segment proof word for word.  The traces are where the image holds them (table header words 2 and 3: log_n, offset). */
        const uint64_t* th = image + 8 + image[3];
        const uint64_t* traces[12];
        const uint64_t* const* seg_traces[2];
        unsigned log_n[12];
        const unsigned* seg_log_n[2];
        const uint64_t* pubs[2];
        size_t npubs[2];
        uint64_t *both[2], *chal[2], chal_words[2][8];
        for (k = 0; k < 12; k++) {
            log_n[k] = (unsigned)th[8 * k + 2];
            traces[k] = image + th[8 * k + 3];
        }
        both[0] = (uint64_t*)malloc(proof_words * 8);
        both[1] = (uint64_t*)malloc(proof_words * 8);
        if (!both[0] || !both[1]) return die("out of host memory", NULL);
        for (k = 0; k < 2; k++) {
            seg_traces[k] = traces; seg_log_n[k] = log_n; pubs[k] = image + 8; npubs[k] = (size_t)image[3]; chal[k] = chal_words[k];
        }
        if (zkm_prove_segments(ctx, &cfg, 2, seg_traces, seg_log_n, pubs, npubs, both, chal, &err)) return die("zkm_prove_segments", err);
        for (k = 0; k < 2; k++)
            if (memcmp(both[k], proofs, proof_words * 8) != 0 || memcmp(chal[k], challenges, 4 * 8) != 0)
                return die("a lock-step proof differs from the single-segment proof", NULL);
        printf("lockstep ok: 2 segments, each blob == the single-segment proof\n");
        free(both[0]);
        free(both[1]);
        {
            /* the one-process multi-device shape (zkm_pool_*): three copies of the segment through a pool of two workers on device 0,
             * groups of at most 2 -> a group of 2 and a group of 1 served side by side; every blob == the single-segment proof */
            const int devices[1] = {0};
            zkm_pool* pool = NULL;
            const uint64_t* const* p_traces[3];
            const unsigned* p_log_n[3];
            const uint64_t* p_pubs[3];
            size_t p_npubs[3], plan[3], ngroups, w0 = 9, g0 = 9, w2 = 9, g2 = 9;
            uint64_t *p_proofs[3], *p_chal[3], p_chal_words[3][8];
            for (k = 0; k < 3; k++) {
                p_traces[k] = traces; p_log_n[k] = log_n; p_pubs[k] = image + 8; p_npubs[k] = (size_t)image[3]; p_chal[k] = p_chal_words[k];
                p_proofs[k] = (uint64_t*)malloc(proof_words * 8);
                if (!p_proofs[k]) return die("out of host memory", NULL);
            }
            ngroups = zkm_pool_plan(3, 2, 2, plan, 3);
            if (ngroups != 2 || plan[0] != 2 || plan[1] != 1) return die("zkm_pool_plan: 3 segments, 2 workers, groups of <= 2 should be 2 + 1", NULL);
            if (zkm_pool_create(devices, 1, 2, &pool, &err)) return die("zkm_pool_create", err);
            if (zkm_pool_workers(pool) != 2 || zkm_pool_device(pool, 1) != 0 || !zkm_pool_context(pool, 1)) return die("pool shape", NULL);
            if (zkm_pool_prove_segments(pool, &cfg, 3, 2, p_traces, p_log_n, p_pubs, p_npubs, p_proofs, p_chal, &err))
                return die("zkm_pool_prove_segments", err);
            for (k = 0; k < 3; k++)
                if (memcmp(p_proofs[k], proofs, proof_words * 8) != 0 || memcmp(p_chal[k], challenges, 4 * 8) != 0)
                    return die("a pool proof differs from the single-segment proof", NULL);
            if (zkm_pool_last_assignment(pool, 0, &w0, &g0) || zkm_pool_last_assignment(pool, 2, &w2, &g2) || g0 != 0 || g2 != 1 || w0 == w2)
                return die("pool assignment: two groups on two workers expected", NULL);
            printf("pool ok: 3 segments, 2 workers on device 0, groups 2 + 1, each blob == the single-segment proof\n");
            zkm_pool_destroy(pool);
            for (k = 0; k < 3; k++) free(p_proofs[k]);
        }
    }
    zkm_ctx_destroy(ctx);

    f = fopen(argv[2], "wb");
    if (!f || fwrite(proofs, 8, proof_words, f) != proof_words) return die("cannot write the proofs", NULL);
    fclose(f);
    for (k = 0; k < ntables; k++) {
        zkm_proof_layout lay;
        if (zkm_proof_get_layout(proofs + offsets[k], &lay) || lay.total_words != offsets[k + 1] - offsets[k])
            return die("a proof blob does not describe itself", NULL);
    }
    printf("ok %lu words, %lu tables, beta0 %016llx\n", (unsigned long)proof_words, (unsigned long)ntables, (unsigned long long)challenges[0]);
    free(proofs);
    free(image);
    return 0;
}
