// pool.hip -- one process, many GPUs: a pool of contexts that proves the independent segments of a program side by side.
//
// The reference drives every segment of a program from ONE process: prove_single_seg_common / prove_multi_seg_common
// (prover/examples/utils/src/utils.rs:57-68, 105-133) loop over the segment files and call prove_with_traces
// (prover/src/prover.rs:130-232) for each.  Segments are independent proofs (no data-path collective), so the MI355X
// shape of that loop is a queue of lock-step groups served by one worker thread per context, `contexts_per_device` contexts on
// each device: every entry point of the library selects its context's device itself (hipSetDevice(c->device), core.hip), so the
// workers need nothing but their own context.  bench.py's process-per-GPU harness (zkm_amd/dist.py) stays what the scaling bench
// launches; this is what a Rust caller binds (integration/rust/prove_hip.rs prove_segments_multi_hip).
//
// Host only: no kernel lives here.
#include <exception>
#include <mutex>
#include <thread>

#include "zkm_internal.h"

struct zkm_pool {
    std::vector<zkm_ctx*> ctxs;        // worker w's context
    std::vector<int> devices;          // ... and its device
    std::vector<size_t> last_worker, last_group;   // of the last call: who proved segment s, in which group
};

static int pool_fail(char** err, const std::string& msg) {
    if (err) {
        *err = (char*)malloc(msg.size() + 1);
        if (*err) memcpy(*err, msg.c_str(), msg.size() + 1);
    }
    return 1;
}

// The groups of one call: consecutive runs of at most `stack` segments, the same number of groups for every worker (the fewest that
// keeps a group within `stack`), sizes as even as the count allows -- 20 segments, 2 workers, stack 4 -> 4, 4, 3, 3, 3, 3
// (zkm_amd/dist.py chunk_segments is the same rule; tests/test_pool.py holds them against each other).
static std::vector<std::pair<size_t, size_t>> pool_groups(size_t nseg, size_t workers, size_t stack) {
    std::vector<std::pair<size_t, size_t>> g;   // (first segment, count)
    if (!nseg) return g;
    workers = std::max<size_t>(1, workers);
    stack = std::max<size_t>(1, stack);
    const size_t per_worker = (nseg + workers * stack - 1) / (workers * stack);
    const size_t ngroups = std::min(nseg, workers * per_worker);
    const size_t base = nseg / ngroups, extra = nseg % ngroups;
    for (size_t k = 0, s = 0; k < ngroups; k++) {
        const size_t size = base + (k < extra ? 1 : 0);
        g.emplace_back(s, size);
        s += size;
    }
    return g;
}

static int pool_prove(const char* what, zkm_pool* p, const zkm_stark_config* cfg, size_t nseg, size_t max_stack,
                      const uint64_t* const* const* traces, const uint64_t* const* const* const* columns, const unsigned* const* log_n,
                      const uint64_t* const* pub, const size_t* npub, uint64_t* const* proofs, uint64_t* const* challenges, char** err) {
    if (!p || !cfg || (!traces && !columns) || !log_n || !proofs || !challenges) return pool_fail(err, std::string(what) + ": null argument");
    if (max_stack == 0) max_stack = 8;
    if (max_stack > ZKM_MAX_SEG) return pool_fail(err, std::string(what) + ": max_stack beyond " + std::to_string(ZKM_MAX_SEG));
    const size_t W = p->ctxs.size();
    const auto groups = pool_groups(nseg, W, max_stack);
    p->last_worker.assign(nseg, ~(size_t)0);
    p->last_group.assign(nseg, ~(size_t)0);
    if (groups.empty()) return 0;
    std::mutex mu;
    size_t next = 0;
    std::string first_error;
    bool failed = false;
    auto worker = [&](size_t w) {
        for (;;) {
            size_t g;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (failed || next >= groups.size()) return;
                g = next++;
            }
            const size_t s0 = groups[g].first, k = groups[g].second;
            char* e = nullptr;
            int rc = 1;
            try {
                rc = zkm_prove_segments_entry(what, p->ctxs[w], cfg, k, traces ? traces + s0 : nullptr, columns ? columns + s0 : nullptr, log_n + s0,
                                              pub ? pub + s0 : nullptr, npub ? npub + s0 : nullptr, proofs + s0, challenges + s0, &e, s0);
            } catch (const std::exception& x) {    // (the entry catches everything itself: belt and braces, a worker thread must not unwind)
                rc = 1;
                const std::string m = x.what();
                e = (char*)malloc(m.size() + 1);
                if (e) memcpy(e, m.c_str(), m.size() + 1);
            } catch (...) {
                rc = 1;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (rc != 0) {
                if (!failed)
                    first_error = "worker " + std::to_string(w) + " (device " + std::to_string(p->devices[w]) + "), segments " + std::to_string(s0) + ".." +
                                  std::to_string(s0 + k - 1) + ": " + (e ? e : "unknown error");
                failed = true;
            } else {
                for (size_t s = s0; s < s0 + k; s++) { p->last_worker[s] = w; p->last_group[s] = g; }
            }
            free(e);
        }
    };
    // one thread per context that has work (a thread start is ~50 us; a group is milliseconds to seconds).  The calling thread is worker 0.
    const size_t nthreads = std::min(W, groups.size());
    std::vector<std::thread> th;
    try {
        for (size_t w = 1; w < nthreads; w++) th.emplace_back(worker, w);
    } catch (...) {
        // (a thread could not be started: the ones that run, and the calling thread below, drain the queue)
    }
    worker(0);
    for (auto& t : th) t.join();
    if (failed) return pool_fail(err, first_error);
    return 0;
}

extern "C" {

int zkm_pool_create(const int* devices, size_t ndevices, size_t contexts_per_device, zkm_pool** out, char** err) {
    if (!out) return pool_fail(err, "zkm_pool_create: null argument");
    *out = nullptr;
    if (!devices || ndevices == 0 || contexts_per_device == 0) return pool_fail(err, "zkm_pool_create: at least one device and one context per device");
    if (ndevices > 64 || contexts_per_device > 64) return pool_fail(err, "zkm_pool_create: at most 64 devices x 64 contexts");
    for (size_t i = 0; i < ndevices; i++)
        for (size_t j = 0; j < i; j++)
            if (devices[i] == devices[j]) return pool_fail(err, "zkm_pool_create: device " + std::to_string(devices[i]) + " listed twice");
    zkm_pool* p = nullptr;
    try {
        p = new zkm_pool();
    } catch (...) {
        return pool_fail(err, "zkm_pool_create: out of memory");
    }
    // worker order: device-major round robin (worker w -> device w % ndevices), so the first `ndevices` groups of a call land on
    // different devices even when the call has fewer groups than workers
    for (size_t k = 0; k < contexts_per_device; k++)
        for (size_t d = 0; d < ndevices; d++) {
            zkm_ctx* c = nullptr;
            char* e = nullptr;
            if (zkm_ctx_create(devices[d], &c, &e) != 0) {
                const std::string msg = "zkm_pool_create: context " + std::to_string(k) + " on device " + std::to_string(devices[d]) + ": " + (e ? e : "failed");
                free(e);
                zkm_pool_destroy(p);
                return pool_fail(err, msg);
            }
            p->ctxs.push_back(c);
            p->devices.push_back(devices[d]);
        }
    *out = p;
    return 0;
}

void zkm_pool_destroy(zkm_pool* p) {
    if (!p) return;
    for (zkm_ctx* c : p->ctxs) zkm_ctx_destroy(c);
    delete p;
}

size_t zkm_pool_workers(const zkm_pool* p) { return p ? p->ctxs.size() : 0; }
zkm_ctx* zkm_pool_context(zkm_pool* p, size_t w) { return (p && w < p->ctxs.size()) ? p->ctxs[w] : nullptr; }
int zkm_pool_device(const zkm_pool* p, size_t w) { return (p && w < p->devices.size()) ? p->devices[w] : -1; }

int zkm_pool_set_tuning(zkm_pool* p, const char* key, uint64_t value, char** err) {
    if (!p) return pool_fail(err, "zkm_pool_set_tuning: null argument");
    for (zkm_ctx* c : p->ctxs) {
        const int rc = zkm_ctx_set_tuning(c, key, value, err);
        if (rc != 0) return rc;
    }
    return 0;
}

int zkm_pool_prove_segments(zkm_pool* p, const zkm_stark_config* cfg, size_t nseg, size_t max_stack, const uint64_t* const* const* traces,
                            const unsigned* const* log_n, const uint64_t* const* pub, const size_t* npub, uint64_t* const* proofs,
                            uint64_t* const* challenges, char** err) {
    return pool_prove("zkm_pool_prove_segments", p, cfg, nseg, max_stack, traces, nullptr, log_n, pub, npub, proofs, challenges, err);
}

int zkm_pool_prove_segments_columns(zkm_pool* p, const zkm_stark_config* cfg, size_t nseg, size_t max_stack,
                                    const uint64_t* const* const* const* columns, const unsigned* const* log_n, const uint64_t* const* pub,
                                    const size_t* npub, uint64_t* const* proofs, uint64_t* const* challenges, char** err) {
    return pool_prove("zkm_pool_prove_segments_columns", p, cfg, nseg, max_stack, nullptr, columns, log_n, pub, npub, proofs, challenges, err);
}

size_t zkm_pool_plan(size_t nseg, size_t workers, size_t max_stack, size_t* group_sizes_out, size_t capacity) {
    if (max_stack == 0) max_stack = 8;
    const auto g = pool_groups(nseg, workers, max_stack);
    for (size_t k = 0; k < g.size() && k < capacity && group_sizes_out; k++) group_sizes_out[k] = g[k].second;
    return g.size();
}

int zkm_pool_last_assignment(const zkm_pool* p, size_t segment, size_t* worker_out, size_t* group_out) {
    if (!p || segment >= p->last_worker.size() || p->last_worker[segment] == ~(size_t)0) return 1;
    if (worker_out) *worker_out = p->last_worker[segment];
    if (group_out) *group_out = p->last_group[segment];
    return 0;
}

}  // extern "C"
