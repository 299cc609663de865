// core.hip -- context, memory, profiling, Fiat-Shamir transcript and the PolynomialBatch part of the C ABI.
#include <sched.h>
#include <time.h>
#include <stdio.h>
#include <algorithm>
#include <atomic>

#include "poseidon_dev.h"
#include "zkm_internal.h"

// ------------------------------------------------------------------ error plumbing
static int fail(char** err, const std::string& msg) {
    if (err) {
        *err = (char*)malloc(msg.size() + 1);
        if (*err) memcpy(*err, msg.c_str(), msg.size() + 1);
    }
    return 1;
}
#define ZKM_API_BEGIN try {
#define ZKM_API_END(err)                 \
    }                                    \
    catch (const std::exception& e) {    \
        return fail(err, e.what());      \
    }                                    \
    catch (...) {                        \
        return fail(err, "unknown error"); \
    }                                    \
    return 0;

// ------------------------------------------------------------------ ctx
// The allocator of a context is used by ONE thread at a time (its owner, or the lane thread of run_on_lanes) -- except for the
// out-of-memory path, which reaches into the caches of the parent and sibling contexts: every allocator therefore has a mutex, held
// for the map operations only (never across hipMalloc of another context's lock: one lock at a time, no ordering to get wrong).
void* zkm_ctx::alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    void* p = nullptr;
    {
        std::lock_guard<std::mutex> g(alloc_mu);
        auto it = free_blocks.find(bytes);
        if (it != free_blocks.end()) {
            p = it->second;
            free_blocks.erase(it);
            live_blocks[p] = bytes;
            return p;
        }
    }
    zkm_ctx* const root = parent ? parent : this;
    // (test hook, zkm_ctx_set_tuning "debug_fail_allocs" = k: the next k first attempts of the family fail as if out of memory, so the
    // retry path -- trimming caches while the lanes work -- runs where a test can watch it)
    const int ticket = root->debug_fail_allocs.load(std::memory_order_relaxed) > 0 ? root->debug_fail_allocs.fetch_sub(1) : 0;
    const bool inject = ticket > 0;
    hipError_t e = inject ? hipErrorOutOfMemory : hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // Drop the caches and retry once (the failed call's error must not stay behind as the "last error" of the next launch check):
        // this context's own cache first, then -- gigabytes may sit cached next door while this lane starves -- the caches of the
        // family (parent and sibling lanes).  Only CACHED blocks are freed, under their owner's lock; nothing another thread is using.
        (void)hipGetLastError();
        trim_self();
        e = (inject && (ticket & 1)) ? hipErrorOutOfMemory : hipMalloc(&p, bytes);   // (the hook sends some requests on to the family path)
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // (root->lanes only grows in ensure_lanes, on the owner's thread BEFORE it starts the lane threads of a call -- never while they
            // run: iterating it here, from a lane's thread, races with nothing)
            if (root != this) root->trim_self();
            for (zkm_ctx* l : root->lanes)
                if (l != this) l->trim_self();
            ZKM_HIP_CHECK(hipMalloc(&p, bytes));
        }
    }
    std::lock_guard<std::mutex> g(alloc_mu);
    live_blocks[p] = bytes;
    return p;
}
// Cached blocks may still be read or written by kernels queued on the owner's streams (a block is released as soon as the host is
// done with it, not the GPU): the streams are drained before anything is freed.  hipStreamSynchronize from a foreign thread is safe.
void zkm_ctx::trim_self() {
    std::multimap<size_t, void*> drop;
    {
        std::lock_guard<std::mutex> g(alloc_mu);
        drop.swap(free_blocks);                                // from here on nobody can be handed these blocks again ...
    }
    (void)hipStreamSynchronize(stream);                        // ... and what was queued on them before has completed after this
    hipStream_t cs, cs2;
    {
        std::lock_guard<std::mutex> g(alloc_mu);               // (the owner publishes its copy streams under the same lock: zkm_batch_build, zkm_trace_stage)
        cs = copy_stream;
        cs2 = copy_stream2;
    }
    if (cs) (void)hipStreamSynchronize(cs);                    // (or was the target of an upload in flight)
    if (cs2) (void)hipStreamSynchronize(cs2);
    for (auto& kv : drop) (void)hipFree(kv.second);
}
// The pinned download area grows with the largest lock-step group seen (up to XFER_DOWN_MAX per context and per lane) and nothing
// else gave it back: a trim between calls returns it to its base size.  Only from the owner's thread, between calls
// (downloads are waited for before their caller returns, the stream is drained by trim_self) -- not from a relative's
// out-of-memory retry, which trims device blocks only.
void zkm_ctx::shrink_down() {
    if (!h_down || down_cap <= XFER_DOWN) return;
    (void)hipStreamSynchronize(stream);
    (void)hipHostFree(h_down);
    h_down = nullptr;
    down_cap = 0;
    ensure_down(XFER_DOWN);
}
// The upload streams go as well (created again on first use): an idle stream still holds one of the runtime's hardware queues, and a GPU
// whose queues are oversubscribed by idle contexts runs everybody's launches slower -- bench.py's small-segment extra, a fresh process,
// fell from 105 to 95 segments/s next to four parked contexts with two copy streams each (round 6).
void zkm_ctx::drop_copy_streams() {
    hipStream_t a, b;
    {
        std::lock_guard<std::mutex> g(alloc_mu);
        a = copy_stream; b = copy_stream2;
        copy_stream = copy_stream2 = nullptr;
    }
    for (hipStream_t st : {a, b})
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
}
void zkm_ctx::trim() {   // public: between calls (zkm_ctx_trim)
    trim_self();
    shrink_down();
    drop_copy_streams();
    for (zkm_ctx* l : lanes) {
        l->trim_self();
        l->shrink_down();
        l->drop_copy_streams();
    }
}
// A stream of the library.  ZKM_CU_MASK_PART = "k/n" (measurement aid, read when a context is created; its lanes inherit it): the
// context's streams may only use the k-th of n equal parts of the GPU's compute units (mask bits [k N/n, (k+1) N/n)) --
// tools/contention_test.py and tools/bench_segment.py use it to find out what launches of different contexts cost each other.
static hipError_t zkm_stream_create(hipStream_t* st, int num_cus, int part_k, int part_n) {
    if (part_n > 1 && part_k >= 0 && part_k < part_n && num_cus >= 8 * part_n) {
        std::vector<uint32_t> mask((size_t)(num_cus + 31) / 32, 0);
        const int lo = (int)((long)num_cus * part_k / part_n), hi = (int)((long)num_cus * (part_k + 1) / part_n);
        for (int i = lo; i < hi; i++) mask[i / 32] |= 1u << (i % 32);
        return hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data());
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
void zkm_ctx::ensure_lanes(size_t k) {
    while (lanes.size() < k) {
        zkm_ctx* l = new zkm_ctx();
        l->parent = this;
        l->device = device;
        l->num_cus = num_cus;
        l->ingest_chunk_cols = ingest_chunk_cols;
        l->keccak_parts_max_points = keccak_parts_max_points;
        l->fri_fused_division_min = fri_fused_division_min;
        l->wide_max_hashes = wide_max_hashes;
        l->leaf_mfma = leaf_mfma;
        l->quad_max_hashes = quad_max_hashes;
        l->small_ntt = small_ntt;
        l->tree_tail = tree_tail;
        l->commit_lanes = commit_lanes;
        l->max_stack = max_stack;
        l->fri_scan_combine = fri_scan_combine;
        l->pow_round_log = pow_round_log;
        l->cu_part_k = cu_part_k;
        l->cu_part_n = cu_part_n;
        hipError_t e = zkm_stream_create(&l->stream, num_cus, cu_part_k, cu_part_n);
        if (e != hipSuccess) {
            delete l;
            ZKM_HIP_CHECK(e);
        }
        lanes.push_back(l);
    }
    for (zkm_ctx* l : lanes) l->profiling = profiling;
}
void zkm_ctx::release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(alloc_mu);
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) return;
    free_blocks.emplace(it->second, p);
    live_blocks.erase(it);
}
uint64_t* zkm_ctx::staging(size_t words) {
    if (words > h_staging_words) {
        if (h_staging) (void)hipHostFree(h_staging);
        size_t w = words < 4096 ? 4096 : words;
        ZKM_HIP_CHECK(hipHostMalloc((void**)&h_staging, w * sizeof(uint64_t)));
        h_staging_words = w;
    }
    return h_staging;
}
// Small downloads (a cap, opening partials, a proof-of-work witness: up to XFER_KERNEL bytes) by a one-workgroup kernel that writes the
// words into pinned host memory and then a sequence number into a flag the host spins on: the transcript's round trips cost the GPU's
// write latency over PCIe instead of a blit + completion signal + the runtime waking the waiting thread (~40 us per round trip, ~120
// round trips in a twelve-table segment).
struct down_args {
    const uint32_t* src[4];
    uint32_t* dst[4];
    uint32_t words[4];   // 32-bit words
    uint32_t n;
};
__global__ __launch_bounds__(1024) void k_download(down_args a, uint64_t* flag, uint64_t seq) {
    for (uint32_t s = 0; s < a.n; s++)
        for (uint32_t i = threadIdx.x; i < a.words[s]; i += 1024) a.dst[s][i] = a.src[s][i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Larger downloads that still fit the pinned area (a table's query rounds: 0.1 .. 1 MB): many workgroups write the words into pinned host
// memory, the last one to finish (a ticket) publishes the sequence number -- no blit, no completion signal, no stream synchronisation
__global__ __launch_bounds__(256) void k_download_wide(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16, uint64_t* flag,
                                                       uint64_t seq, unsigned* counter) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = atomicAdd(counter, 1u);
        if (ticket == gridDim.x - 1) {
            *counter = 0;
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
static char* pinned_alloc(size_t bytes) {
    char* p = nullptr;
    if (hipHostMalloc((void**)&p, bytes, hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();   // (a runtime that refuses the explicit flag: the default pinned allocation is coherent on this platform too)
        p = nullptr;
        ZKM_HIP_CHECK(hipHostMalloc((void**)&p, bytes, hipHostMallocDefault));
    }
    return p;
}
void zkm_ctx::ensure_xfer() {
    if (h_xfer) return;
    h_xfer = pinned_alloc(XFER_UP + 64);
    memset(h_xfer + XFER_UP, 0, 64);
    ensure_down(XFER_DOWN);
}
// The download area grows with the largest transfer seen (a lock-step group of K segments brings K caps, K sets of opening partials, K
// tables' query rounds down in one trip): power-of-two sizes up to XFER_DOWN_MAX; nothing is in flight into the old area when it goes
// (downloads are waited for before their caller returns).
void zkm_ctx::ensure_down(size_t bytes) {
    if (bytes <= down_cap) return;
    if (bytes > XFER_DOWN_MAX) throw std::runtime_error("internal: download beyond the pinned area's maximum");
    size_t cap = down_cap ? down_cap : XFER_DOWN;
    while (cap < bytes) cap <<= 1;
    if (h_down) {
        ZKM_HIP_CHECK(hipStreamSynchronize(stream));
        (void)hipHostFree(h_down);
        h_down = nullptr;
        down_cap = 0;
    }
    h_down = pinned_alloc(cap);
    down_cap = cap;
}
// Waiting for the flag.  A spinning thread sees the words ~40 us earlier than one the runtime has to wake -- as long as it has a core
// to spin on.  The deployed shape is 8 ranks x k contexts x ZKM_COMMIT_LANES threads on one host, each rank confined to its GPU's share
// of the cores (zkm_amd/dist.py pin_to_gpu): once the threads waiting in this process outnumber half of the CPUs it may run on,
// spinning steals the cores the other contexts need to launch their kernels.  So: spin for `block_after_us` (default 50 us: most round
// trips of an idle GPU end inside it); after that, a crowded process parks the thread on a blocking-sync event (interrupt-driven
// wake-up) while an uncrowded one keeps polling with sched_yield.
static std::atomic<int> g_live_contexts{0};
int zkm_live_contexts() { return g_live_contexts.load(std::memory_order_relaxed); }
static std::atomic<int> g_waiting{0};
static int allowed_cpus() {      // the affinity mask, capped by the container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota: a box that
    static const int n = [] {    // shows 256 CPUs may grant 16 -- the GPU boxes of this build do)
        int cpus = 1 << 20;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) cpus = (int)CPU_COUNT(&set);
        long long quota = -1, period = 100000;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%lld", &period) != 1) period = 100000;
                fclose(h);
            }
        }
        if (quota > 0 && period > 0) {
            const int q = (int)((quota + period - 1) / period);
            if (q >= 1 && q < cpus) cpus = q;
        }
        return cpus;
    }();
    return n;
}
static inline uint64_t now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
void zkm_ctx::wait_flag(const uint64_t* flag, uint64_t seq) {
    struct waiter { waiter() { g_waiting.fetch_add(1, std::memory_order_relaxed); } ~waiter() { g_waiting.fetch_sub(1, std::memory_order_relaxed); } } w;
    const uint64_t t0 = now_ns(), spin_ns = (parent ? parent : this)->block_after_us * 1000ull;
    for (uint64_t spins = 1;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return;
        if (spins % 256 == 0 && now_ns() - t0 > spin_ns) {
            if (2 * g_waiting.load(std::memory_order_relaxed) > allowed_cpus() || spin_ns == 0) {
                if (!block_event) ZKM_HIP_CHECK(hipEventCreateWithFlags(&block_event, hipEventBlockingSync | hipEventDisableTiming));
                ZKM_HIP_CHECK(hipEventRecord(block_event, stream));      // behind k_download on the in-order stream
                ZKM_HIP_CHECK(hipEventSynchronize(block_event));         // (complete: the words are in host memory whatever the flag's cache line says)
                blocked_waits++;
                return;
            }
            if (spins % 8192 == 0) {                                  // has the stream failed (or finished unseen)?
                hipError_t e = hipStreamQuery(stream);
                if (e == hipSuccess) return;
                if (e != hipErrorNotReady) ZKM_HIP_CHECK(e);
            }
            sched_yield();                                            // a long wait (the GPU is busy with other contexts): let other threads run
        }
#if defined(__x86_64__)
        else __builtin_ia32_pause();
#endif
    }
}
uint64_t zkm_ctx::xfer_begin(size_t bytes, uint64_t** host_slot, uint64_t** flag, unsigned** counter) {
    ensure_xfer();
    ensure_down(bytes);
    if (!d_counter) {
        d_counter = (unsigned*)alloc(64);
        resident_bytes += 64;
        ZKM_HIP_CHECK(hipMemsetAsync(d_counter, 0, 64, stream));
    }
    *host_slot = (uint64_t*)h_down;
    *flag = (uint64_t*)(h_xfer + XFER_UP);
    *counter = d_counter;
    return ++down_seq;
}
unsigned long long* zkm_ctx::pow_best() {
    if (!d_pow_best) {
        d_pow_best = (unsigned long long*)alloc(8 * ZKM_MAX_SEG);      // one word per search of a stack
        resident_bytes += 8 * ZKM_MAX_SEG;
        ZKM_HIP_CHECK(hipMemsetAsync(d_pow_best, 0xff, 8 * ZKM_MAX_SEG, stream));
    }
    return d_pow_best;
}
void zkm_ctx::xfer_finish(uint64_t seq, void* dst, size_t bytes) {
    wait_flag((const uint64_t*)(h_xfer + XFER_UP), seq);
    up_off = 0;                                               // everything queued before the kernel has completed, uploads included
    memcpy(dst, h_down, bytes);
}
void zkm_ctx::download(std::initializer_list<xfer> xs) {
    ensure_xfer();
    constexpr size_t XFER_KERNEL = (size_t)1 << 16;
    size_t total = 0, nx = 0;
    bool words_ok = true;
    for (const xfer& x : xs) {
        total += (x.bytes + 63) & ~(size_t)63;
        nx += x.bytes != 0;
        words_ok = words_ok && x.bytes % 4 == 0 && (uintptr_t)x.src % 4 == 0;
    }
    if (total <= XFER_DOWN_MAX) ensure_down(total);
    if (nx && nx <= 4 && total <= XFER_KERNEL && words_ok) {
        down_args a{};
        size_t off = 0;
        for (const xfer& x : xs) {
            if (!x.bytes) continue;
            a.src[a.n] = (const uint32_t*)x.src;
            a.dst[a.n] = (uint32_t*)(h_down + off);
            a.words[a.n] = (uint32_t)(x.bytes / 4);
            a.n++;
            off += (x.bytes + 63) & ~(size_t)63;
        }
        uint64_t* flag = (uint64_t*)(h_xfer + XFER_UP);
        const uint64_t seq = ++down_seq;
        hipLaunchKernelGGL(k_download, dim3(1), dim3(1024), 0, stream, a, flag, seq);
        ZKM_HIP_CHECK(hipGetLastError());
        wait_flag(flag, seq);
        up_off = 0;                                               // everything queued before the kernel has completed, uploads included
        off = 0;
        for (const xfer& x : xs) {
            if (!x.bytes) continue;
            memcpy(x.dst, h_down + off, x.bytes);
            off += (x.bytes + 63) & ~(size_t)63;
        }
        return;
    }
    if (nx == 1 && total <= down_cap) {
        const xfer* one = nullptr;
        for (const xfer& x : xs)
            if (x.bytes) one = &x;
        if (one->bytes % 16 == 0 && (uintptr_t)one->src % 16 == 0) {
            uint64_t *slot, *flag;
            unsigned* counter;
            const uint64_t seq = xfer_begin(one->bytes, &slot, &flag, &counter);
            const uint32_t n16 = (uint32_t)(one->bytes / 16);
            const unsigned grid = std::min<unsigned>(64, (n16 + 255) / 256);
            hipLaunchKernelGGL(k_download_wide, dim3(grid), dim3(256), 0, stream, (const uint4*)one->src, (uint4*)slot, n16, flag, seq, counter);
            ZKM_HIP_CHECK(hipGetLastError());
            xfer_finish(seq, one->dst, one->bytes);
            return;
        }
    }
    size_t off = 0;
    for (const xfer& x : xs) {
        if (!x.bytes) continue;
        const bool small = off + x.bytes <= down_cap;
        ZKM_HIP_CHECK(hipMemcpyAsync(small ? (void*)(h_down + off) : x.dst, x.src, x.bytes, hipMemcpyDeviceToHost, stream));
        if (small) off += (x.bytes + 63) & ~(size_t)63;
    }
    sync();
    off = 0;
    for (const xfer& x : xs) {
        if (!x.bytes || off + x.bytes > down_cap) continue;
        memcpy(x.dst, h_down + off, x.bytes);
        off += (x.bytes + 63) & ~(size_t)63;
    }
}
// Small uploads (challenge powers, query indices, descriptors: <= 64 KB) are copied by a one-workgroup KERNEL reading the pinned ring
// slot, not by the copy engines: a hipMemcpyAsync of a few hundred bytes queues behind whatever the engines are busy with -- with the
// next proof's trace being staged behind the current proof (zkm_trace_stage, 2.2 GB in 268 MB pieces) every transcript round trip waited
// for a piece: 70 ms per proof instead of 58 (round 6).  The kernel runs on the context's compute stream like the work that needs it.
static std::atomic<int> g_staged_live{0};     // staged traces / segments outstanding in this process (zkm_trace_stage .. zkm_staged_free)
__global__ __launch_bounds__(256) void k_upload_small(void* __restrict__ dst, const void* __restrict__ src, size_t bytes) {
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 7) == 0) {
        for (size_t i = threadIdx.x; i < bytes / 8; i += 256) ((uint64_t*)dst)[i] = ((const uint64_t*)src)[i];
    } else if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 3) == 0) {
        for (size_t i = threadIdx.x; i < bytes / 4; i += 256) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
    } else {
        for (size_t i = threadIdx.x; i < bytes; i += 256) ((uint8_t*)dst)[i] = ((const uint8_t*)src)[i];
    }
}
void zkm_ctx::upload(void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    if (bytes > XFER_UP / 4) {   // large: straight from the caller's memory, and waited for (the runtime may pin `src` and copy later)
        ZKM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        sync();
        return;
    }
    ensure_xfer();
    if (up_off + bytes > XFER_UP) sync();                        // the ring is full: wait for the uploads in flight (sync() rewinds it)
    char* slot = h_xfer + up_off;
    memcpy(slot, src, bytes);
    // ... but only while some context of the process has a staged trace outstanding: on idle copy engines the hipMemcpyAsync is the faster of
    // the two for a chain of dependent launches (one 2^16-cycle segment alone: 28.9 ms against 29.3 with the kernel, ~120 small uploads)
    static const int forced = getenv("ZKM_UPLOAD_KERNEL") ? atoi(getenv("ZKM_UPLOAD_KERNEL")) : -1;              // (measurement aid: 0 / 1)
    const bool by_kernel = forced >= 0 ? forced != 0 : g_staged_live.load(std::memory_order_relaxed) > 0;
    if (by_kernel) {
        hipLaunchKernelGGL(k_upload_small, dim3(1), dim3(256), 0, stream, dst, (const void*)slot, bytes);
        ZKM_HIP_CHECK(hipGetLastError());
    } else {
        ZKM_HIP_CHECK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, stream));
    }
    up_off += (bytes + 63) & ~(size_t)63;
}
hipEvent_t zkm_ctx::get_event() {
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    ZKM_HIP_CHECK(hipEventCreate(&e));
    return e;
}
size_t zkm_ctx::prof_begin(const char* name) {
    zkm_prof_rec r{name, get_event(), get_event()};
    ZKM_HIP_CHECK(hipEventRecord(r.start, stream));
    prof.push_back(r);
    prof_agg_valid = false;
    return prof.size() - 1;
}
void zkm_ctx::prof_end(size_t idx) {
    // (a destructor must not throw: a failed record leaves the pair unmeasured, prof_aggregate reports 0 ms for it)
    if (idx < prof.size()) (void)hipEventRecord(prof[idx].stop, stream);
}

// Stark::lookups() of the tables with constraint kernels.  Memory: RANGE_CHECK (10) looked up in COUNTER (11) with
// FREQUENCIES (12), memory_stark.rs:476-483.
static const uint32_t MEMORY_LOOKUP_COLS[1] = {10};
static const zkm_table_lookup MEMORY_LOOKUPS[1] = {{1, MEMORY_LOOKUP_COLS, 11, 12}};
// Arithmetic: the 18 shared columns (26..43) looked up in RANGE_COUNTER (44) with RC_FREQUENCIES (45), arithmetic_stark.rs:269-276.
static const uint32_t ARITH_LOOKUP_COLS[18] = {26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43};
static const zkm_table_lookup ARITH_LOOKUPS[1] = {{18, ARITH_LOOKUP_COLS, 44, 45}};
const zkm_table_lookup* zkm_table_lookups(int table_id, size_t* n) {
    if (table_id == ZKM_TABLE_MEMORY) { *n = 1; return MEMORY_LOOKUPS; }
    if (table_id == ZKM_TABLE_ARITHMETIC) { *n = 1; return ARITH_LOOKUPS; }
    *n = 0;
    return nullptr;
}

// The HIP runtime spreads the streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one queue
// run their kernels one after the other.  A process proving small segments with k contexts has k x ZKM_COMMIT_LANES streams (+ copy streams) whose
// kernels are short and meant to overlap: with 8 contexts, 43 segments/s at 4 queues, 50 at 8, 57 at 16, 47 at 32
// (profiles/r03_hw_queues.txt; the 2^20-row proofs do not care).  The variable is read when the runtime initialises, i.e. at the first
// HIP call of the PROCESS: it belongs to the launcher (bench.py, tools/run_config3.sh, the Rust host's main()) -- a
// shared library does not edit the environment of the process that loads it.  zkm_ctx_create reports an unset variable once, on stderr.
static void zkm_hw_queues_hint() {
    static std::atomic<bool> said{false};
    if (getenv("GPU_MAX_HW_QUEUES") || getenv("ZKM_QUIET") || said.exchange(true)) return;
    fprintf(stderr, "zkm-hip: GPU_MAX_HW_QUEUES is not set (runtime default: 4 hardware queues); small-segment throughput with several "
                    "contexts is ~25 %% higher with GPU_MAX_HW_QUEUES=16 exported before the first HIP call (see INTEGRATION.md)\n");
}

extern "C" {

const char* zkm_version(void) { return "zkm-hip 0.1 (gfx950)"; }

int zkm_ctx_create(int device, zkm_ctx** out, char** err) {
    ZKM_API_BEGIN
    int n = 0;
    ZKM_HIP_CHECK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) throw std::runtime_error("zkm_ctx_create: no such HIP device " + std::to_string(device));
    ZKM_HIP_CHECK(hipSetDevice(device));
    zkm_ctx* c = new zkm_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    ZKM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    if (const char* part = getenv("ZKM_CU_MASK_PART")) {
        int k = -1, n = 0;
        if (sscanf(part, "%d/%d", &k, &n) == 2) { c->cu_part_k = k; c->cu_part_n = n; }
    }
    ZKM_HIP_CHECK(zkm_stream_create(&c->stream, c->num_cus, c->cu_part_k, c->cu_part_n));
    zkm_hw_queues_hint();
    g_live_contexts.fetch_add(1, std::memory_order_relaxed);
    *out = c;
    ZKM_API_END(err)
}

void zkm_ctx_destroy(zkm_ctx* c) {
    if (!c) return;
    if (!c->parent) g_live_contexts.fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(c->device);
    for (zkm_ctx* l : c->lanes) zkm_ctx_destroy(l);
    c->lanes.clear();
    (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->copy_stream2) { (void)hipStreamSynchronize(c->copy_stream2); (void)hipStreamDestroy(c->copy_stream2); }
    for (auto& kv : c->free_blocks) (void)hipFree(kv.second);
    for (auto& kv : c->live_blocks) (void)hipFree(kv.first);
    for (auto& r : c->prof) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->h_staging) (void)hipHostFree(c->h_staging);
    if (c->h_xfer) (void)hipHostFree(c->h_xfer);
    if (c->h_down) (void)hipHostFree(c->h_down);
    if (c->block_event) (void)hipEventDestroy(c->block_event);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int zkm_ctx_set_tuning(zkm_ctx* c, const char* key, uint64_t value, char** err) {
    ZKM_API_BEGIN
    if (!c || !key) throw std::runtime_error("zkm_ctx_set_tuning: null argument");
    const std::string k(key);
    auto set = [&](zkm_ctx* x) {
        if (k == "ingest_chunk_cols") x->ingest_chunk_cols = (size_t)value;
        else if (k == "keccak_parts_max_points") x->keccak_parts_max_points = (size_t)value;
        else if (k == "fri_fused_division_min") x->fri_fused_division_min = (size_t)value;
        else if (k == "wide_max_hashes") x->wide_max_hashes = (size_t)value;
        else if (k == "leaf_mfma") x->leaf_mfma = value ? 1 : 0;
        else if (k == "quad_max_hashes") x->quad_max_hashes = (size_t)value;
        else if (k == "block_after_us") x->block_after_us = value;
        else if (k == "small_ntt") x->small_ntt = value ? 1 : 0;
        else if (k == "tree_tail") x->tree_tail = value ? 1 : 0;
        else if (k == "pow_round_log") x->pow_round_log = value < 8 ? 8 : (value > 22 ? 22 : (unsigned)value);
        else if (k == "fri_scan_combine") x->fri_scan_combine = value ? 1 : 0;
        else if (k == "aux_pipeline") x->aux_pipeline = value ? 1 : 0;
        else if (k == "commit_lanes") x->commit_lanes = value < 1 ? 1 : (value > 8 ? 8 : (size_t)value);
        else if (k == "segments_memory_budget") x->segments_memory_budget = (size_t)value;
        else if (k == "max_stack") x->max_stack = value < 1 ? 1 : (value > ZKM_MAX_SEG ? ZKM_MAX_SEG : (size_t)value);
        else if (k == "throughput_profile") {
            // MANY contexts on one GPU proving small segments (profiles/r04_throughput_profile.txt): one stream per context -- the runtime
            // has ~16 hardware queues, and streams that share one run behind each other's long kernels -- and the latency forms of the
            // permutation only where a launch is tiny (with a dozen independent chains in flight the issue slots they cost are somebody
            // else's work).  0 restores the defaults of a context that has the GPU (nearly) to itself.
            x->commit_lanes = value ? 1 : ZKM_COMMIT_LANES;
            x->wide_max_hashes = value ? 256 : 1024;
            x->quad_max_hashes = value ? 4096 : 32768;
            x->pow_round_log = value ? 16 : 17;                 // (half-filled SIMDs are somebody else's slots here: 76.3 vs 75.4 segments/s)
        }
        else if (k == "debug_fail_allocs") {
            // test hook, not a tuning: only a process that asks for the hooks (ZKM_ENABLE_TEST_HOOKS=1 in its environment) may set it
            const char* hooks = getenv("ZKM_ENABLE_TEST_HOOKS");
            if (!hooks || strcmp(hooks, "1") != 0) throw std::runtime_error("zkm_ctx_set_tuning: unknown key '" + k + "'");
            if (x == c) x->debug_fail_allocs.store((int)value);
        }
        else throw std::runtime_error("zkm_ctx_set_tuning: unknown key '" + k + "'");
    };
    set(c);
    for (zkm_ctx* l : c->lanes) set(l);
    ZKM_API_END(err)
}

int zkm_ctx_synchronize(zkm_ctx* c, char** err) {
    ZKM_API_BEGIN
    c->sync();
    ZKM_API_END(err)
}
void* zkm_ctx_stream(zkm_ctx* c) { return (void*)c->stream; }

void zkm_ctx_memory(const zkm_ctx* c, size_t* live_bytes, size_t* cached_bytes) {
    size_t live = 0, cached = 0;
    for (auto& kv : c->live_blocks) live += kv.second;
    for (auto& kv : c->free_blocks) cached += kv.first;
    for (const zkm_ctx* l : c->lanes) {
        size_t a = 0, b = 0;
        zkm_ctx_memory(l, &a, &b);
        live += a;
        cached += b;
    }
    if (live_bytes) *live_bytes = live;
    if (cached_bytes) *cached_bytes = cached;
}

size_t zkm_ctx_resident_bytes(const zkm_ctx* c) {
    size_t r = c->resident_bytes;
    for (const zkm_ctx* l : c->lanes) r += zkm_ctx_resident_bytes(l);
    return r;
}

void zkm_ctx_trim(zkm_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->trim();
}

// Pinned host memory for the witness generator's buffers: hipMemcpyAsync from pageable memory is staged by the runtime and
// blocks the host; from pinned memory the chunked upload of zkm_batch_build overlaps with compute.
int zkm_host_alloc(zkm_ctx* c, size_t bytes, void** out, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    ZKM_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault));
    ZKM_API_END(err)
}
int zkm_host_free(zkm_ctx* c, void* p) {
    (void)hipSetDevice(c->device);
    return hipHostFree(p) == hipSuccess ? 0 : 1;
}
int zkm_host_register(zkm_ctx* c, void* p, size_t bytes, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    ZKM_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    ZKM_API_END(err)
}
int zkm_host_unregister(zkm_ctx* c, void* p) {
    (void)hipSetDevice(c->device);
    return hipHostUnregister(p) == hipSuccess ? 0 : 1;
}

int zkm_dev_alloc(zkm_ctx* c, size_t bytes, void** out, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    *out = c->alloc(bytes);
    ZKM_API_END(err)
}
int zkm_dev_free(zkm_ctx* c, void* p) {
    c->release(p);
    return 0;
}
int zkm_dev_upload(zkm_ctx* c, void* dst, const void* src, size_t bytes, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));   // (the current device is per host thread; contexts are driven from worker threads)
    ZKM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    c->sync();
    ZKM_API_END(err)
}
int zkm_dev_download(zkm_ctx* c, void* dst, const void* src, size_t bytes, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    ZKM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    ZKM_API_END(err)
}

// ------------------------------------------------------------------ staged traces: the upload of the NEXT proof behind the CURRENT one
// The reference commits from host Vecs (prover/src/prover.rs:144-167); over PCIe a 262 x 2^20 trace is 2.2 GB = ~40 ms, most of a
// proof.  The intra-proof pipeline of zkm_batch_build (column chunks absorbed as they arrive) hides that behind hashing at the price of
// the chunked kernels (sponge state parked in HBM, short transforms, the multiply-add leaf form).  With the next segment's traces at
// hand while the current one is being proven -- the witness generator runs ahead of the prover -- the upload belongs BEHIND THE
// PREVIOUS PROOF instead: zkm_trace_stage queues it on the context's two copy streams (alternate pieces of >= 64 MB)
// and returns; the proof that consumes it runs the device-resident path at full speed.
struct zkm_staged {
    zkm_ctx* ctx;
    gl_t* dev;
    size_t words;
    hipEvent_t done[2];
    bool joined, canonical;
    size_t off[13];     // a staged SEGMENT: word offset of table t (Table::all() order) in the block, off[12] = words; one matrix: unused
    bool segment;
};

static zkm_staged* stage_begin(zkm_ctx* c, size_t words, int canonical) {
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    for (hipStream_t* cs : {&c->copy_stream, &c->copy_stream2})
        if (!*cs) {
            hipStream_t st = nullptr;
            ZKM_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            std::lock_guard<std::mutex> g(c->alloc_mu);      // (published under the allocator's lock, like zkm_batch_build's)
            *cs = st;
        }
    zkm_staged* s = new zkm_staged();
    s->ctx = c; s->words = words; s->joined = false; s->canonical = canonical != 0;
    s->done[0] = s->done[1] = nullptr;
    s->segment = false;
    for (size_t& o : s->off) o = 0;
    try {
        s->dev = (gl_t*)c->alloc(words * sizeof(gl_t));
        // the block may be one a finished call of this context released: whatever the compute stream still has queued on it comes first
        hipEvent_t e = c->get_event();
        ZKM_HIP_CHECK(hipEventRecord(e, c->stream));
        ZKM_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, e, 0));
        ZKM_HIP_CHECK(hipStreamWaitEvent(c->copy_stream2, e, 0));
        c->event_pool.push_back(e);
    } catch (...) {
        if (s->dev) c->release(s->dev);
        delete s;
        throw;
    }
    g_staged_live.fetch_add(1, std::memory_order_relaxed);
    return s;
}
static void stage_end(zkm_staged* s) {
    zkm_ctx* c = s->ctx;
    s->done[0] = c->get_event();
    s->done[1] = c->get_event();
    ZKM_HIP_CHECK(hipEventRecord(s->done[0], c->copy_stream));
    ZKM_HIP_CHECK(hipEventRecord(s->done[1], c->copy_stream2));
}
static void stage_abort(zkm_staged* s) {
    g_staged_live.fetch_sub(1, std::memory_order_relaxed);
    (void)hipStreamSynchronize(s->ctx->copy_stream);
    (void)hipStreamSynchronize(s->ctx->copy_stream2);
    s->ctx->release(s->dev);
    for (hipEvent_t e : s->done)
        if (e) s->ctx->event_pool.push_back(e);
    delete s;
}

int zkm_trace_stage(zkm_ctx* c, const uint64_t* values, size_t ncols, unsigned log_n, int canonical, zkm_staged** out, char** err) {
    ZKM_API_BEGIN
    if (!c || !values || !out || !ncols || log_n > 30) throw std::runtime_error("zkm_trace_stage: bad argument");
    const size_t n = (size_t)1 << log_n;
    zkm_staged* s = stage_begin(c, ncols * n, canonical);
    try {
        // pieces of >= 64 MB (8 columns at 2^20 rows; a short table is ONE copy): the two streams take alternate pieces; small pieces let
        // the other contexts' copies interleave (4 contexts at 2^20 rows: 16.3 proofs/s against 15.6 with 268 MB pieces), tiny ones are
        // all call overhead (a 2431-column table of 2^10 rows in 8-column pieces was 304 copies of 64 KB)
        size_t piece = std::max<size_t>(8, (((size_t)64 << 20) / (n * sizeof(gl_t)) + 7) / 8 * 8);
        if (const char* e = getenv("ZKM_STAGE_PIECE_COLS")) piece = std::max<size_t>(1, (size_t)atoi(e));   // (measurement aid)
        for (size_t c0 = 0, k = 0; c0 < ncols; c0 += piece, k++) {
            const size_t nc = std::min(piece, ncols - c0);
            ZKM_HIP_CHECK(hipMemcpyAsync(s->dev + c0 * n, values + c0 * n, nc * n * sizeof(gl_t), hipMemcpyHostToDevice, (k & 1) ? c->copy_stream2 : c->copy_stream));
        }
        stage_end(s);
    } catch (...) {
        stage_abort(s);
        throw;
    }
    *out = s;
    ZKM_API_END(err)
}
int zkm_trace_stage_columns(zkm_ctx* c, const uint64_t* const* columns, size_t ncols, unsigned log_n, int canonical, zkm_staged** out, char** err) {
    ZKM_API_BEGIN
    if (!c || !columns || !out || !ncols || log_n > 30) throw std::runtime_error("zkm_trace_stage_columns: bad argument");
    for (size_t i = 0; i < ncols; i++)
        if (!columns[i]) throw std::runtime_error("zkm_trace_stage_columns: null column pointer");
    const size_t n = (size_t)1 << log_n;
    zkm_staged* s = stage_begin(c, ncols * n, canonical);
    try {
        for (size_t i = 0; i < ncols; i++)
            ZKM_HIP_CHECK(hipMemcpyAsync(s->dev + i * n, columns[i], n * sizeof(gl_t), hipMemcpyHostToDevice, ((i / 8) & 1) ? c->copy_stream2 : c->copy_stream));
        stage_end(s);
    } catch (...) {
        stage_abort(s);
        throw;
    }
    *out = s;
    ZKM_API_END(err)
}
// All twelve tables of ONE segment in one call (Table::all() order, zkm_table_width columns x 2^log_n[t] words each): one block, one pair
// of events -- a lock-step call of K segments is K stage calls instead of 12 K (96 calls per 8-segment call cost its host thread 9 ms of
// the 600 the call takes).  Exactly one of traces / columns is non-null.
static int stage_segment(const char* what, zkm_ctx* c, const uint64_t* const* traces, const uint64_t* const* const* columns, const unsigned* log_n,
                         int canonical, zkm_staged** out, char** err) {
    ZKM_API_BEGIN
    if (!c || (!traces && !columns) || !log_n || !out) throw std::runtime_error(std::string(what) + ": null argument");
    static const int order[12] = {ZKM_TABLE_ARITHMETIC, ZKM_TABLE_CPU, ZKM_TABLE_POSEIDON, ZKM_TABLE_POSEIDON_SPONGE, ZKM_TABLE_KECCAK,
                                  ZKM_TABLE_KECCAK_SPONGE, ZKM_TABLE_SHA_EXTEND, ZKM_TABLE_SHA_EXTEND_SPONGE, ZKM_TABLE_SHA_COMPRESS,
                                  ZKM_TABLE_SHA_COMPRESS_SPONGE, ZKM_TABLE_LOGIC, ZKM_TABLE_MEMORY};   // Table::all(), all_stark.rs:117-134
    size_t off[13], W[12];
    off[0] = 0;
    for (int t = 0; t < 12; t++) {
        if (log_n[t] > 30 || (traces && !traces[t]) || (columns && !columns[t])) throw std::runtime_error(std::string(what) + ": bad table");
        W[t] = zkm_table_width(order[t]);
        off[t + 1] = off[t] + (W[t] << log_n[t]);
    }
    zkm_staged* s = stage_begin(c, off[12], canonical);
    try {
        s->segment = true;
        for (int t = 0; t <= 12; t++) s->off[t] = off[t];
        size_t k = 0, bytes_on[2] = {0, 0};
        for (int t = 0; t < 12; t++) {
            const size_t n = (size_t)1 << log_n[t];
            const size_t piece = std::max<size_t>(8, (((size_t)64 << 20) / (n * sizeof(gl_t)) + 7) / 8 * 8);
            for (size_t c0 = 0; c0 < W[t]; c0 += columns ? 1 : piece) {
                const size_t nc = columns ? 1 : std::min(piece, W[t] - c0);
                if (!columns || c0 % piece == 0) k = bytes_on[0] <= bytes_on[1] ? 0 : 1;     // the less loaded stream takes the next piece
                if (columns && !columns[t][c0]) throw std::runtime_error(std::string(what) + ": null column pointer");
                const void* src = columns ? (const void*)columns[t][c0] : (const void*)(traces[t] + c0 * n);
                ZKM_HIP_CHECK(hipMemcpyAsync(s->dev + off[t] + c0 * n, src, nc * n * sizeof(gl_t), hipMemcpyHostToDevice, k ? c->copy_stream2 : c->copy_stream));
                bytes_on[k] += nc * n * sizeof(gl_t);
            }
        }
        stage_end(s);
    } catch (...) {
        stage_abort(s);
        throw;
    }
    *out = s;
    ZKM_API_END(err)
}
int zkm_segment_stage(zkm_ctx* c, const uint64_t* const* traces, const unsigned* log_n, int canonical, zkm_staged** out, char** err) {
    return stage_segment("zkm_segment_stage", c, traces, nullptr, log_n, canonical, out, err);
}
int zkm_segment_stage_columns(zkm_ctx* c, const uint64_t* const* const* columns, const unsigned* log_n, int canonical, zkm_staged** out, char** err) {
    return stage_segment("zkm_segment_stage_columns", c, nullptr, columns, log_n, canonical, out, err);
}
// the twelve device matrices of a staged segment (each as zkm_staged_ptr gives a single matrix: ordered behind the upload)
int zkm_staged_segment_ptrs(zkm_staged* s, const uint64_t** ptrs_out) {
    if (!s || !s->segment || !ptrs_out) return 1;
    const uint64_t* base = zkm_staged_ptr(s);
    if (!base) return 1;
    for (int t = 0; t < 12; t++) ptrs_out[t] = base + s->off[t];
    return 0;
}

// The device matrix, ordered behind its upload on the context's compute stream (a device-side wait: the host does not block); the
// first call also canonicalises words the caller did not vouch for.
const uint64_t* zkm_staged_ptr(zkm_staged* s) {
    if (!s) return nullptr;
    zkm_ctx* c = s->ctx;
    if (!s->joined) {
        (void)hipSetDevice(c->device);
        if (hipStreamWaitEvent(c->stream, s->done[0], 0) != hipSuccess || hipStreamWaitEvent(c->stream, s->done[1], 0) != hipSuccess) return nullptr;
        try {
            if (!s->canonical) zkm_launch_canon(c, s->dev, s->words);
        } catch (...) {
            return nullptr;
        }
        s->joined = true;
    }
    return s->dev;
}
// host-side: has the upload finished (1), is it still in flight (0)?  `wait` != 0 blocks until it has.
int zkm_staged_ready(zkm_staged* s, int wait) {
    if (!s) return 1;
    (void)hipSetDevice(s->ctx->device);
    for (hipEvent_t e : s->done) {
        if (wait) {
            if (hipEventSynchronize(e) != hipSuccess) return -1;
        } else {
            const hipError_t q = hipEventQuery(e);
            if (q == hipErrorNotReady) return 0;
            if (q != hipSuccess) return -1;
        }
    }
    return 1;
}
void zkm_staged_free(zkm_staged* s) {
    if (!s) return;
    zkm_ctx* c = s->ctx;
    (void)hipSetDevice(c->device);
    // the host source may go away after this returns, and the block goes back to an allocator whose blocks are reused by later work of
    // the COMPUTE stream only: the uploads must have landed
    (void)hipEventSynchronize(s->done[0]);
    (void)hipEventSynchronize(s->done[1]);
    c->release(s->dev);
    c->event_pool.push_back(s->done[0]);
    c->event_pool.push_back(s->done[1]);
    delete s;
    g_staged_live.fetch_sub(1, std::memory_order_relaxed);
}

// ------------------------------------------------------------------ profiling
void zkm_profile_enable(zkm_ctx* c, int on) {
    c->profiling = on != 0;
    for (zkm_ctx* l : c->lanes) l->profiling = c->profiling;
}
void zkm_profile_reset(zkm_ctx* c) {
    (void)hipStreamSynchronize(c->stream);
    for (auto& r : c->prof) { c->event_pool.push_back(r.start); c->event_pool.push_back(r.stop); }
    c->prof.clear();
    c->prof_agg.clear();
    c->prof_agg_valid = false;
    for (zkm_ctx* l : c->lanes) zkm_profile_reset(l);
}
// (records of the commit lanes are reported with the context's: launches that ran side by side on different lanes each count
// their own duration, so a sum over kernels can exceed the wall time of a segment)
static void prof_collect(zkm_ctx* from, std::vector<zkm_ctx::agg>& agg) {
    (void)hipStreamSynchronize(from->stream);
    for (auto& r : from->prof) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) ms = 0;
        bool found = false;
        for (auto& a : agg)
            if (a.name == r.name || !strcmp(a.name, r.name)) { a.launches++; a.ms += ms; found = true; break; }
        if (!found) agg.push_back({r.name, 1, (double)ms});
    }
}
static void prof_aggregate(zkm_ctx* c) {
    bool lane_records = false;
    for (zkm_ctx* l : c->lanes) lane_records = lane_records || !l->prof.empty();
    if (c->prof_agg_valid && !lane_records) return;
    c->prof_agg.clear();
    prof_collect(c, c->prof_agg);
    for (zkm_ctx* l : c->lanes) prof_collect(l, c->prof_agg);
    c->prof_agg_valid = !lane_records;   // (lane records may still grow without invalidating this context's flag)
}
size_t zkm_profile_count(zkm_ctx* c) {
    prof_aggregate(c);
    return c->prof_agg.size();
}
int zkm_profile_get(zkm_ctx* c, size_t i, const char** name, uint64_t* launches, double* total_ms) {
    prof_aggregate(c);
    if (i >= c->prof_agg.size()) return 1;
    *name = c->prof_agg[i].name;
    *launches = c->prof_agg[i].launches;
    *total_ms = c->prof_agg[i].ms;
    return 0;
}

// ------------------------------------------------------------------ Fiat-Shamir (host)
// plonky2 Challenger: overwrite-mode duplex sponge over the Poseidon permutation.
}  // extern "C"

// (zkm_host_poseidon_permute: host_poseidon.hip)

extern "C" {
void zkm_challenger_init(zkm_challenger* ch) { memset(ch, 0, sizeof *ch); }
static void challenger_duplex(zkm_challenger* c) {
    for (uint32_t i = 0; i < c->n_in; i++) c->state[i] = c->in_buf[i];
    c->n_in = 0;
    zkm_host_poseidon_permute(c->state);
    for (int i = 0; i < 8; i++) c->out_buf[i] = c->state[i];
    c->n_out = 8;
}
void zkm_challenger_observe(zkm_challenger* c, const uint64_t* e, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->n_out = 0;
        c->in_buf[c->n_in++] = e[i];
        if (c->n_in == 8) challenger_duplex(c);
    }
}
uint64_t zkm_challenger_get(zkm_challenger* c) {
    if (c->n_in != 0 || c->n_out == 0) challenger_duplex(c);
    return c->out_buf[--c->n_out];
}
void zkm_challenger_compact(zkm_challenger* c, uint64_t out[12]) {
    if (c->n_in != 0) challenger_duplex(c);
    c->n_out = 0;
    memcpy(out, c->state, sizeof c->state);
}

void zkm_standard_config(zkm_stark_config* c) {
    c->rate_bits = 2; c->cap_height = 4; c->pow_bits = 16; c->num_challenges = 2;
    c->num_queries = 37; c->arity_bits = 4; c->final_poly_bits = 5;
}
}  // extern "C"

// ------------------------------------------------------------------ PolynomialBatch
// from_values: iNTT each column (coefficients kept), then from_coeffs: coset LDE (x 2^rate_bits, shift g),
// rows in bit-reversed order, Poseidon Merkle tree, cap.  src may be a host or device pointer.
// Values that arrive from the host and STAY on the device for other readers (dev_values: the CTL / lookup column kernels of
// prove_with_traces read them with canonical arithmetic) are canonicalised once after the upload: a plonky2 GoldilocksField may hold
// any u64 representing its residue (products are reduced to < 2^64, not < p), and the Rust side hands its columns over as they lie.
// The transforms themselves accept any representative (loose arithmetic), so the commitment is that of the same polynomial.
__global__ __launch_bounds__(256) void k_canon(gl_t* __restrict__ v, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) v[i] = gl_canon(v[i]);
}
// The traces of a lock-step group gathered into their stack, canonical on the way (one read + one write per word instead of a copy and
// an in-place pass): segment blockIdx.y's `words` words from its own device block.
struct seg_ptrs { const uint64_t* p[ZKM_MAX_SEG]; };
__global__ __launch_bounds__(256) void k_gather_canon(seg_ptrs src, size_t words, gl_t* __restrict__ dst) {
    const uint64_t* __restrict__ s = src.p[blockIdx.y];
    gl_t* __restrict__ d = dst + (size_t)blockIdx.y * words;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) d[i] = gl_canon(s[i]);
}

void zkm_launch_canon(zkm_ctx* c, gl_t* v, size_t total) {
    zkm_prof_scope ps(c, "ingest_canonicalise");
    hipLaunchKernelGGL(k_canon, dim3((total + 255) / 256), dim3(256), 0, c->stream, v, total);
    ZKM_HIP_CHECK(hipGetLastError());
}

// src_cols (optional, instead of src): one pointer per column, each to n words -- the reference's Vec<PolynomialValues<F>> is one heap
// allocation per column (prover.rs:154-163), so the Rust side hands the column pointers over instead of flattening 2 GiB on the host.
// A stacked batch (b->nseg > 1: the same-shaped polynomials of nseg segments, zkm_internal.h) takes its source as ONE block of nseg x
// ncols x n words (src), as nseg x ncols column pointers (src_cols), or as one pointer per segment (seg_srcs: ncols x n words each, host
// or device -- the traces of the segments of a lock-step group); everything below then works on nseg * ncols columns.
void zkm_batch_build(zkm_batch* b, const uint64_t* src, bool src_is_values, gl_t* dev_values, const uint64_t* const* src_cols,
                     const uint64_t* const* seg_srcs) {
    zkm_ctx* c = b->ctx;
    const size_t nseg = b->nseg;
    size_t n = b->n(), N = b->N(), ncols = b->ncols * nseg;   // (ncols: all columns of the stack)
    if (b->log_n + b->rate_bits > 30) throw std::runtime_error("polynomial batch too large");
    if (nseg == 0 || nseg > ZKM_MAX_SEG) throw std::runtime_error("polynomial batch: bad segment count");
    if (ncols > 65535 && nseg > 1) throw std::runtime_error("polynomial batch: too many stacked columns");
    if (!src && !src_cols && !seg_srcs) throw std::runtime_error("polynomial batch: no source");
    if (src_cols)
        for (size_t i = 0; i < ncols; i++)
            if (!src_cols[i]) throw std::runtime_error("polynomial batch: null column pointer");
    if (seg_srcs)
        for (size_t i = 0; i < nseg; i++)
            if (!seg_srcs[i]) throw std::runtime_error("polynomial batch: null segment pointer");
    // columns [c0, c0 + nc) of the source to dst (column stride n) on stream st
    auto copy_cols = [&](gl_t* dst, size_t c0, size_t nc, hipMemcpyKind kind, hipStream_t st) {
        if (seg_srcs) {     // (whole stack only)
            for (size_t sg = 0; sg < nseg; sg++)
                ZKM_HIP_CHECK(hipMemcpyAsync(dst + sg * b->ncols * n, seg_srcs[sg], b->ncols * n * sizeof(gl_t), hipMemcpyDefault, st));
            return;
        }
        if (!src_cols) {
            ZKM_HIP_CHECK(hipMemcpyAsync(dst, src + c0 * n, nc * n * sizeof(gl_t), kind, st));
            return;
        }
        for (size_t i = 0; i < nc; i++)
            ZKM_HIP_CHECK(hipMemcpyAsync(dst + i * n, src_cols[c0 + i], n * sizeof(gl_t), hipMemcpyDefault, st));
    };
    b->coeffs = (gl_t*)c->alloc(ncols * n * sizeof(gl_t));
    b->lde = (gl_t*)c->alloc(ncols * N * sizeof(gl_t));
    // Coefficient layout (zkm_internal.h): 2^18 .. 2^20 rows at rate 4 take the two-pass inverse transform, which leaves the coefficients
    // in the digit order the first LDE pass reads as contiguous runs; every batch of such a height has that layout (from_coeffs
    // converts its natural-order input after the LDE), so kernels that walk several batches position by position agree.
    b->coeff_s1 = b->rate_bits == 2 ? zkm_coeff_layout_s1(b->log_n) : 0;
    const unsigned s1 = b->coeff_s1;
    // values (nc columns, stride n, device) -> coefficients of columns [c0, c0 + nc)
    auto inverse_transform = [&](const gl_t* vals, size_t c0, size_t nc) {
        if (s1) zkm_intt_digit(c, vals, n, b->coeffs + c0 * n, n, nc, b->log_n);
        else zkm_ntt_natural_ex(c, vals, n, b->lde + c0 * N, n, b->coeffs + c0 * n, n, nc, b->log_n, /*inverse=*/true, 0);
    };
    size_t dwords = zkm_merkle_layout(b->lde_bits(), b->cap_height, b->level_off);
    b->dig_words = dwords;
    b->digests = (gl_t*)c->alloc(nseg * dwords * sizeof(gl_t));
    bool dev = !src_cols && !seg_srcs && zkm_is_device_ptr(src);   // (column / segment pointers are staged like host values, wherever each one lives)
    hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const size_t CH = c->ingest_chunk_cols;
    bool leaves_done = false;
    if (src_is_values && !dev && nseg == 1 && CH && CH % 8 == 0 && ncols >= 2 * CH && b->log_n >= 13) {  // (short tables: one upload, wide leaf hashing)
        // Host-resident values (prover.rs:144-167: the traces arrive as Vec<PolynomialValues>): pipelined ingest.  The upload is
        // split into chunks of CH columns on the copy stream; the compute stream transforms (iNTT, LDE) and ABSORBS chunk k
        // (leaf sponge, hash.hip k_merkle_leaves_chunk) while chunk k + 1 .. are in flight, so PCIe time hides behind hashing.
        // Each chunk is staged in the LDE region of its own columns (or lands in dev_values), so there is no buffer to recycle.
        if (!c->copy_stream) {   // (published under the allocator's lock: a relative's out-of-memory path reads it from its own thread)
            hipStream_t cs = nullptr;
            ZKM_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            std::lock_guard<std::mutex> g(c->alloc_mu);
            c->copy_stream = cs;
        }
        const size_t nchunks = (ncols + CH - 1) / CH;
        std::vector<hipEvent_t> ev;
        ev.reserve(nchunks + 1);
        zkm_scratch state(c, 12 * N * sizeof(gl_t));
        // Every exit path -- also a HIP error in the middle of the loop -- drains BOTH streams before the sponge state, the events
        // and (in the callers' unwinding) the batch's buffers go back to the allocator: uploads may still be in flight into them, and
        // the caller may free the host source as soon as this returns.
        auto drain = [&]() {
            (void)hipStreamSynchronize(c->copy_stream);
            (void)hipStreamSynchronize(c->stream);
            for (auto e : ev) c->event_pool.push_back(e);
            ev.clear();
        };
        try {
            ev.push_back(c->get_event());
            ZKM_HIP_CHECK(hipEventRecord(ev[0], c->stream));           // the LDE buffer may still be in use by queued work of a freed batch
            ZKM_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, ev[0], 0));
            for (size_t k = 0; k < nchunks; k++) {
                size_t c0 = k * CH, nc = std::min(CH, ncols - c0);
                gl_t* dst = dev_values ? dev_values + c0 * n : b->lde + c0 * N;
                copy_cols(dst, c0, nc, hipMemcpyHostToDevice, c->copy_stream);
                ev.push_back(c->get_event());
                ZKM_HIP_CHECK(hipEventRecord(ev.back(), c->copy_stream));
            }
            for (size_t k = 0; k < nchunks; k++) {
                size_t c0 = k * CH, nc = std::min(CH, ncols - c0);
                gl_t* vals = dev_values ? dev_values + c0 * n : b->lde + c0 * N;
                ZKM_HIP_CHECK(hipStreamWaitEvent(c->stream, ev[k + 1], 0));
                if (dev_values) zkm_launch_canon(c, vals, nc * n);
                inverse_transform(vals, c0, nc);
                zkm_lde_bitrev(c, b->coeffs + c0 * n, b->lde + c0 * N, nc, b->log_n, b->rate_bits, GL_GENERATOR, s1);
                zkm_launch_merkle_leaves_chunk(c, b->lde + c0 * N, N, nc, N, state.as<gl_t>(), k == 0, k + 1 == nchunks, b->digests);
            }
        } catch (...) {
            drain();
            throw;
        }
        drain();
        leaves_done = true;
    } else if (src_is_values && dev) {
        // device-resident values are only read (first NTT pass); the not-yet-used LDE buffer holds the intermediate passes
        inverse_transform(src, 0, ncols);
    } else if (src_is_values) {
        // stage the values in the (not yet used) LDE buffer (or the caller's device copy), transform, land natural-order coefficients
        gl_t* vals = dev_values ? dev_values : b->lde;
        bool all_dev = seg_srcs != nullptr;
        for (size_t sg = 0; all_dev && sg < nseg; sg++) all_dev = zkm_is_device_ptr(seg_srcs[sg]);
        if (all_dev && !dev_values) {
            // device-resident value matrices nobody needs a stacked copy of: every segment's inverse transform reads its matrix where
            // it lies and lands its coefficients in the segment's columns of the stack (no gather: 2 GiB per 262 x 2^20 matrix)
            for (size_t sg = 0; sg < nseg; sg++) inverse_transform(seg_srcs[sg], sg * b->ncols, b->ncols);
        } else if (all_dev) {     // device-resident traces of a lock-step group: gathered and canonicalised in one pass
            seg_ptrs sp{};
            for (size_t sg = 0; sg < nseg; sg++) sp.p[sg] = seg_srcs[sg];
            const size_t words = b->ncols * n;
            zkm_prof_scope ps(c, "ingest_canonicalise");
            hipLaunchKernelGGL(k_gather_canon, dim3((unsigned)std::min<size_t>((words + 255) / 256, 4096), (unsigned)nseg), dim3(256), 0, c->stream, sp,
                               words, vals);
            ZKM_HIP_CHECK(hipGetLastError());
        } else {
            copy_cols(vals, 0, ncols, kind, c->stream);
            if (dev_values) zkm_launch_canon(c, vals, ncols * n);
        }
        if (!(all_dev && !dev_values)) inverse_transform(vals, 0, ncols);
    } else {
        copy_cols(b->coeffs, 0, ncols, kind, c->stream);
    }
    if (!leaves_done) {
        zkm_lde_bitrev(c, b->coeffs, b->lde, ncols, b->log_n, b->rate_bits, GL_GENERATOR, src_is_values ? s1 : 0);
        zkm_launch_merkle_leaves(c, b->lde, N, b->ncols, N, b->digests, nseg, b->lde_seg(), dwords);
    }
    if (!src_is_values && s1) {
        // from_coeffs: the LDE above read the caller's natural order; the batch keeps the coefficients in the common layout
        zkm_scratch nat(c, ncols * n * sizeof(gl_t));
        ZKM_HIP_CHECK(hipMemcpyAsync(nat.p, b->coeffs, ncols * n * sizeof(gl_t), hipMemcpyDeviceToDevice, c->stream));
        zkm_coeff_layout_convert(c, nat.as<gl_t>(), n, b->coeffs, n, ncols, b->log_n, /*to_natural=*/false);
    }
    size_t capw = (size_t)4 << b->cap_height;
    b->cap.resize(nseg * capw);
    zkm_merkle_build_inner_cap(c, b->digests, b->level_off, b->lde_bits(), b->cap_height, b->cap.data(), nseg, dwords);
}

// out[i * ncols + col] = lde[col][bitrev((index_start + i) * step)]: lanes run along i, so reads of one column are scattered
// (bit-reversed rows) but writes are dense; the accessor is off the proving path
__global__ __launch_bounds__(256) void k_gather_lde_rows(const gl_t* __restrict__ lde, size_t N, unsigned lde_bits, size_t ncols,
                                                         size_t index_start, size_t step, size_t count, gl_t* __restrict__ out) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * ncols) return;
    size_t i = idx / ncols, col = idx % ncols;
    size_t row = bitrev32((uint32_t)((index_start + i) * step), lde_bits);
    out[idx] = lde[col * N + row];
}

static zkm_batch* batch_new(zkm_ctx* c, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height) {
    if (ncols == 0) throw std::runtime_error("empty polynomial batch");
    if (cap_height > log_n + rate_bits) throw std::runtime_error("cap_height exceeds LDE size");
    zkm_batch* b = new zkm_batch();
    b->ctx = c; b->ncols = ncols; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
    return b;
}

// from_values keeping the uploaded values in dev_values (see zkm_batch_build); throws
zkm_batch* zkm_batch_commit_values_keep(zkm_ctx* c, const uint64_t* values, size_t ncols, unsigned log_n, unsigned rate_bits,
                                        unsigned cap_height, gl_t* dev_values, const uint64_t* const* columns) {
    zkm_batch* b = batch_new(c, ncols, log_n, rate_bits, cap_height);
    try {
        zkm_batch_build(b, values, true, dev_values, columns);
    } catch (...) {
        zkm_batch_free(b);
        throw;
    }
    return b;
}

extern "C" {

int zkm_batch_commit_values(zkm_ctx* c, const uint64_t* values, size_t ncols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, zkm_batch** out, char** err) {
    zkm_batch* b = nullptr;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        b = batch_new(c, ncols, log_n, rate_bits, cap_height);
        zkm_batch_build(b, values, true);
        *out = b;
    } catch (const std::exception& e) {
        zkm_batch_free(b);
        return fail(err, e.what());
    }
    return 0;
}
int zkm_batch_commit_columns(zkm_ctx* c, const uint64_t* const* columns, size_t ncols, unsigned log_n, int columns_are_values,
                             unsigned rate_bits, unsigned cap_height, zkm_batch** out, char** err) {
    zkm_batch* b = nullptr;
    try {
        if (!c || !columns || !out) throw std::runtime_error("zkm_batch_commit_columns: null argument");
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        b = batch_new(c, ncols, log_n, rate_bits, cap_height);
        zkm_batch_build(b, nullptr, columns_are_values != 0, nullptr, columns);
        *out = b;
    } catch (const std::exception& e) {
        zkm_batch_free(b);
        return fail(err, e.what());
    } catch (...) {
        zkm_batch_free(b);
        return fail(err, "zkm_batch_commit_columns: unknown error");
    }
    return 0;
}
int zkm_batch_commit_coeffs(zkm_ctx* c, const uint64_t* coeffs, size_t ncols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, zkm_batch** out, char** err) {
    zkm_batch* b = nullptr;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        b = batch_new(c, ncols, log_n, rate_bits, cap_height);
        zkm_batch_build(b, coeffs, false);
        *out = b;
    } catch (const std::exception& e) {
        zkm_batch_free(b);
        return fail(err, e.what());
    }
    return 0;
}
void zkm_batch_free(zkm_batch* b) {
    if (!b) return;
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->ctx->copy_stream) (void)hipStreamSynchronize(b->ctx->copy_stream);
    b->ctx->release(b->coeffs);
    b->ctx->release(b->lde);
    b->ctx->release(b->digests);
    delete b;
}
int zkm_batch_cap(const zkm_batch* b, uint64_t* out) {
    memcpy(out, b->cap.data(), b->cap.size() * sizeof(uint64_t));
    return 0;
}
int zkm_batch_coeffs(const zkm_batch* b, uint64_t* out) {
    // .polynomials in NATURAL order, whatever layout the batch keeps them in
    zkm_ctx* c = b->ctx;
    const size_t bytes = b->ncols * b->n() * sizeof(gl_t);
    const bool dev = zkm_is_device_ptr(out);
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!b->coeff_s1) {
            ZKM_HIP_CHECK(hipMemcpyAsync(out, b->coeffs, bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        } else {
            zkm_scratch nat(c, dev ? 8 : bytes);
            gl_t* d = dev ? out : nat.as<gl_t>();
            zkm_coeff_layout_convert(c, b->coeffs, b->n(), d, b->n(), b->ncols, b->log_n, /*to_natural=*/true);
            if (!dev) ZKM_HIP_CHECK(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, c->stream));
            c->sync();   // (the scratch goes back to the allocator on return)
        }
        c->sync();
    } catch (...) {
        return 1;
    }
    return 0;
}
int zkm_batch_leaf(const zkm_batch* b, size_t leaf, uint64_t* out) {
    if (leaf >= b->N()) return 1;
    // one strided gather: ncols words at stride N
    if (hipMemcpy2DAsync(out, sizeof(gl_t), b->lde + leaf, b->N() * sizeof(gl_t), sizeof(gl_t), b->ncols, hipMemcpyDeviceToHost,
                         b->ctx->stream) != hipSuccess)
        return 1;
    return hipStreamSynchronize(b->ctx->stream) == hipSuccess ? 0 : 1;
}
int zkm_batch_lde_row(const zkm_batch* b, size_t natural_index, uint64_t* out) {
    if (natural_index >= b->N()) return 1;
    return zkm_batch_leaf(b, bitrev32((uint32_t)natural_index, b->lde_bits()), out);
}
int zkm_batch_lde_rows(const zkm_batch* b, size_t index_start, size_t step, size_t count, uint64_t* out) {
    // get_lde_values_packed(index_start, step) for `count` consecutive indices (prover.rs:687, 723-748): row i of the output is
    // get_lde_values(index_start + i, step) = leaves[reverse_bits((index_start + i) * step)], ncols words; out host or device
    if (!count) return 0;
    if (step == 0) return 1;
    const size_t last_ok = (b->N() - 1) / step;                     // largest index whose (index * step) is a row of the LDE
    if (index_start > last_ok || count - 1 > last_ok - index_start) return 1;   // (no wrap-around for hostile arguments)
    zkm_ctx* c = b->ctx;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        bool dev = zkm_is_device_ptr(out);
        zkm_scratch tmp(c, dev ? 8 : count * b->ncols * sizeof(gl_t));
        gl_t* d = dev ? out : tmp.as<gl_t>();
        size_t total = count * b->ncols;
        hipLaunchKernelGGL(k_gather_lde_rows, dim3((total + 255) / 256), dim3(256), 0, c->stream, b->lde, b->N(), b->lde_bits(), b->ncols,
                           index_start, step, count, d);
        ZKM_HIP_CHECK(hipGetLastError());
        if (!dev) ZKM_HIP_CHECK(hipMemcpyAsync(out, d, total * sizeof(gl_t), hipMemcpyDeviceToHost, c->stream));
        c->sync();
    } catch (...) {
        return 1;
    }
    return 0;
}
int zkm_batch_merkle_path(const zkm_batch* b, size_t leaf, uint64_t* sib) {
    if (leaf >= b->N()) return 1;
    for (unsigned l = 0; l < b->top(); l++)
        if (hipMemcpyAsync(sib + 4 * l, b->digests + b->level_off[l] + 4 * ((leaf >> l) ^ 1), 32, hipMemcpyDeviceToHost,
                           b->ctx->stream) != hipSuccess)
            return 1;
    return hipStreamSynchronize(b->ctx->stream) == hipSuccess ? 0 : 1;
}
int zkm_batch_digest_layer(const zkm_batch* b, unsigned level, uint64_t* out) {
    if (level > b->top()) return 1;
    size_t words = (size_t)4 << (b->lde_bits() - level);
    if (hipMemcpyAsync(out, b->digests + b->level_off[level], words * sizeof(gl_t), hipMemcpyDeviceToHost, b->ctx->stream) != hipSuccess)
        return 1;
    return hipStreamSynchronize(b->ctx->stream) == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------ NTT / hashes / trace
int zkm_ntt(zkm_ctx* c, uint64_t* cols, size_t ncols, unsigned log_n, int inverse, uint64_t coset_shift, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    if (log_n > 30) throw std::runtime_error("zkm_ntt: log_n too large");
    if (coset_shift >= GL_P) throw std::runtime_error("zkm_ntt: coset_shift not canonical");
    size_t n = (size_t)1 << log_n, bytes = ncols * n * sizeof(gl_t);
    if (bytes == 0) return 0;
    bool dev = zkm_is_device_ptr(cols);
    gl_t* scratch = (gl_t*)c->alloc(bytes);
    gl_t* work = dev ? cols : (gl_t*)c->alloc(bytes);
    // transform from scratch (copy of the input) into `work`
    ZKM_HIP_CHECK(hipMemcpyAsync(scratch, cols, bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    zkm_ntt_natural(c, scratch, work, ncols, n, n, log_n, inverse != 0, coset_shift);
    if (!dev) ZKM_HIP_CHECK(hipMemcpyAsync(cols, work, bytes, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->release(scratch);
    if (!dev) c->release(work);
    ZKM_API_END(err)
}

int zkm_poseidon_permute_batch(zkm_ctx* c, uint64_t* states, size_t k, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    size_t bytes = k * 12 * sizeof(uint64_t);
    if (!bytes) return 0;
    bool dev = zkm_is_device_ptr(states);
    gl_t* d = dev ? states : (gl_t*)c->alloc(bytes);
    if (!dev) ZKM_HIP_CHECK(hipMemcpyAsync(d, states, bytes, hipMemcpyHostToDevice, c->stream));
    zkm_launch_poseidon_permute(c, d, k);
    if (!dev) ZKM_HIP_CHECK(hipMemcpyAsync(states, d, bytes, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (!dev) c->release(d);
    ZKM_API_END(err)
}

int zkm_keccakf_batch(zkm_ctx* c, uint64_t* states, size_t k, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    size_t bytes = k * 25 * sizeof(uint64_t);
    if (!bytes) return 0;
    bool dev = zkm_is_device_ptr(states);
    uint64_t* d = dev ? states : (uint64_t*)c->alloc(bytes);
    if (!dev) ZKM_HIP_CHECK(hipMemcpyAsync(d, states, bytes, hipMemcpyHostToDevice, c->stream));
    zkm_launch_keccakf(c, d, k);
    if (!dev) ZKM_HIP_CHECK(hipMemcpyAsync(states, d, bytes, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (!dev) c->release(d);
    ZKM_API_END(err)
}

int zkm_poseidon_trace(zkm_ctx* c, uint64_t seed, size_t num_perms, unsigned log_n, uint64_t* out_dev, char** err) {
    ZKM_API_BEGIN
    ZKM_HIP_CHECK(hipSetDevice(c->device));
    if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error("zkm_poseidon_trace: out must be a device pointer");
    zkm_launch_poseidon_trace(c, seed, nullptr, nullptr, num_perms, log_n, out_dev);
    c->sync();
    ZKM_API_END(err)
}

// shared host side of the two sponge witness generators: rows per operation = len / rate + 1
static int sponge_trace(zkm_ctx* c, const char* what, size_t rate, bool poseidon, const uint8_t* inputs, const uint64_t* input_off,
                        const uint64_t* meta, size_t nops, unsigned log_n, uint64_t* out_dev, size_t* rows_used_out, char** err) {
    std::vector<void*> tmp;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error(std::string(what) + ": out must be a device pointer");
        size_t n = (size_t)1 << log_n;
        std::vector<uint64_t> row_off(nops + 1, 0);
        for (size_t i = 0; i < nops; i++) {
            if (input_off[i + 1] <= input_off[i]) throw std::runtime_error(std::string(what) + ": empty operation (base_address[0] is required)");
            row_off[i + 1] = row_off[i] + (input_off[i + 1] - input_off[i]) / rate + 1;
        }
        if (row_off[nops] > n) throw std::runtime_error(std::string(what) + ": operations need more rows than 2^log_n");
        size_t nbytes = nops ? input_off[nops] : 0;
        bool idev = zkm_is_device_ptr(inputs);
        const uint8_t* d_in = inputs;
        if (!idev && nbytes) {
            void* p = c->alloc(nbytes);
            tmp.push_back(p);
            ZKM_HIP_CHECK(hipMemcpyAsync(p, inputs, nbytes, hipMemcpyHostToDevice, c->stream));
            d_in = (const uint8_t*)p;
        }
        uint64_t* d_off = (uint64_t*)c->alloc((nops + 1) * 8);
        tmp.push_back(d_off);
        uint64_t* d_meta = (uint64_t*)c->alloc((nops ? nops : 1) * 32);
        tmp.push_back(d_meta);
        uint64_t* d_row = (uint64_t*)c->alloc((nops + 1) * 8);
        tmp.push_back(d_row);
        ZKM_HIP_CHECK(hipMemcpyAsync(d_off, input_off, (nops + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (nops) ZKM_HIP_CHECK(hipMemcpyAsync(d_meta, meta, nops * 32, hipMemcpyHostToDevice, c->stream));
        ZKM_HIP_CHECK(hipMemcpyAsync(d_row, row_off.data(), (nops + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (poseidon) zkm_launch_poseidon_sponge_trace(c, d_in, d_off, d_meta, d_row, nops, log_n, out_dev);
        else zkm_launch_keccak_sponge_trace(c, d_in, d_off, d_meta, d_row, nops, (size_t)row_off[nops], log_n, out_dev);
        c->sync();
        if (rows_used_out) *rows_used_out = row_off[nops];
        for (void* p : tmp) c->release(p);
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}

int zkm_keccak_sponge_trace(zkm_ctx* c, const uint8_t* inputs, const uint64_t* input_off, const uint64_t* meta, size_t nops,
                            unsigned log_n, uint64_t* out_dev, size_t* rows_used_out, char** err) {
    return sponge_trace(c, "zkm_keccak_sponge_trace", 136, false, inputs, input_off, meta, nops, log_n, out_dev, rows_used_out, err);
}

int zkm_poseidon_sponge_trace(zkm_ctx* c, const uint8_t* inputs, const uint64_t* input_off, const uint64_t* meta, size_t nops,
                              unsigned log_n, uint64_t* out_dev, size_t* rows_used_out, char** err) {
    return sponge_trace(c, "zkm_poseidon_sponge_trace", 32, true, inputs, input_off, meta, nops, log_n, out_dev, rows_used_out, err);
}

int zkm_poseidon_trace_inputs(zkm_ctx* c, const uint64_t* inputs, const uint64_t* timestamps, size_t num_perms, unsigned log_n,
                              uint64_t* out_dev, char** err) {
    std::vector<void*> tmp;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error("zkm_poseidon_trace_inputs: out must be a device pointer");
        if (num_perms > ((size_t)1 << log_n)) throw std::runtime_error("zkm_poseidon_trace_inputs: more permutations than 2^log_n rows");
        if (num_perms && (!inputs || !timestamps)) throw std::runtime_error("zkm_poseidon_trace_inputs: inputs and timestamps are required");
        auto to_dev = [&](const uint64_t* p, size_t words) -> const uint64_t* {
            if (!words || zkm_is_device_ptr(p)) return p;
            void* d = c->alloc(words * 8);
            tmp.push_back(d);
            ZKM_HIP_CHECK(hipMemcpyAsync(d, p, words * 8, hipMemcpyHostToDevice, c->stream));
            return (const uint64_t*)d;
        };
        const uint64_t* d_in = to_dev(inputs, num_perms * 12);
        const uint64_t* d_ts = to_dev(timestamps, num_perms);
        zkm_launch_poseidon_trace(c, 0, num_perms ? d_in : nullptr, num_perms ? d_ts : nullptr, num_perms, log_n, out_dev);
        c->sync();
        for (void* p : tmp) c->release(p);
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}

size_t zkm_num_lookup_columns(int table_id, const zkm_stark_config* cfg) {
    size_t nl = 0, total = 0;
    const zkm_table_lookup* d = zkm_table_lookups(table_id, &nl);
    for (size_t i = 0; i < nl; i++) total += ((d[i].ncols + 1) / 2 + 1) * cfg->num_challenges;
    return total;
}

size_t zkm_table_width(int table_id) {
    switch (table_id) {
        case ZKM_TABLE_POSEIDON: return ZKM_POSEIDON_COLS;
        case ZKM_TABLE_LOGIC: return ZKM_LOGIC_COLS;
        case ZKM_TABLE_KECCAK_SPONGE: return ZKM_KECCAK_SPONGE_COLS;
        case ZKM_TABLE_KECCAK: return ZKM_KECCAK_COLS;
        case ZKM_TABLE_MEMORY: return ZKM_MEMORY_COLS;
        case ZKM_TABLE_POSEIDON_SPONGE: return ZKM_POSEIDON_SPONGE_COLS;
        case ZKM_TABLE_SHA_EXTEND: return ZKM_SHA_EXTEND_COLS;
        case ZKM_TABLE_SHA_EXTEND_SPONGE: return ZKM_SHA_EXTEND_SPONGE_COLS;
        case ZKM_TABLE_SHA_COMPRESS: return ZKM_SHA_COMPRESS_COLS;
        case ZKM_TABLE_SHA_COMPRESS_SPONGE: return ZKM_SHA_COMPRESS_SPONGE_COLS;
        case ZKM_TABLE_ARITHMETIC: return ZKM_ARITHMETIC_COLS;
        case ZKM_TABLE_CPU: return ZKM_CPU_COLS;
        default: return 0;
    }
}

// host -> device staging of small argument arrays for the witness entry points
static const void* stage_arg(zkm_ctx* c, std::vector<void*>& tmp, const void* p, size_t bytes) {
    if (!bytes || zkm_is_device_ptr(p)) return p;
    void* d = c->alloc(bytes);
    tmp.push_back(d);
    ZKM_HIP_CHECK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream));
    return d;
}

int zkm_sha_extend_trace(zkm_ctx* c, const uint8_t* inputs, const uint64_t* timestamps, size_t nrows, unsigned log_n, uint64_t* out_dev,
                         char** err) {
    std::vector<void*> tmp;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error("zkm_sha_extend_trace: out must be a device pointer");
        size_t n = (size_t)1 << log_n;
        if (nrows > n) throw std::runtime_error("zkm_sha_extend_trace: more rows than 2^log_n");
        const uint8_t* d_in = (const uint8_t*)stage_arg(c, tmp, inputs, nrows * 16);
        const uint64_t* d_ts = (const uint64_t*)stage_arg(c, tmp, timestamps, nrows * 8);
        zkm_launch_sha_extend_trace(c, d_in, d_ts, nrows, n, out_dev);
        c->sync();
        for (void* p : tmp) c->release(p);
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}

int zkm_sha_extend_sponge_trace(zkm_ctx* c, const uint32_t* w16, const uint64_t* meta, size_t nblocks, unsigned log_n, uint64_t* out_dev,
                                char** err) {
    std::vector<void*> tmp;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error("zkm_sha_extend_sponge_trace: out must be a device pointer");
        size_t n = (size_t)1 << log_n;
        if (48 * nblocks > n) throw std::runtime_error("zkm_sha_extend_sponge_trace: message schedules need more rows than 2^log_n (48 each)");
        const uint32_t* d_w = (const uint32_t*)stage_arg(c, tmp, w16, nblocks * 64);
        const uint64_t* d_meta = (const uint64_t*)stage_arg(c, tmp, meta, nblocks * 32);
        zkm_launch_sha_extend_sponge_trace(c, d_w, d_meta, nblocks, n, out_dev);
        c->sync();
        for (void* p : tmp) c->release(p);
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}

static int sha_compress_trace(zkm_ctx* c, bool sponge, const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t ncomp,
                              unsigned log_n, uint64_t* out_dev, char** err) {
    std::vector<void*> tmp;
    const char* what = sponge ? "zkm_sha_compress_sponge_trace" : "zkm_sha_compress_trace";
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error(std::string(what) + ": out must be a device pointer");
        size_t n = (size_t)1 << log_n;
        if ((sponge ? 1 : 65) * ncomp > n) throw std::runtime_error(std::string(what) + ": compressions need more rows than 2^log_n");
        const uint32_t* d_hx = (const uint32_t*)stage_arg(c, tmp, hx, ncomp * 32);
        const uint32_t* d_w = (const uint32_t*)stage_arg(c, tmp, w, ncomp * 256);
        const uint64_t* d_meta = (const uint64_t*)stage_arg(c, tmp, meta, ncomp * 64);
        zkm_launch_sha_compress_trace(c, sponge, d_hx, d_w, d_meta, ncomp, n, out_dev);
        c->sync();
        for (void* p : tmp) c->release(p);
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}
int zkm_sha_compress_trace(zkm_ctx* c, const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t ncomp, unsigned log_n,
                           uint64_t* out_dev, char** err) {
    return sha_compress_trace(c, false, hx, w, meta, ncomp, log_n, out_dev, err);
}
int zkm_sha_compress_sponge_trace(zkm_ctx* c, const uint32_t* hx, const uint32_t* w, const uint64_t* meta, size_t ncomp, unsigned log_n,
                                  uint64_t* out_dev, char** err) {
    return sha_compress_trace(c, true, hx, w, meta, ncomp, log_n, out_dev, err);
}

int zkm_keccak_trace(zkm_ctx* c, const uint64_t* inputs, const uint64_t* timestamps, size_t nperms, unsigned log_n, uint64_t* out_dev,
                     char** err) {
    std::vector<void*> tmp;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error("zkm_keccak_trace: out must be a device pointer");
        size_t n = (size_t)1 << log_n;
        if (nperms * 24 > n) throw std::runtime_error("zkm_keccak_trace: permutations need more rows than 2^log_n (24 each)");
        auto to_dev = [&](const uint64_t* p, size_t words) -> const uint64_t* {
            if (!words || zkm_is_device_ptr(p)) return p;
            void* d = c->alloc(words * 8);
            tmp.push_back(d);
            ZKM_HIP_CHECK(hipMemcpyAsync(d, p, words * 8, hipMemcpyHostToDevice, c->stream));
            return (const uint64_t*)d;
        };
        const uint64_t* d_in = to_dev(inputs, nperms * 25);
        const uint64_t* d_ts = to_dev(timestamps, nperms);
        zkm_launch_keccak_trace(c, d_in, d_ts, nperms, n, out_dev);
        c->sync();
        for (void* p : tmp) c->release(p);
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}

int zkm_logic_trace(zkm_ctx* c, const uint32_t* ops, size_t nops, unsigned log_n, uint64_t* out_dev, char** err) {
    std::vector<void*> tmp;
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!zkm_is_device_ptr(out_dev)) throw std::runtime_error("zkm_logic_trace: out must be a device pointer");
        size_t n = (size_t)1 << log_n;
        if (nops > n) throw std::runtime_error("zkm_logic_trace: more operations than 2^log_n rows");
        const uint32_t* d_ops = ops;
        if (nops && !zkm_is_device_ptr(ops)) {
            void* p = c->alloc(nops * 12);
            tmp.push_back(p);
            ZKM_HIP_CHECK(hipMemcpyAsync(p, ops, nops * 12, hipMemcpyHostToDevice, c->stream));
            d_ops = (const uint32_t*)p;
        }
        int* d_bad = (int*)c->alloc(sizeof(int));
        tmp.push_back(d_bad);
        ZKM_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int), c->stream));
        zkm_launch_logic_trace(c, d_ops, nops, n, out_dev, d_bad);
        int bad = 0;
        ZKM_HIP_CHECK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        c->sync();
        for (void* p : tmp) c->release(p);
        tmp.clear();
        if (bad) throw std::runtime_error("zkm_logic_trace: op code out of range (0 and, 1 or, 2 xor, 3 nor)");
    } catch (const std::exception& e) {
        (void)hipStreamSynchronize(c->stream);
        for (void* p : tmp) c->release(p);
        return fail(err, e.what());
    }
    return 0;
}

}  // extern "C"
