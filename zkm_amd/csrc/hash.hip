// hash.hip -- Poseidon / Keccak-f / Merkle kernels for gfx950.
//
// K4  leaf hashing  == plonky2 MerkleTree::new leaf digests = PoseidonHash::hash_or_noop(row)
//     (overwrite-mode sponge, rate 8; rows of <= 4 elements are copied, not hashed -- SURVEY App. A.4)
// K5  inner nodes   == PoseidonHash::two_to_one(left, right), level by level up to the cap
// K15 Keccak-f[1600] batch (reference: cpu/kernel/keccak_util.rs:6-31 via tiny-keccak)
// (witness kernels: witness.hip)
//
// Layout: the LDE is column-major with rows already bit-reversed, so "leaf j" = element j of every
// column: lane = leaf, and each per-column load is a contiguous 512 B wave access -- the transpose
// (K3 in SURVEY §2.1) is fused into the hashing loads.  Integer VALU bound (not HBM, not MFMA).
#include "poseidon_dev.h"
#include "hash_constants_dev.h"
#include "zkm_internal.h"

// ------------------------------------------------------------------ Poseidon permutation batch
__global__ __launch_bounds__(256) void k_poseidon_permute(gl_t* states, size_t k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    uint64_t s[12];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = states[i * 12 + j];
    poseidon_permute(s);
#pragma unroll
    for (int j = 0; j < 12; j++) states[i * 12 + j] = s[j];
}

void zkm_launch_poseidon_permute(zkm_ctx* c, gl_t* states, size_t k) {
    if (!k) return;
    zkm_prof_scope ps(c, "poseidon_permute");
    hipLaunchKernelGGL(k_poseidon_permute, dim3((k + 255) / 256), dim3(256), 0, c->stream, states, k);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ Merkle leaves (column-major rows)
__global__ __launch_bounds__(256) void k_merkle_leaves(const gl_t* __restrict__ lde, size_t nrows, size_t ncols,
                                                       size_t col_stride, gl_t* __restrict__ digests) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nrows) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    if (ncols <= 4) {  // hash_or_noop: copy
        for (size_t c = 0; c < ncols; c++) s[c] = lde[c * col_stride + j];
    } else {
        const gl_t* p = lde + j;
        size_t c = 0;
        for (; c + 8 <= ncols; c += 8) {
            uint64_t v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = p[(c + i) * col_stride];
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = v[i];
            // another whole chunk follows: it replaces words 0..7, only the capacity words are carried over.  Before a ragged chunk
            // (it keeps words rem..7) and at the end (digest) the whole state is needed.
            poseidon_permute_out(s, c + 16 <= ncols ? POSEIDON_OUT_CAPACITY : (c + 8 == ncols ? POSEIDON_OUT_DIGEST : POSEIDON_OUT_ALL));
        }
        if (c < ncols) {
            size_t rem = ncols - c;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if ((size_t)i < rem) s[i] = p[(c + i) * col_stride];
            poseidon_permute_out(s, POSEIDON_OUT_DIGEST);
        }
    }
    uint64_t* d = digests + 4 * j;
    *reinterpret_cast<ulonglong2*>(d) = make_ulonglong2(s[0], s[1]);
    *reinterpret_cast<ulonglong2*>(d + 2) = make_ulonglong2(s[2], s[3]);
}

__global__ void k_merkle_leaves_wide(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* __restrict__ digests);
static size_t wide_max_leaves();

void zkm_launch_merkle_leaves(zkm_ctx* c, const gl_t* lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* digests) {
    // rows of <= 4 elements are copied, not hashed (hash_or_noop): profile them under their own name
    zkm_prof_scope ps(c, ncols <= 4 ? "merkle_leaves_copy" : "merkle_leaves");
    if (ncols > 4 && nrows <= wide_max_leaves())
        // short, wide matrices (Keccak: 2431 columns x a few thousand rows): one lane per leaf leaves the machine empty and pays one
        // permutation's full latency per 8 columns; 16 lanes per leaf cut that latency to a third and fill 16x the lanes
        hipLaunchKernelGGL(k_merkle_leaves_wide, dim3((nrows * 16 + 255) / 256), dim3(256), 0, c->stream, lde, nrows, ncols, col_stride, digests);
    else
        hipLaunchKernelGGL(k_merkle_leaves, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, lde, nrows, ncols, col_stride, digests);
    ZKM_HIP_CHECK(hipGetLastError());
}

// Leaf hashing in column chunks (pipelined host ingest, core.hip zkm_batch_build): the sponge absorbs eight columns per
// permutation in column order, so a chunk of columns [c0, c0 + nc) -- c0 a multiple of 8 -- can be absorbed as soon as its LDE
// exists, with the 12-word sponge state of every row parked in HBM between chunks (state[i * nrows + j], lane-contiguous).
// first: the state starts at zero; last: the digest is written instead of the state.  Same permutation sequence as
// k_merkle_leaves, so digests are identical.
__global__ __launch_bounds__(256) void k_merkle_leaves_chunk(const gl_t* __restrict__ lde, size_t nrows, size_t nc, size_t col_stride,
                                                             gl_t* __restrict__ state, int first, int last, gl_t* __restrict__ digests) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nrows) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = first ? 0 : state[(size_t)i * nrows + j];
    const gl_t* p = lde + j;
    size_t c = 0;
    for (; c + 8 <= nc; c += 8) {
        uint64_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = p[(c + i) * col_stride];
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = v[i];
        // (the state parked between chunks is the whole, canonical state: the last permutation of a launch keeps all twelve words)
        poseidon_permute_out(s, c + 16 <= nc ? POSEIDON_OUT_CAPACITY : POSEIDON_OUT_ALL);
    }
    if (c < nc) {  // ragged tail: only legal in the last chunk
        size_t rem = nc - c;
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((size_t)i < rem) s[i] = p[(c + i) * col_stride];
        poseidon_permute_out(s, POSEIDON_OUT_ALL);
    }
    if (last) {
        uint64_t* d = digests + 4 * j;
        *reinterpret_cast<ulonglong2*>(d) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(d + 2) = make_ulonglong2(s[2], s[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 12; i++) state[(size_t)i * nrows + j] = s[i];
    }
}

void zkm_launch_merkle_leaves_chunk(zkm_ctx* c, const gl_t* lde, size_t nrows, size_t nc, size_t col_stride, gl_t* state, bool first,
                                    bool last, gl_t* digests) {
    if (!last && nc % 8) throw std::runtime_error("merkle_leaves_chunk: only the last chunk may hold a ragged group of columns");
    zkm_prof_scope ps(c, "merkle_leaves");
    hipLaunchKernelGGL(k_merkle_leaves_chunk, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, lde, nrows, nc, col_stride, state,
                       first ? 1 : 0, last ? 1 : 0, digests);
    ZKM_HIP_CHECK(hipGetLastError());
}

// leaves of a FRI layer: leaf k = arity consecutive F2 values (bit-reversed order), flattened c0,c1,c0,c1..
__global__ __launch_bounds__(256) void k_merkle_leaves_ext(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1,
                                                           size_t nleaves, unsigned arity, gl_t* __restrict__ digests) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nleaves) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const gl_t* a = c0 + k * arity;
    const gl_t* b = c1 + k * arity;
    // 2*arity words, arity a multiple of 4 -> whole chunks of 8 words = 4 pairs
    for (unsigned e = 0; e < arity; e += 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            s[2 * i] = a[e + i];
            s[2 * i + 1] = b[e + i];
        }
        poseidon_permute_out(s, e + 4 < arity ? POSEIDON_OUT_CAPACITY : POSEIDON_OUT_DIGEST);
    }
    uint64_t* d = digests + 4 * k;
    *reinterpret_cast<ulonglong2*>(d) = make_ulonglong2(s[0], s[1]);
    *reinterpret_cast<ulonglong2*>(d + 2) = make_ulonglong2(s[2], s[3]);
}

// ------------------------------------------------------------------ Merkle inner level
__global__ __launch_bounds__(256) void k_merkle_compress(const gl_t* __restrict__ children, gl_t* __restrict__ parents,
                                                         size_t nparents) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nparents) return;
    const ulonglong2* ch = reinterpret_cast<const ulonglong2*>(children + 8 * i);
    ulonglong2 a = ch[0], b = ch[1], cc = ch[2], d = ch[3];
    uint64_t s[12] = {a.x, a.y, b.x, b.y, cc.x, cc.y, d.x, d.y, 0, 0, 0, 0};
    poseidon_permute_out(s, POSEIDON_OUT_DIGEST);
    uint64_t* o = parents + 4 * i;
    *reinterpret_cast<ulonglong2*>(o) = make_ulonglong2(s[0], s[1]);
    *reinterpret_cast<ulonglong2*>(o + 2) = make_ulonglong2(s[2], s[3]);
}

// ---- one permutation across 12 lanes (small tree levels) ----
// The top levels of every tree hold too few nodes to fill the machine, so a launch costs one permutation's LATENCY (~35 us for
// the one-lane-per-hash form: ~18k dependent-ish instructions).  Here a hash owns a 16-lane row of the wave (lanes 0..11 = the
// twelve state words): every round is constant add, x^7 (lane 0 only in the partial rounds), and the circulant MDS with the
// twelve rotated neighbours fetched by ds_bpermute -- the textbook rounds (poseidon_stark.rs:65-95, 164-169, 239-251, 310-345),
// 30 x ~190 instructions per lane instead of ~18k, i.e. a quarter of the latency at ~5x the total work.  Used when a level has
// at most wide_max_parents() nodes (16384; ZKM_WIDE_MAX tunes it).  Bit-exact with poseidon_permute (the fused partial rounds are an algebraic regrouping).
__device__ __forceinline__ uint64_t poseidon_permute_wide(uint64_t x, unsigned lane) {
    const unsigned idx = lane & 15, base = lane & ~15u;
    const bool active = idx < 12;
    int src[12];
#pragma unroll
    for (int i = 1; i < 12; i++) src[i] = (int)(base + (idx + i) % 12);
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t diag = idx == 0 ? 8u : 0u;
    const gl_t* rcp = PC::ZKM_POSEIDON_RC + (active ? idx : 0);
    x = gl_add_loose(x, rcp[0]);
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        const uint64_t y = poseidon_sbox7(x);
        x = (full || idx == 0) ? y : x;
        const uint64_t k = r + 1 < 30 ? rcp[(r + 1) * 12] : 0;
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        uint64_t al = (uint64_t)(uint32_t)k + (uint64_t)lo * (C[0] + diag), ah = (k >> 32) + (uint64_t)hi * (C[0] + diag);
#pragma unroll
        for (int i = 1; i < 12; i++) {
            al += (uint64_t)(uint32_t)__shfl((int)lo, src[i]) * C[i];
            ah += (uint64_t)(uint32_t)__shfl((int)hi, src[i]) * C[i];
        }
        x = poseidon_fold(al, ah);
    }
    return gl_canon(x);
}

__global__ __launch_bounds__(256) void k_merkle_compress_wide(const gl_t* __restrict__ children, gl_t* __restrict__ parents, size_t nparents) {
    const unsigned lane = threadIdx.x & 63, idx = lane & 15;
    const size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = p < nparents;  // uniform over the 16-lane row; every lane of the wave takes part in the shuffles
    uint64_t x = 0;
    if (live && idx < 8) x = children[8 * p + idx];
    x = poseidon_permute_wide(x, lane);
    if (live && idx < 4) parents[4 * p + idx] = x;
}

// Column-major leaves, one leaf per 16-lane row (see poseidon_permute_wide): lanes 0..7 of the row fetch the next eight columns of
// their leaf (overwrite-mode absorb: a ragged tail overwrites only the words that exist), all 12 lanes permute.  Bit-exact with
// k_merkle_leaves.  Used when the matrix has at most wide_max_leaves() rows.
__global__ __launch_bounds__(256) void k_merkle_leaves_wide(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride,
                                                            gl_t* __restrict__ digests) {
    const unsigned lane = threadIdx.x & 63, idx = lane & 15;
    const size_t leaf = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = leaf < nrows;  // uniform over the 16-lane row; every lane of the wave takes part in the shuffles
    uint64_t x = 0;
    for (size_t c = 0; c < ncols; c += 8) {
        if (live && idx < 8 && c + idx < ncols) x = lde[(c + idx) * col_stride + leaf];
        x = poseidon_permute_wide(x, lane);
    }
    if (live && idx < 4) digests[4 * leaf + idx] = x;
}
static size_t wide_max_leaves() {
    static size_t v = [] { const char* e = getenv("ZKM_WIDE_MAX_LEAVES"); return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)16384; }();
    return v;
}

static size_t wide_max_parents() {
    static size_t v = [] { const char* e = getenv("ZKM_WIDE_MAX"); return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)16384; }();
    return v;
}

// FRI layer leaves, one hash per 16-lane row (small layers): word m of leaf k is component m & 1 of value k * arity + (m >> 1).
__global__ __launch_bounds__(256) void k_merkle_leaves_ext_wide(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1, size_t nleaves,
                                                                unsigned arity, gl_t* __restrict__ digests) {
    const unsigned lane = threadIdx.x & 63, idx = lane & 15;
    const size_t k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = k < nleaves;
    const gl_t* col = (idx & 1) ? c1 : c0;
    uint64_t x = 0;
    for (unsigned m = 0; m < 2 * arity; m += 8) {
        if (live && idx < 8) x = col[k * arity + ((m + idx) >> 1)];
        x = poseidon_permute_wide(x, lane);
    }
    if (live && idx < 4) digests[4 * k + idx] = x;
}

void zkm_launch_merkle_leaves_ext(zkm_ctx* c, const gl_t* c0, const gl_t* c1, size_t nleaves, unsigned arity, gl_t* digests) {
    if (arity % 4 || 2 * arity <= 4) throw std::runtime_error("merkle_leaves_ext: unsupported arity");
    zkm_prof_scope ps(c, "merkle_leaves_ext");
    if (nleaves <= wide_max_parents() / 4)
        hipLaunchKernelGGL(k_merkle_leaves_ext_wide, dim3((nleaves * 16 + 255) / 256), dim3(256), 0, c->stream, c0, c1, nleaves, arity, digests);
    else
        hipLaunchKernelGGL(k_merkle_leaves_ext, dim3((nleaves + 255) / 256), dim3(256), 0, c->stream, c0, c1, nleaves, arity, digests);
    ZKM_HIP_CHECK(hipGetLastError());
}

void zkm_launch_merkle_compress(zkm_ctx* c, const gl_t* children, gl_t* parents, size_t nparents) {
    zkm_prof_scope ps(c, "merkle_compress");
    if (nparents <= wide_max_parents())
        hipLaunchKernelGGL(k_merkle_compress_wide, dim3((nparents * 16 + 255) / 256), dim3(256), 0, c->stream, children, parents, nparents);
    else
        hipLaunchKernelGGL(k_merkle_compress, dim3((nparents + 255) / 256), dim3(256), 0, c->stream, children, parents, nparents);
    ZKM_HIP_CHECK(hipGetLastError());
}

size_t zkm_merkle_layout(unsigned log_leaves, unsigned cap_height, std::vector<size_t>& level_off) {
    if (cap_height > log_leaves) throw std::runtime_error("cap_height exceeds tree height");
    level_off.clear();
    size_t total = 0;
    for (unsigned l = 0; l + cap_height <= log_leaves; l++) {
        level_off.push_back(total);
        total += ((size_t)1 << (log_leaves - l)) * 4;
    }
    return total;
}

void zkm_merkle_build_inner(zkm_ctx* c, gl_t* digests, const std::vector<size_t>& level_off, unsigned log_leaves,
                            unsigned cap_height) {
    unsigned top = log_leaves - cap_height;
    for (unsigned l = 1; l <= top; l++)
        zkm_launch_merkle_compress(c, digests + level_off[l - 1], digests + level_off[l], (size_t)1 << (log_leaves - l));
}

// ------------------------------------------------------------------ Keccak-f[1600]
__global__ __launch_bounds__(256) void k_keccakf(uint64_t* states, size_t k) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= k) return;
    uint64_t a[25];
    uint64_t* st = states + idx * 25;
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = st[i];
    keccakf_dev(a);
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = a[i];
}

void zkm_launch_keccakf(zkm_ctx* c, uint64_t* states, size_t k) {
    if (!k) return;
    zkm_prof_scope ps(c, "keccakf");
    hipLaunchKernelGGL(k_keccakf, dim3((k + 255) / 256), dim3(256), 0, c->stream, states, k);
    ZKM_HIP_CHECK(hipGetLastError());
}
