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
#include <atomic>
#include "poseidon_dev.h"
#include "poseidon_lat_dev.h"
#include "poseidon_mfma_dev.h"
#include "hash_constants_dev.h"
#include "zkm_internal.h"

// five waves per SIMD (93 registers) with the layer's high-half sums parked in LDS, four (110 registers) without (poseidon_mfma_dev.h)
#ifndef ZKM_LEAF_MFMA_WAVES
#define ZKM_LEAF_MFMA_WAVES (ZKM_MFMA_PARK ? 5 : 4)
#endif

// The latency forms of the permutation (16 lanes or a quad per hash) exist for launches that are a CHAIN of dependent permutations
// with too few hashes to fill the machine: the top of every tree, the leaves of short tables, small FRI layers.  When such a launch
// shares the GPU with throughput kernels (the commit lanes of the same segment, other contexts), its few waves compete for issue
// slots with five resident waves per SIMD of somebody's leaf hashing, and every step of the chain waits its turn.  s_setprio raises
// the wave's priority at the SIMD's instruction arbiter: the chain runs at (nearly) its solo latency and the throughput kernel
// fills what is left.  ZKM_LATENCY_PRIO = 0 builds without it (A/B: profiles/r04_latency_prio.txt).
#ifndef ZKM_LATENCY_PRIO
#define ZKM_LATENCY_PRIO 3
#endif
#define ZKM_RAISE_PRIO()                                              \
    do {                                                              \
        if (ZKM_LATENCY_PRIO) __builtin_amdgcn_s_setprio(ZKM_LATENCY_PRIO); \
    } while (0)

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
// (every leaf / tree kernel: blockIdx.z = segment of a stacked batch, zkm_internal.h -- its matrix starts lde_seg words, its digest
// block dig_seg words after the previous segment's)
// MFMA = true: the MDS layers of the full rounds run on the matrix core (poseidon_mfma_dev.h).  MFMA ignores EXEC, so that form keeps
// the lanes past the last row alive on the last row's data and only masks their store.
template <bool MFMA>
__device__ __forceinline__ void merkle_leaves_body(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* __restrict__ digests,
                                                   size_t lde_seg, size_t dig_seg) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j < nrows;
    if (!MFMA && !live) return;
    if (!live) j = nrows - 1;
    lde += (size_t)blockIdx.z * lde_seg;
    digests += (size_t)blockIdx.z * dig_seg;
    typename std::conditional<MFMA, poseidon_mds_mfma, poseidon_mds_valu>::type mds;
    if constexpr (MFMA) {
        mds.A = poseidon_mfma_operand();
#if ZKM_MFMA_PARK
        __shared__ uint32_t park[24 * ZKM_MFMA_PARK_STRIDE];
        mds.park = park + threadIdx.x;
#endif
    }
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    if (ncols <= 4) {  // hash_or_noop: copy
        for (size_t c = 0; c < ncols; c++) s[c] = lde[c * col_stride + j];
    } else {
        const gl_t* p = lde + j;
        size_t c = 0;
        // ONE copy of the round code: the ragged last chunk (it overwrites the words that exist) goes through the same call site.  (Round 5's
        // 126-register layer needed a second, multiply-add copy for it; at 93 registers it does not, and two copies buy nothing: 39.25
        // against 39.33 ms, measured.)  Another whole chunk follows: it replaces words 0..7 and only the capacity words are carried over;
        // before a ragged chunk (it keeps words rem..7) the whole state is needed, at the end the digest.
        for (; c < ncols; c += 8) {
            const size_t rem = ncols - c;          // (uniform)
            if (rem >= 8) {
                uint64_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = p[(c + i) * col_stride];
#pragma unroll
                for (int i = 0; i < 8; i++) s[i] = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if ((size_t)i < rem) s[i] = p[(c + i) * col_stride];
            }
            poseidon_permute_out_t(s, c + 16 <= ncols ? POSEIDON_OUT_CAPACITY : (c + 8 >= ncols ? POSEIDON_OUT_DIGEST : POSEIDON_OUT_ALL), mds);
        }
    }
    if (!live) return;
    uint64_t* d = digests + 4 * j;
    *reinterpret_cast<ulonglong2*>(d) = make_ulonglong2(s[0], s[1]);
    *reinterpret_cast<ulonglong2*>(d + 2) = make_ulonglong2(s[2], s[3]);
}

__global__ __launch_bounds__(256) void k_merkle_leaves(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride,
                                                       gl_t* __restrict__ digests, size_t lde_seg, size_t dig_seg) {
    merkle_leaves_body<false>(lde, nrows, ncols, col_stride, digests, lde_seg, dig_seg);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ZKM_LEAF_MFMA_WAVES, ZKM_LEAF_MFMA_WAVES)))
void k_merkle_leaves_mfma(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* __restrict__ digests, size_t lde_seg,
                          size_t dig_seg) {
    merkle_leaves_body<true>(lde, nrows, ncols, col_stride, digests, lde_seg, dig_seg);
}

__global__ void k_merkle_leaves_quad(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* __restrict__ digests,
                                     size_t lde_seg, size_t dig_seg);
__global__ void k_merkle_leaves_wide(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* __restrict__ digests,
                                     size_t lde_seg, size_t dig_seg);

void zkm_launch_merkle_leaves(zkm_ctx* c, const gl_t* lde, size_t nrows, size_t ncols, size_t col_stride, gl_t* digests, size_t nseg,
                              size_t lde_seg, size_t dig_seg) {
    // rows of <= 4 elements are copied, not hashed (hash_or_noop): profile them under their own name
    zkm_prof_scope ps(c, ncols <= 4 ? "merkle_leaves_copy" : "merkle_leaves");
    if (nseg == 0 || nseg > 65535) throw std::runtime_error("merkle_leaves: bad segment count");
    const size_t hashes = nrows * nseg;    // the launch's hashes decide the form of the permutation: the segments of a stacked batch fill the machine together
    const unsigned z = (unsigned)nseg;
    if (ncols > 4 && hashes <= c->wide_max_hashes)
        // the shortest matrices: 16 lanes per leaf, the lowest latency per absorb step
        hipLaunchKernelGGL(k_merkle_leaves_wide, dim3((nrows * 16 + 255) / 256, 1, z), dim3(256), 0, c->stream, lde, nrows, ncols, col_stride, digests,
                           lde_seg, dig_seg);
    else if (ncols > 4 && hashes <= c->quad_max_hashes)
        // short, wide matrices (Keccak: 2431 columns x a few thousand rows): one lane per leaf leaves the machine empty and pays one
        // permutation's full latency per 8 columns; four lanes per leaf cut that latency to a third and fill 4x the lanes
        hipLaunchKernelGGL(k_merkle_leaves_quad, dim3((nrows * 4 + 255) / 256, 1, z), dim3(256), 0, c->stream, lde, nrows, ncols, col_stride, digests,
                           lde_seg, dig_seg);
    else if (c->leaf_mfma && ncols > 4)
        hipLaunchKernelGGL(k_merkle_leaves_mfma, dim3((nrows + 255) / 256, 1, z), dim3(256), 0, c->stream, lde, nrows, ncols, col_stride, digests,
                           lde_seg, dig_seg);
    else
        hipLaunchKernelGGL(k_merkle_leaves, dim3((nrows + 255) / 256, 1, z), dim3(256), 0, c->stream, lde, nrows, ncols, col_stride, digests,
                           lde_seg, dig_seg);
    ZKM_HIP_CHECK(hipGetLastError());
}

// Leaf hashing in column chunks (pipelined host ingest, core.hip zkm_batch_build): the sponge absorbs eight columns per
// permutation in column order, so a chunk of columns [c0, c0 + nc) -- c0 a multiple of 8 -- can be absorbed as soon as its LDE
// exists, with the 12-word sponge state of every row parked in HBM between chunks (state[i * nrows + j], lane-contiguous).
// first: the state starts at zero; last: the digest is written instead of the state.  Same permutation sequence as
// k_merkle_leaves, so digests are identical.
template <bool MFMA>
__device__ __forceinline__ void merkle_leaves_chunk_body(const gl_t* __restrict__ lde, size_t nrows, size_t nc, size_t col_stride,
                                                         gl_t* __restrict__ state, int first, int last, gl_t* __restrict__ digests) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j < nrows;
    if (!MFMA && !live) return;
    if (!live) j = nrows - 1;          // (MFMA ignores EXEC: lanes past the last row hash the last row and store nothing)
    typename std::conditional<MFMA, poseidon_mds_mfma, poseidon_mds_valu>::type mds;
    if constexpr (MFMA) {
        mds.A = poseidon_mfma_operand();
#if ZKM_MFMA_PARK
        __shared__ uint32_t park[24 * ZKM_MFMA_PARK_STRIDE];
        mds.park = park + threadIdx.x;
#endif
    }
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
        poseidon_permute_out_t(s, c + 16 <= nc ? POSEIDON_OUT_CAPACITY : POSEIDON_OUT_ALL, mds);
    }
    if (c < nc) {  // ragged tail: only legal in the last chunk
        size_t rem = nc - c;
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((size_t)i < rem) s[i] = p[(c + i) * col_stride];
        poseidon_permute_out_t(s, POSEIDON_OUT_ALL, poseidon_mds_valu{});   // (the ragged chunk's one permutation: the second copy of the round code stays on the multiply-add form, as in k_merkle_leaves_mfma)
    }
    if (!live) return;
    if (last) {
        uint64_t* d = digests + 4 * j;
        *reinterpret_cast<ulonglong2*>(d) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(d + 2) = make_ulonglong2(s[2], s[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 12; i++) state[(size_t)i * nrows + j] = s[i];
    }
}
__global__ __launch_bounds__(256) void k_merkle_leaves_chunk(const gl_t* __restrict__ lde, size_t nrows, size_t nc, size_t col_stride,
                                                             gl_t* __restrict__ state, int first, int last, gl_t* __restrict__ digests) {
    merkle_leaves_chunk_body<false>(lde, nrows, nc, col_stride, state, first, last, digests);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ZKM_LEAF_MFMA_WAVES, ZKM_LEAF_MFMA_WAVES)))
void k_merkle_leaves_chunk_mfma(const gl_t* __restrict__ lde, size_t nrows, size_t nc, size_t col_stride, gl_t* __restrict__ state, int first, int last,
                                gl_t* __restrict__ digests) {
    merkle_leaves_chunk_body<true>(lde, nrows, nc, col_stride, state, first, last, digests);
}

void zkm_launch_merkle_leaves_chunk(zkm_ctx* c, const gl_t* lde, size_t nrows, size_t nc, size_t col_stride, gl_t* state, bool first,
                                    bool last, gl_t* digests) {
    if (!last && nc % 8) throw std::runtime_error("merkle_leaves_chunk: only the last chunk may hold a ragged group of columns");
    zkm_prof_scope ps(c, "merkle_leaves");
    if (c->leaf_mfma)
        hipLaunchKernelGGL(k_merkle_leaves_chunk_mfma, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, lde, nrows, nc, col_stride, state,
                           first ? 1 : 0, last ? 1 : 0, digests);
    else
        hipLaunchKernelGGL(k_merkle_leaves_chunk, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, lde, nrows, nc, col_stride, state,
                           first ? 1 : 0, last ? 1 : 0, digests);
    ZKM_HIP_CHECK(hipGetLastError());
}

// leaves of a FRI layer: leaf k = arity consecutive F2 values (bit-reversed order), flattened c0,c1,c0,c1..
// (MFMA: the matrix-core MDS layer of the leaf kernel -- MFMA ignores EXEC: lanes past the last leaf hash the last leaf and do not store)
template <bool MFMA>
__device__ __forceinline__ void merkle_leaves_ext_body(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1, size_t nleaves, unsigned arity,
                                                       gl_t* __restrict__ digests, size_t val_seg, size_t dig_seg) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = k < nleaves;
    if (!MFMA && !live) return;
    if (!live) k = nleaves - 1;
    c0 += (size_t)blockIdx.z * val_seg; c1 += (size_t)blockIdx.z * val_seg; digests += (size_t)blockIdx.z * dig_seg;
    typename std::conditional<MFMA, poseidon_mds_mfma, poseidon_mds_valu>::type mds;
    if constexpr (MFMA) {
        mds.A = poseidon_mfma_operand();
#if ZKM_MFMA_PARK
        __shared__ uint32_t park[24 * ZKM_MFMA_PARK_STRIDE];
        mds.park = park + threadIdx.x;
#endif
    }
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
        poseidon_permute_out_t(s, e + 4 < arity ? POSEIDON_OUT_CAPACITY : POSEIDON_OUT_DIGEST, mds);
    }
    if (!live) return;
    uint64_t* d = digests + 4 * k;
    *reinterpret_cast<ulonglong2*>(d) = make_ulonglong2(s[0], s[1]);
    *reinterpret_cast<ulonglong2*>(d + 2) = make_ulonglong2(s[2], s[3]);
}
__global__ __launch_bounds__(256) void k_merkle_leaves_ext(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1,
                                                           size_t nleaves, unsigned arity, gl_t* __restrict__ digests, size_t val_seg, size_t dig_seg) {
    merkle_leaves_ext_body<false>(c0, c1, nleaves, arity, digests, val_seg, dig_seg);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ZKM_LEAF_MFMA_WAVES, ZKM_LEAF_MFMA_WAVES)))
void k_merkle_leaves_ext_mfma(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1, size_t nleaves, unsigned arity, gl_t* __restrict__ digests,
                              size_t val_seg, size_t dig_seg) {
    merkle_leaves_ext_body<true>(c0, c1, nleaves, arity, digests, val_seg, dig_seg);
}

// ------------------------------------------------------------------ Merkle inner levels, fused (K5)
// plonky2 builds a tree in one call (MerkleTree::new, call sites prover.rs:154-163); here a launch takes up to THREE levels.
// Large levels (first level of the launch >= 2^15 parents): one hash per lane.  A workgroup of 256 threads owns 512 children of
// level l and produces their 256 parents, then -- from LDS, no trip through HBM -- the 128 and the 64 nodes above (whole waves
// stay active: 4, 2, 1, and the waves that are done leave at once).  Every level is also stored (Merkle paths read all of them).  The permutation code exists once (the level
// loop is not unrolled), so the kernel stays within the instruction cache like k_merkle_leaves.
struct merkle_fused_args {
    const gl_t* children;   // level l: 2 * nparents digests
    gl_t* parents[3];       // levels l + 1 .. l + 3
    uint32_t levels;        // 1 .. 3
    size_t dig_seg;         // blockIdx.z-th tree: every pointer + blockIdx.z * dig_seg
};

__global__ __launch_bounds__(256) void k_merkle_fused(merkle_fused_args p) {
    __shared__ uint64_t sh[2][4][256];                    // [buffer][word][node]: parents of the level just made, word-major
    const unsigned tid = threadIdx.x;
    uint64_t s[12];
    {
        const ulonglong2* ch = reinterpret_cast<const ulonglong2*>(p.children + (size_t)blockIdx.z * p.dig_seg + 8 * ((size_t)blockIdx.x * 256 + tid));
        ulonglong2 a = ch[0], b = ch[1], cc = ch[2], d = ch[3];
        s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = cc.x; s[5] = cc.y; s[6] = d.x; s[7] = d.y;
    }
#pragma unroll 1
    for (unsigned lvl = 0; lvl < p.levels; lvl++) {
        const unsigned width = 256u >> lvl;               // nodes of this level in the workgroup: 256, 128, 64 == the surviving threads
        if (lvl) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                s[w] = sh[(lvl - 1) & 1][w][2 * tid];
                s[4 + w] = sh[(lvl - 1) & 1][w][2 * tid + 1];
            }
        }
#pragma unroll
        for (int w = 8; w < 12; w++) s[w] = 0;
        poseidon_permute_out(s, POSEIDON_OUT_DIGEST);
        uint64_t* o = p.parents[lvl] + (size_t)blockIdx.z * p.dig_seg + 4 * ((size_t)blockIdx.x * width + tid);
        *reinterpret_cast<ulonglong2*>(o) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(o + 2) = make_ulonglong2(s[2], s[3]);
        if (lvl + 1 < p.levels) {
#pragma unroll
            for (int w = 0; w < 4; w++) sh[lvl & 1][w][tid] = s[w];
        }
        if (lvl + 1 == p.levels) return;
        __syncthreads();
        // Waves without a node in the next level leave NOW and free their slots for the next workgroup (s_barrier only waits for the
        // surviving waves of a workgroup): parked at the barriers they would hold a third of the CU's wave slots idle, and this
        // kernel needs every slot it can get (one permutation per lane is ~12k dependent-ish instructions).
        if (tid >= (width >> 1)) return;
    }
}
// Small levels, fused: a workgroup (256 threads = 16 hash slots of 16 lanes) owns a subtree with 2^J children (J <= 6) and climbs
// its J levels through LDS -- ceil(2^(J-k) / 16) rounds of wide permutations at level k -- instead of one launch per level: the top
// of a tree is a chain of dependent permutations (one wide permutation ~10 us), and every launch boundary added its gap to it.
struct merkle_fused_wide_args {
    const gl_t* children;   // level l: nsub * 2^J digests
    gl_t* parents[6];       // levels l + 1 .. l + J
    uint32_t J;             // levels in this launch (1 .. 6)
    size_t dig_seg;
};

__global__ __launch_bounds__(256) void k_merkle_fused_wide(merkle_fused_wide_args p) {
    ZKM_RAISE_PRIO();
    __shared__ uint64_t sh[2][64 * 4];                    // digests of the current level of this subtree (AoS, as in HBM)
    const unsigned tid = threadIdx.x, lane = tid & 63, idx = lane & 15, slot = tid >> 4;
    const unsigned C = 1u << p.J;
    const size_t seg_off = (size_t)blockIdx.z * p.dig_seg;
    {
        const gl_t* src = p.children + seg_off + (size_t)blockIdx.x * C * 4;
        for (unsigned w = tid; w < C * 4; w += 256) sh[0][w] = src[w];
    }
    __syncthreads();
    for (unsigned lvl = 0; lvl < p.J; lvl++) {
        const unsigned np = C >> (lvl + 1);               // parents of this level in the subtree
        const uint64_t* in = sh[lvl & 1];
        uint64_t* out = sh[(lvl + 1) & 1];
        gl_t* g = p.parents[lvl] + seg_off + (size_t)blockIdx.x * np * 4;
        if ((slot & ~3u) >= np) return;                  // this wave has no node here or above: leave (the barrier only counts survivors)
        for (unsigned h0 = 0; h0 < np; h0 += 16) {
            // (a wave holds four slots; one with no live slot in this round skips it -- uniform over the wave)
            if (h0 + (slot & ~3u) >= np) continue;
            const unsigned h = h0 + slot;
            const bool live = h < np;                     // uniform over the 16-lane row; every lane of the wave shuffles
            uint64_t x = (live && idx < 8) ? in[8 * h + idx] : 0;
            x = poseidon_permute_wide(x, lane);
            if (live && idx < 4) {
                out[4 * h + idx] = x;
                g[4 * h + idx] = x;
            }
        }
        __syncthreads();
    }
}

// Column-major leaves, one leaf per 16-lane row (see poseidon_permute_wide): lanes 0..7 of the row fetch the next eight columns of
// their leaf (overwrite-mode absorb: a ragged tail overwrites only the words that exist), all 12 lanes permute.  Bit-exact with
// k_merkle_leaves.  Used when the matrix has at most zkm_ctx::wide_max_hashes rows.
__global__ __launch_bounds__(256) void k_merkle_leaves_wide(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride,
                                                            gl_t* __restrict__ digests, size_t lde_seg, size_t dig_seg) {
    ZKM_RAISE_PRIO();
    lde += (size_t)blockIdx.z * lde_seg;
    digests += (size_t)blockIdx.z * dig_seg;
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
// Rows up to which a leaf gets a 16-lane row (k_merkle_leaves_wide).  One absorb step takes ~13 us in the 16-lane form while every SIMD
// holds at most one such wave (4096 rows), ~21 us with two (8192), ~41 us with four (16384); the four-lane form takes over there.
// zkm_ctx::wide_max_hashes = 1024: up to 4096 the 16-lane form is the fastest for a context alone, but it spends 4.2x the issue slots
// of the one-lane form, and with four or more contexts on the GPU those slots are somebody else's work (2^16-cycle segments, 1 / 4 / 8
// contexts: 24.5 / 52.5 / 51.8 segments/s at 4096, 25.5 / 56.9 / 59.0 at 1024 -- profiles/r03_hw_queues.txt).


// FRI layer leaves, one hash per 16-lane row (small layers): word m of leaf k is component m & 1 of value k * arity + (m >> 1).
__global__ __launch_bounds__(256) void k_merkle_leaves_ext_wide(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1, size_t nleaves,
                                                                unsigned arity, gl_t* __restrict__ digests, size_t val_seg, size_t dig_seg) {
    ZKM_RAISE_PRIO();
    c0 += (size_t)blockIdx.z * val_seg; c1 += (size_t)blockIdx.z * val_seg; digests += (size_t)blockIdx.z * dig_seg;
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

// Small levels, fused: a workgroup (256 threads = 64 hash slots of 4 lanes) owns a subtree with 2^J children (J <= 7) and climbs its J
// levels through LDS -- ceil(2^(J-k-1) / 64) rounds of quad permutations at level k -- instead of one launch per level: the top of a
// tree is a chain of dependent permutations, and every launch boundary added its gap to it.
struct merkle_fused_quad_args {
    const gl_t* children;   // level l: nsub * 2^J digests
    gl_t* parents[7];       // levels l + 1 .. l + J
    uint32_t J;             // levels in this launch (1 .. 7)
    size_t dig_seg;
};

__global__ __launch_bounds__(256) void k_merkle_fused_quad(merkle_fused_quad_args p) {
    ZKM_RAISE_PRIO();
    __shared__ uint64_t sh[2][128 * 4];                   // digests of the current level of this subtree (AoS, as in HBM)
    const unsigned tid = threadIdx.x, q = tid & 3, slot = tid >> 2, wave_slot0 = (tid >> 6) << 4;
    __shared__ __attribute__((aligned(16))) uint32_t qtab[ZKM_QUAD_TAB_WORDS];
    quad_tab_load(qtab);
    const poseidon_quad Q(tid, qtab);
    const unsigned C = 1u << p.J;
    const size_t seg_off = (size_t)blockIdx.z * p.dig_seg;
    {
        const gl_t* src = p.children + seg_off + (size_t)blockIdx.x * C * 4;
        for (unsigned w = tid; w < C * 4; w += 256) sh[0][w] = src[w];
    }
    __syncthreads();
    for (unsigned lvl = 0; lvl < p.J; lvl++) {
        const unsigned np = C >> (lvl + 1);               // parents of this level in the subtree
        const uint64_t* in = sh[lvl & 1];
        uint64_t* out = sh[(lvl + 1) & 1];
        gl_t* g = p.parents[lvl] + seg_off + (size_t)blockIdx.x * np * 4;
        if (wave_slot0 >= np) return;                     // this wave has no node here or above: leave (the barrier only counts survivors)
        for (unsigned h0 = 0; h0 < np; h0 += 64) {
            if (h0 + wave_slot0 >= np) continue;          // (a wave holds sixteen slots; uniform over the wave)
            const unsigned h = h0 + slot;
            const bool live = h < np;                     // uniform over the quad; every lane of the wave takes part in the DPP moves
            uint64_t s[3] = {live ? in[8 * h + q] : 0, live ? in[8 * h + 4 + q] : 0, 0};
            poseidon_permute_quad(s, Q);
            if (live) {
                out[4 * h + q] = s[0];
                g[4 * h + q] = s[0];
            }
        }
        __syncthreads();
    }
}

// Column-major leaves, one leaf per quad: lane q fetches columns c + q and c + 4 + q of its leaf for the absorb step at column c
// (overwrite-mode absorb: a ragged tail overwrites only the words that exist).  Bit-exact with k_merkle_leaves.  Used when the matrix
// has at most zkm_ctx::quad_max_hashes rows.
__global__ __launch_bounds__(256) void k_merkle_leaves_quad(const gl_t* __restrict__ lde, size_t nrows, size_t ncols, size_t col_stride,
                                                            gl_t* __restrict__ digests, size_t lde_seg, size_t dig_seg) {
    ZKM_RAISE_PRIO();
    lde += (size_t)blockIdx.z * lde_seg;
    digests += (size_t)blockIdx.z * dig_seg;
    const unsigned q = threadIdx.x & 3;
    const size_t leaf = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const bool live = leaf < nrows;  // uniform over the quad; every lane of the wave takes part in the DPP moves
    __shared__ __attribute__((aligned(16))) uint32_t qtab[ZKM_QUAD_TAB_WORDS];
    quad_tab_load(qtab);
    const poseidon_quad Q(threadIdx.x, qtab);
    uint64_t s[3] = {0, 0, 0};
    for (size_t c = 0; c < ncols; c += 8) {
        if (live && c + q < ncols) s[0] = lde[(c + q) * col_stride + leaf];
        if (live && c + 4 + q < ncols) s[1] = lde[(c + 4 + q) * col_stride + leaf];
        poseidon_permute_quad(s, Q);
    }
    if (live) digests[4 * leaf + q] = s[0];
}
// Rows up to which a leaf gets a quad of lanes.  One absorb step takes ~16 us in the quad form while a SIMD holds at most one such wave
// (16384 rows = 1024 waves; ~21 us with two, 32768 rows), against ~39 us for the one-lane form on a wave that has its SIMD to itself;
// it costs 1.45x the instructions per hash since its partial rounds are fused (poseidon_lat_dev.h; 1.9x and 24 us before that).
// (zkm_ctx::quad_max_hashes = 32768)

// FRI layer leaves, one hash per quad (small layers): word m of leaf k is component m & 1 of value k * arity + (m >> 1).
__global__ __launch_bounds__(256) void k_merkle_leaves_ext_quad(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1, size_t nleaves,
                                                                unsigned arity, gl_t* __restrict__ digests, size_t val_seg, size_t dig_seg) {
    ZKM_RAISE_PRIO();
    c0 += (size_t)blockIdx.z * val_seg; c1 += (size_t)blockIdx.z * val_seg; digests += (size_t)blockIdx.z * dig_seg;
    const unsigned q = threadIdx.x & 3;
    const size_t k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const bool live = k < nleaves;
    __shared__ __attribute__((aligned(16))) uint32_t qtab[ZKM_QUAD_TAB_WORDS];
    quad_tab_load(qtab);
    const poseidon_quad Q(threadIdx.x, qtab);
    const gl_t* col = (q & 1) ? c1 : c0;                  // words q and q + 4 have the parity of q
    uint64_t s[3] = {0, 0, 0};
    for (unsigned m = 0; m < 2 * arity; m += 8) {
        if (live) {
            s[0] = col[k * arity + ((m + q) >> 1)];
            s[1] = col[k * arity + ((m + 4 + q) >> 1)];
        }
        poseidon_permute_quad(s, Q);
    }
    if (live) digests[4 * k + q] = s[0];
}

void zkm_launch_merkle_leaves_ext(zkm_ctx* c, const gl_t* c0, const gl_t* c1, size_t nleaves, unsigned arity, gl_t* digests, size_t nseg,
                                  size_t val_seg, size_t dig_seg) {
    if (arity % 4 || 2 * arity <= 4) throw std::runtime_error("merkle_leaves_ext: unsupported arity");
    if (nseg == 0 || nseg > 65535) throw std::runtime_error("merkle_leaves_ext: bad segment count");
    zkm_prof_scope ps(c, "merkle_leaves_ext");
    const size_t hashes = nleaves * nseg;
    const unsigned z = (unsigned)nseg;
    if (hashes <= c->wide_max_hashes)    // the smallest layers: one hash per 16-lane row
        hipLaunchKernelGGL(k_merkle_leaves_ext_wide, dim3((nleaves * 16 + 255) / 256, 1, z), dim3(256), 0, c->stream, c0, c1, nleaves, arity, digests,
                           val_seg, dig_seg);
    else if (hashes <= c->quad_max_hashes)   // small layers: one hash per quad of lanes
        hipLaunchKernelGGL(k_merkle_leaves_ext_quad, dim3((nleaves * 4 + 255) / 256, 1, z), dim3(256), 0, c->stream, c0, c1, nleaves, arity, digests,
                           val_seg, dig_seg);
    else if (c->leaf_mfma)     // (the leaf kernel's matrix-core layer: layer 0 of a 2^22-row FRI proof 1.70 -> 1.62 ms)
        hipLaunchKernelGGL(k_merkle_leaves_ext_mfma, dim3((nleaves + 255) / 256, 1, z), dim3(256), 0, c->stream, c0, c1, nleaves, arity, digests, val_seg,
                           dig_seg);
    else
        hipLaunchKernelGGL(k_merkle_leaves_ext, dim3((nleaves + 255) / 256, 1, z), dim3(256), 0, c->stream, c0, c1, nleaves, arity, digests, val_seg,
                           dig_seg);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ---- a whole small tree in ONE launch, cap delivered to the host by the same launch (VERDICT r03 missing #4)
// Trees of up to 2^15 leaves (every commitment of a table of <= 2^13 rows, every FRI layer below that) used to be 2-3 launches for the
// levels plus the download kernel for the cap.  Here a workgroup owns the subtree under ONE cap entry (2^J leaf digests, J = log2(leaves)
// - cap_height <= 11), climbs its J levels through LDS -- the quad form of the permutation while a level has more parents than
// `wide_np_max`, the 16-lane form below -- stores every level (Merkle paths read them) and writes its cap entry to pinned host memory as
// well; the last workgroup to finish publishes the sequence number the host is polling (zkm_ctx::wait_flag).
struct merkle_tail_args {
    const gl_t* children;   // level 0: gridDim.x * 2^J digests
    gl_t* parents[11];      // levels 1 .. J
    uint32_t J, wide_np_max;
    uint64_t* host_cap;     // pinned: gridDim.z x gridDim.x digests
    uint64_t* flag;
    uint64_t seq;
    unsigned* counter;      // device word, zero between launches
    size_t dig_seg;         // blockIdx.z-th tree
};
__global__ __launch_bounds__(256) void k_merkle_tail(merkle_tail_args p) {
    ZKM_RAISE_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint64_t tail_lds[];
    const unsigned tid = threadIdx.x, lane = tid & 63, C = 1u << p.J;
    uint64_t* const buf0 = tail_lds;             // C digests
    uint64_t* const buf1 = tail_lds + 4 * C;     // C / 2 digests
    const size_t seg_off = (size_t)blockIdx.z * p.dig_seg;
    uint64_t* const host_cap = p.host_cap + 4 * ((size_t)blockIdx.z * gridDim.x + blockIdx.x);
    {
        const gl_t* src = p.children + seg_off + (size_t)blockIdx.x * C * 4;
        for (unsigned w = tid; w < C * 4; w += blockDim.x) buf0[w] = src[w];
    }
    __syncthreads();
    __shared__ __attribute__((aligned(16))) uint32_t qtab[ZKM_QUAD_TAB_WORDS];
    quad_tab_load(qtab);
    const poseidon_quad Q(tid, qtab);
    for (unsigned lvl = 0; lvl < p.J; lvl++) {
        const unsigned np = C >> (lvl + 1);
        const uint64_t* in = (lvl & 1) ? buf1 : buf0;
        uint64_t* out = (lvl & 1) ? buf0 : buf1;
        gl_t* g = p.parents[lvl] + seg_off + (size_t)blockIdx.x * np * 4;
        if (np > p.wide_np_max) {                                  // a quad of lanes per hash: blockDim / 4 hashes per round
            const unsigned q = tid & 3, slot = tid >> 2, wave_slot0 = (tid >> 6) << 4;
            for (unsigned h0 = 0; h0 < np; h0 += blockDim.x >> 2) {
                if (h0 + wave_slot0 >= np) continue;               // (uniform over the wave: every lane of a working wave permutes)
                const unsigned h = h0 + slot;
                const bool live = h < np;
                uint64_t st[3] = {live ? in[8 * h + q] : 0, live ? in[8 * h + 4 + q] : 0, 0};
                poseidon_permute_quad(st, Q);
                if (live) {
                    out[4 * h + q] = st[0];
                    g[4 * h + q] = st[0];
                    if (np == 1) host_cap[q] = st[0];
                }
            }
        } else {                                                   // 16 lanes per hash: blockDim / 16 hashes per round
            const unsigned idx = lane & 15, slot = tid >> 4, wave_slot0 = (tid >> 6) << 2;
            for (unsigned h0 = 0; h0 < np; h0 += blockDim.x >> 4) {
                if (h0 + wave_slot0 >= np) continue;
                const unsigned h = h0 + slot;
                const bool live = h < np;
                uint64_t x = (live && idx < 8) ? in[8 * h + idx] : 0;
                x = poseidon_permute_wide(x, lane);
                if (live && idx < 4) {
                    out[4 * h + idx] = x;
                    g[4 * h + idx] = x;
                    if (np == 1) host_cap[idx] = x;
                }
            }
        }
        __syncthreads();
    }
    // every workgroup's cap words are in host memory before its ticket is; the holder of the last ticket publishes
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = atomicAdd(p.counter, 1u);
        if (ticket == gridDim.x * gridDim.z - 1) {
            *p.counter = 0;
            __threadfence_system();
            __hip_atomic_store(p.flag, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// true: the levels above the leaves were built and the cap is in cap_out (host); false: not a tree this kernel takes
// (l0: the level the kernel starts from -- its "leaves" are the 2^(log_leaves - l0) nodes of level l0)
bool zkm_merkle_tail(zkm_ctx* c, gl_t* digests, const std::vector<size_t>& level_off, unsigned log_leaves, unsigned cap_height, uint64_t* cap_out,
                     unsigned l0, size_t nseg, size_t dig_seg) {
    const unsigned J = log_leaves - l0 - cap_height;
    if (!c->tree_tail || J == 0 || J > 11 || cap_height > 6) return false;
    // (a first level the tuning gives to the one-lane form -- more parents than both latency thresholds -- stays with the level kernels)
    if (((size_t)1 << (log_leaves - l0 - 1)) > c->quad_max_hashes && ((size_t)1 << (log_leaves - l0 - 1)) > c->wide_max_hashes) return false;
    // LDS: two level buffers of C and C / 2 digests (6 KB at the usual J = 7, 96 KB at J = 11) next to the static table of the quad form.
    // Only a launch beyond the default 64 KB needs the raised limit; a part that cannot grant it keeps the level kernels.
    const size_t tail_lds = (((size_t)1 << J) * 4 + ((size_t)1 << J) * 2) * sizeof(uint64_t);
    if (tail_lds + ZKM_QUAD_TAB_WORDS * sizeof(uint32_t) > 64 * 1024) {
        static std::atomic<uint64_t> lds_ok{0}, lds_bad{0};
        const uint64_t bit = (uint64_t)1 << (c->device & 63);
        if (lds_bad.load(std::memory_order_acquire) & bit) return false;
        if (!(lds_ok.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)k_merkle_tail, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                lds_bad.fetch_or(bit, std::memory_order_release);
                return false;
            }
            lds_ok.fetch_or(bit, std::memory_order_release);
        }
    }
    merkle_tail_args a{};
    a.children = digests + level_off[l0];
    for (unsigned k = 0; k < J; k++) a.parents[k] = digests + level_off[l0 + 1 + k];
    a.J = J;
    a.dig_seg = dig_seg;
    a.wide_np_max = c->wide_max_hashes ? 16 : 0;                      // 16 lanes per hash where a level is ONE round of them (16 slots per
                                                                      // 256 threads: 15.6 us against 24 us for a round of quads)
    const size_t capw = ((size_t)4 << cap_height) * nseg;
    const uint64_t seq = c->xfer_begin(capw * 8, &a.host_cap, &a.flag, &a.counter);
    a.seq = seq;
    const size_t C = (size_t)1 << J;
    unsigned threads = (unsigned)(C >> 1) * 4;                        // a quad per parent of the first level, one wave per SIMD at most
    if (threads > 256) threads = 256;
    if (threads < 64) threads = 64;
    {
        zkm_prof_scope ps(c, "merkle_compress");
        hipLaunchKernelGGL(k_merkle_tail, dim3(1u << cap_height, 1, (unsigned)nseg), dim3(threads), (C * 4 + C * 2) * sizeof(uint64_t), c->stream, a);
        ZKM_HIP_CHECK(hipGetLastError());
    }
    c->xfer_finish(seq, cap_out, capw * 8);
    return true;
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

// All digest levels above the leaves, up to the cap: fused launches (k_merkle_fused: <= 3 levels, one hash per lane, while the first
// level of the launch has >= 2^15 nodes; k_merkle_fused_quad: four lanes per hash, the two levels with 2^14 and 2^13 parents;
// k_merkle_fused_wide: <= 6 levels, 16 lanes per hash, from 2^12 parents on).  A 2^22-leaf tree with a
// 16-digest cap (18 levels) is 3 + 2 launches instead of 18.  (Measured and dropped, profiles/r03_merkle_inner_levels.txt: a lane
// reducing a 16-node subtree by itself -- no barrier, 15 back-to-back permutations -- leaves 4096 waves for a 2^22-leaf tree, four per
// SIMD, and the permutation needs five to saturate the issue port: 2.65 G permutations/s against 2.7 here and 3.33 in the leaf kernel.)
// The four-lane form takes ~40 % of the latency of a permutation at 1.7x its issue slots (330 instead of 190 wave instructions per
// hash), so it is kept to the levels where a launch is mostly latency.
void zkm_merkle_build_inner(zkm_ctx* c, gl_t* digests, const std::vector<size_t>& level_off, unsigned log_leaves,
                            unsigned cap_height, size_t nseg, size_t dig_seg) {
    const unsigned top = log_leaves - cap_height;
    const unsigned z = (unsigned)nseg;
    unsigned l = 0;                                        // level of the children of the next launch
    while (l < top) {
        zkm_prof_scope ps(c, "merkle_compress");
        const unsigned log_p1 = log_leaves - l - 1;       // log2(#parents) of the first level made
        const unsigned rem = top - l;
        const size_t p1 = (size_t)1 << log_p1, h1 = p1 * nseg;   // (h1: hashes of the launch's first level over all stacked trees)
        if (h1 > c->quad_max_hashes && h1 > c->wide_max_hashes && log_p1 >= 8) {   // (the one-lane kernel works on blocks of 256 parents)
            merkle_fused_args a{};
            a.dig_seg = dig_seg;
            a.children = digests + level_off[l];
            a.levels = rem < 3 ? rem : 3;
            for (unsigned k = 0; k < a.levels; k++) a.parents[k] = digests + level_off[l + 1 + k];
            // (the matrix-core layer was measured here too, rounds 5 and 6: the 24 KB it parks in LDS next to this kernel's 16 KB leave four
            // workgroups per CU -- merkle_compress 1.59 -> 1.72 ms per 262 x 2^20 commitment, lock-step segments 108 -> 106/s: not adopted)
            hipLaunchKernelGGL(k_merkle_fused, dim3((unsigned)(((size_t)1 << log_p1) / 256), 1, z), dim3(256), 0, c->stream, a);
            l += a.levels;
        } else if (h1 <= c->wide_max_hashes) {
            merkle_fused_wide_args a{};
            a.dig_seg = dig_seg;
            a.children = digests + level_off[l];
            unsigned J = rem < 6 ? rem : 6;
            if (J > log_p1 + 1) J = log_p1 + 1;           // (a subtree cannot have more children than the level)
            a.J = J;
            for (unsigned k = 0; k < J; k++) a.parents[k] = digests + level_off[l + 1 + k];
            hipLaunchKernelGGL(k_merkle_fused_wide, dim3((unsigned)(((size_t)2 << log_p1) >> J), 1, z), dim3(256), 0, c->stream, a);
            l += J;
        } else {
            merkle_fused_quad_args a{};
            a.dig_seg = dig_seg;
            a.children = digests + level_off[l];
            unsigned J = 1;                                  // the levels that are still too large for the 16-lane form (default: 2^14 and 2^13 parents)
            while (J < rem && J < 7 && J <= log_p1 && (h1 >> J) > c->wide_max_hashes) J++;
            a.J = J;
            for (unsigned k = 0; k < J; k++) a.parents[k] = digests + level_off[l + 1 + k];
            hipLaunchKernelGGL(k_merkle_fused_quad, dim3((unsigned)(((size_t)2 << log_p1) >> J), 1, z), dim3(256), 0, c->stream, a);
            l += J;
        }
        ZKM_HIP_CHECK(hipGetLastError());
    }
}

// caps of nseg stacked trees from their cap levels: one download (segment s: 4 << cap_height words at digests + s * dig_seg + level_off[top])
static void download_caps(zkm_ctx* c, const gl_t* digests, size_t cap_off, unsigned cap_height, uint64_t* cap_out, size_t nseg, size_t dig_seg) {
    const size_t capw = (size_t)4 << cap_height;
    if (nseg == 1) {
        c->download(cap_out, digests + cap_off, capw * sizeof(uint64_t));
        return;
    }
    zkm_scratch tmp(c, nseg * capw * sizeof(gl_t));
    ZKM_HIP_CHECK(hipMemcpy2DAsync(tmp.p, capw * sizeof(gl_t), digests + cap_off, dig_seg * sizeof(gl_t), capw * sizeof(gl_t), nseg,
                                   hipMemcpyDeviceToDevice, c->stream));
    c->download(cap_out, tmp.p, nseg * capw * sizeof(uint64_t));
}

void zkm_merkle_build_inner_cap(zkm_ctx* c, gl_t* digests, const std::vector<size_t>& level_off, unsigned log_leaves, unsigned cap_height,
                                uint64_t* cap_out, size_t nseg, size_t dig_seg) {
    // A workgroup of the tail kernel owns everything under one cap entry, and a chain of permutations is fastest with ONE wave per SIMD
    // (tools/ubench_perm_latency.hip: 24 us per quad permutation alone, 46 us with four waves per SIMD): 256 threads, 64 quads per
    // round.  From 2^11 nodes on a level is one round; with more to start from the first levels took several rounds on 16 CUs while
    // the rest of the GPU watched (measured with 2^13 / 2^15: +1.5-2 ms per 2^16-cycle segment).  Larger trees climb to 2^11 nodes with
    // the level kernels first (many workgroups, one round per level).
    const unsigned TAIL_LOG = 11;
    if (c->tree_tail && log_leaves <= 24 && log_leaves >= cap_height + 1) {
        const unsigned l0 = log_leaves > TAIL_LOG ? log_leaves - TAIL_LOG : 0;
        if (log_leaves - l0 > cap_height) {
            if (l0) zkm_merkle_build_inner(c, digests, level_off, log_leaves, log_leaves - l0, nseg, dig_seg);    // levels 1 .. l0 ("cap" = level l0)
            if (zkm_merkle_tail(c, digests, level_off, log_leaves, cap_height, cap_out, l0, nseg, dig_seg)) return;
            // (refused: finish with the level kernels from where we are -- build_inner starts at level 0, so only without a head start)
            if (l0) {
                std::vector<size_t> rest(level_off.begin() + l0, level_off.end());
                zkm_merkle_build_inner(c, digests, rest, log_leaves - l0, cap_height, nseg, dig_seg);
                download_caps(c, digests, level_off[log_leaves - cap_height], cap_height, cap_out, nseg, dig_seg);
                return;
            }
        }
    }
    zkm_merkle_build_inner(c, digests, level_off, log_leaves, cap_height, nseg, dig_seg);
    download_caps(c, digests, level_off[log_leaves - cap_height], cap_height, cap_out, nseg, dig_seg);
}

// ------------------------------------------------------------------ Keccak-f[1600] batch (K15)
// States are 25-word records (AoS: the layout of the reference's [u64; 25], cpu/kernel/keccak_util.rs:6-31), one state per lane:
// ~6.3k VALU instructions per permutation (xor, v_alignbit_b32 rotations, v_bfi chi) for 400 B moved.  Each lane moves its own record
// with 16-byte accesses (records are 8-byte aligned: 200 = 8 * 25); a wave's 64 records are 12.8 KB of contiguous memory, so every
// cache line comes from HBM once and the partial-line accesses are served by L1 / L2.  Measured on 2^22 states
// (profiles/r03_keccakf_variants.txt): 8 B accesses + generic 64-bit rotates 0.96 ms (r02); this form 0.82 ms = 5.1 G permutations/s;
// the records staged and transposed through LDS with fully coalesced 512 B accesses 0.91 ms (one 12.8 KB image per wave: three
// waves per SIMD) and 1.01 ms (two 6.4 KB halves: four waves, twice the LDS instructions) -- the transposition costs more than
// the uncoalesced 16 B accesses do.
struct __attribute__((aligned(8))) keccak_pair { uint64_t x, y; };

__global__ __launch_bounds__(256) void k_keccakf(uint64_t* __restrict__ states, size_t k) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= k) return;
    uint64_t a[25];
    uint64_t* st = states + idx * 25;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const keccak_pair v = *reinterpret_cast<const keccak_pair*>(st + 2 * i);
        a[2 * i] = v.x;
        a[2 * i + 1] = v.y;
    }
    a[24] = st[24];
    keccakf_dev(a);
#pragma unroll
    for (int i = 0; i < 12; i++) *reinterpret_cast<keccak_pair*>(st + 2 * i) = keccak_pair{a[2 * i], a[2 * i + 1]};
    st[24] = a[24];
}

void zkm_launch_keccakf(zkm_ctx* c, uint64_t* states, size_t k) {
    if (!k) return;
    zkm_prof_scope ps(c, "keccakf");
    hipLaunchKernelGGL(k_keccakf, dim3((k + 255) / 256), dim3(256), 0, c->stream, states, k);
    ZKM_HIP_CHECK(hipGetLastError());
}
