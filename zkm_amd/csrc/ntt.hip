// ntt.hip -- batched Goldilocks NTT / coset LDE kernels for gfx950 (K1, K2, K9, K12).
//
// Replaces plonky2's PolynomialValues::{ifft, coset_ifft} and PolynomialCoeffs::{coset_fft, lde}
// as called from PolynomialBatch::from_values / from_coeffs (reference call sites
// prover/src/prover.rs:154-163, 514-521, 579-586) and compute_quotient_polys (:678-681, :787).
//
// Orders: a decimation-in-frequency transform maps natural-order input to bit-reversed output, which
// is exactly the row order of plonky2's Merkle leaves (reverse_index_bits_in_place after the LDE,
// SURVEY App. A.5) -- so the LDE needs no separate permutation pass.
//
// Twiddles: per-stage tables tw[(1<<s) + j] = w_{2^(s+1)}^j, so the butterflies of one stage read
// consecutive entries with consecutive lanes.
#include <atomic>
#include "zkm_internal.h"

// ------------------------------------------------------------------ twiddle / power tables
__global__ void k_build_twiddles(gl_t* tw, unsigned log_max, bool inverse) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)1 << log_max;
    if (idx >= total) return;
    if (idx == 0) { tw[0] = 0; return; }
    unsigned s = 63 - __clzll((unsigned long long)idx);  // stage: table of w_{2^(s+1)}
    size_t j = idx - ((size_t)1 << s);
    gl_t root = gl_root_of_unity(s + 1);
    if (inverse) root = gl_inv(root);
    tw[idx] = gl_pow(root, j);
}

void zkm_ctx::ensure_twiddles(unsigned log_n) {
    if (log_n <= tw.log_max && tw.fwd) return;
    unsigned lm = log_n < 12 ? 12 : log_n;
    if (tw.fwd) { release(tw.fwd); release(tw.inv); resident_bytes -= 2 * (sizeof(gl_t) << tw.log_max); }
    size_t total = (size_t)1 << lm;
    resident_bytes += 2 * total * sizeof(gl_t);
    tw.fwd = (gl_t*)alloc(total * sizeof(gl_t));
    tw.inv = (gl_t*)alloc(total * sizeof(gl_t));
    hipLaunchKernelGGL(k_build_twiddles, dim3((total + 255) / 256), dim3(256), 0, stream, tw.fwd, lm, false);
    hipLaunchKernelGGL(k_build_twiddles, dim3((total + 255) / 256), dim3(256), 0, stream, tw.inv, lm, true);
    ZKM_HIP_CHECK(hipGetLastError());
    tw.log_max = lm;
}

// two-level power table for shift^i, i < 2^log_n: lo[i & (2^h - 1)] * hi[i >> h], h = ceil(log_n / 2)
const gl_t* zkm_ctx::pow_table(uint64_t shift, unsigned log_n) {
    auto key = std::make_pair(shift, log_n);
    auto it = pow_tables.find(key);
    if (it != pow_tables.end()) return it->second;
    unsigned h = (log_n + 1) / 2;
    size_t nlo = (size_t)1 << h, nhi = (size_t)1 << (log_n - h);
    std::vector<gl_t> host(nlo + nhi);
    gl_t p = 1;
    for (size_t i = 0; i < nlo; i++) { host[i] = p; p = gl_mul(p, shift); }
    gl_t step = p;  // shift^(2^h)
    p = 1;
    for (size_t i = 0; i < nhi; i++) { host[nlo + i] = p; p = gl_mul(p, step); }
    gl_t* d = (gl_t*)alloc(host.size() * sizeof(gl_t));
    ZKM_HIP_CHECK(hipMemcpyAsync(d, host.data(), host.size() * sizeof(gl_t), hipMemcpyHostToDevice, stream));
    ZKM_HIP_CHECK(hipStreamSynchronize(stream));  // host vector goes out of scope
    pow_tables[key] = d;
    resident_bytes += host.size() * sizeof(gl_t);
    return d;
}

__device__ __forceinline__ gl_t pow_lookup(const gl_t* __restrict__ tab, unsigned log_n, size_t i) {
    unsigned h = (log_n + 1) / 2;
    return gl_mul_loose(tab[i & (((size_t)1 << h) - 1)], tab[((size_t)1 << h) + (i >> h)]);
}

// ------------------------------------------------------------------ field-primitive self test (parity / debug)
// out[0][i] = a + b, out[1][i] = a - b, out[2][i] = a * 2^24, out[3][i] = a * 2^48, out[4][i] = a * 2^72, out[5][i] = a * b (out[6]: the same
// product from stark.hip's branch-free build of gl_mul_loose), all mod p and
// canonical, for ANY 64-bit words a, b (also >= p): the loose-arithmetic primitives of the butterflies (gl_add_rr / gl_sub_rr with their
// never-taken second-correction branches, gl_mul_pow2, gl_mul_loose) on exactly the inputs random data does not produce.
__global__ void k_field_selftest(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, size_t n, uint64_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = a[i], y = b[i];
    out[i] = gl_canon(gl_add_rr(x, y));
    out[n + i] = gl_canon(gl_sub_rr(x, y));
    out[2 * n + i] = gl_canon(gl_mul_pow2<24>(x));
    out[3 * n + i] = gl_canon(gl_mul_pow2<48>(x));
    out[4 * n + i] = gl_canon(gl_mul_pow2<72>(x));
    out[5 * n + i] = gl_mul(x, y);
}
extern "C" int zkm_field_selftest(zkm_ctx* c, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        zkm_scratch da(c, n * 8), db(c, n * 8), dout(c, 7 * n * 8);
        ZKM_HIP_CHECK(hipMemcpyAsync(da.p, a, n * 8, hipMemcpyHostToDevice, c->stream));
        ZKM_HIP_CHECK(hipMemcpyAsync(db.p, b, n * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_field_selftest, dim3((n + 255) / 256), dim3(256), 0, c->stream, da.as<uint64_t>(), db.as<uint64_t>(), n, dout.as<uint64_t>());
        ZKM_HIP_CHECK(hipGetLastError());
        zkm_launch_mul_selftest_branchfree(c, da.as<uint64_t>(), db.as<uint64_t>(), n, dout.as<uint64_t>() + 6 * n);
        ZKM_HIP_CHECK(hipMemcpyAsync(out, dout.p, 7 * n * 8, hipMemcpyDeviceToHost, c->stream));
        c->sync();
    } catch (const std::exception& e) {
        if (err) *err = strdup(e.what());
        return 1;
    }
    return 0;
}

// ------------------------------------------------------------------ multi-stage LDS pass kernel
// One launch performs S (3..8) consecutive radix-2 DIF stages of many length-2^S sub-transforms.
// A workgroup owns a tile of R = 2^S "rows" (the butterfly dimension, element stride sa) x T "columns"
// (independent sub-transforms, element stride sb); one of the two dimensions is contiguous in HBM and
// the lanes of a wave run along it for every global access (T*8 B or R*8 B segments).  The tile is
// staged in LDS ([R][T(+1)]); every thread keeps 8 rows in registers per round and does up to three
// stages (12 butterflies) between LDS exchanges, with lanes along the T dimension (conflict-free
// ds_read/write_b64).  Twiddles of all rounds live in registers and are reused for every polynomial
// column the workgroup loops over, so twiddle traffic is amortised over `cpb` columns.
//
// In-place DIF leaves sub-transform outputs bit-reversed along the row dimension; storing LDS row a'
// at output row bitrev(a') instead (rev_rows) yields natural order for free, which is how the
// natural->natural transforms (values -> coefficients) avoid a separate permutation pass.
struct ntt_pass_args {
    const gl_t* in;
    gl_t* out;
    size_t cs_in, cs_out;
    uint32_t ncols, cpb;
    uint32_t log_T, tp, n_lo;
    size_t bi_hi, bi_lo, bo_hi, bo_lo;
    size_t sa_in, sb_in, sa_out, sb_out;
    uint32_t in_contig_a, out_contig_a, rev_rows, m;
    uint32_t zero_padded;  // first pass of a x4 zero-padded transform: rows >= 2^S / 4 of every tile are zero
    uint32_t canon_out;  // last pass of a transform: outputs must be canonical (intermediate passes may store loose words)
    uint32_t pow2_last;  // last pass (m == 0) with S % 3 == 0: the last round's twiddles are powers of two (1 forward, 2 inverse roots)
    const gl_t* tw;
    const gl_t* pre_tab;
    uint32_t pre_log;
    size_t n_in;
    gl_t post_scale;
    const gl_t* post_tab;
    uint32_t post_log;
    // coset fusion (first pass of the coset-split LDE): ncoset independent transforms of the SAME input, coset k pre-scaled with
    // pre_tab_k[k] and written at out + out_off_k[k]; workgroups of one tile's cosets are dispatched next to each other on one XCD
    uint32_t ncoset;
    const gl_t* pre_tab_k[4];
    size_t out_off_k[4];
    // PRE == 3: the pre-scale s_k^t of element t = t0 + j * row_step is B * D_k^j with B = s_k^t0 looked up once per thread and
    // pre_dj[k][j] = (s_k^row_step)^j wave-uniform: no table access inside the column loop
    gl_t pre_dj[4][8];
};

template <int S, int K>
struct ntt_round {
    static constexpr int st = S - 1 - 3 * K;           // top local stage handled by this round
    static constexpr int q = st >= 2 ? st - 2 : 0;     // position of the 3 register-resident row bits
    static constexpr int top = st - q;                 // highest j-bit that is a stage of this round
    __device__ static __forceinline__ int row(int rg, int j) { return ((rg >> q) << (q + 3)) | (j << q) | (rg & ((1 << q) - 1)); }
    __device__ static __forceinline__ void load_tw(gl_t (&w)[7], const gl_t* __restrict__ tw, int rg, uint32_t m, size_t i_low) {
        const int rlow = rg & ((1 << q) - 1);
#pragma unroll
        for (int jb = 0; jb < 3; jb++) {
            if (jb > top) continue;
            const int sl = q + jb;  // local stage
            const int base = jb == 2 ? 0 : (jb == 1 ? 4 : 6);
#pragma unroll
            for (int jl = 0; jl < (1 << jb); jl++) {
                size_t amod = (size_t)(rlow | (jl << q));
                w[base + jl] = tw[((size_t)1 << (sl + m)) + (amod << m) + i_low];
            }
        }
    }
    __device__ static __forceinline__ void bfly(gl_t& u, gl_t& v, gl_t w) {
        // all-loose arithmetic inside a pass: the sum and the difference take their one probable correction from the add's / subtract's
        // own carry-out and the improbable second one on a never-taken uniform branch (gl_add_rr / gl_sub_rr); the product is not
        // canonicalised at all; the last pass canonicalises on the way out (canon_out)
        const uint64_t t = gl_add_rr(u, v);
        v = gl_mul_loose(gl_sub_rr(u, v), w);
        u = t;
    }
    // up to three DIF stages on the 8 register-resident rows; sched_barrier keeps the compiler from
    // interleaving all 12 butterflies (which costs >200 VGPRs)
    __device__ static __forceinline__ void compute(gl_t (&x)[8], const gl_t (&w)[7]) {
        if (top >= 2) {
            bfly(x[0], x[4], w[0]); bfly(x[1], x[5], w[1]);
            __builtin_amdgcn_sched_barrier(0);
            bfly(x[2], x[6], w[2]); bfly(x[3], x[7], w[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (top >= 1) {
            bfly(x[0], x[2], w[4]); bfly(x[1], x[3], w[5]);
            __builtin_amdgcn_sched_barrier(0);
            bfly(x[4], x[6], w[4]); bfly(x[5], x[7], w[5]);
            __builtin_amdgcn_sched_barrier(0);
        }
        bfly(x[0], x[1], w[6]); bfly(x[2], x[3], w[6]);
        __builtin_amdgcn_sched_barrier(0);
        bfly(x[4], x[5], w[6]); bfly(x[6], x[7], w[6]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // The LAST round of the LAST pass (q == 0, m == 0): its twiddles are the 8th / 4th roots of unity w_8^j = 2^(24 j), w_4 = 2^48 (and
    // 1 for the final stage), so 7 of the 12 butterflies need no product at all and the other 5 a shift (gl_mul_pow2).  Inverse
    // roots are the negated powers w_8^-j = -2^(96 - 24 j): the sign goes into the subtraction (v - u instead of u - v).  Same field elements as compute() with the table twiddles, hence bit-exact.
    template <int E, bool NEG>
    __device__ static __forceinline__ void bfly_pow2(gl_t& u, gl_t& v) {
        const uint64_t t = gl_add_rr(u, v), d = NEG ? gl_sub_rr(v, u) : gl_sub_rr(u, v);
        v = E ? gl_mul_pow2<(E ? E : 1)>(d) : d;
        u = t;
    }
    template <bool INV>
    __device__ static __forceinline__ void compute_pow2(gl_t (&x)[8]) {
        static_assert(top == 2, "compute_pow2: three stages");   // (as a round of its own: the three LOWEST stages, q == 0, m == 0)
        if (!INV) {
            bfly_pow2<0, false>(x[0], x[4]); bfly_pow2<24, false>(x[1], x[5]);
            __builtin_amdgcn_sched_barrier(0);
            bfly_pow2<48, false>(x[2], x[6]); bfly_pow2<72, false>(x[3], x[7]);
            __builtin_amdgcn_sched_barrier(0);
            bfly_pow2<0, false>(x[0], x[2]); bfly_pow2<48, false>(x[1], x[3]);
            __builtin_amdgcn_sched_barrier(0);
            bfly_pow2<0, false>(x[4], x[6]); bfly_pow2<48, false>(x[5], x[7]);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            bfly_pow2<0, false>(x[0], x[4]); bfly_pow2<72, true>(x[1], x[5]);     // w_8^-1 = -2^72
            __builtin_amdgcn_sched_barrier(0);
            bfly_pow2<48, true>(x[2], x[6]); bfly_pow2<24, true>(x[3], x[7]);     // w_8^-2 = -2^48, w_8^-3 = -2^24
            __builtin_amdgcn_sched_barrier(0);
            bfly_pow2<0, false>(x[0], x[2]); bfly_pow2<48, true>(x[1], x[3]);     // w_4^-1 = -2^48
            __builtin_amdgcn_sched_barrier(0);
            bfly_pow2<0, false>(x[4], x[6]); bfly_pow2<48, true>(x[5], x[7]);
            __builtin_amdgcn_sched_barrier(0);
        }
        bfly_pow2<0, false>(x[0], x[1]); bfly_pow2<0, false>(x[2], x[3]);
        __builtin_amdgcn_sched_barrier(0);
        bfly_pow2<0, false>(x[4], x[5]); bfly_pow2<0, false>(x[6], x[7]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // The same three stages with the twiddles FACTORED OUT of the butterflies (radix-8 decimation in frequency).  The stage twiddles
    // of a round are w_top[j] = W w_8^j, w_mid[j] = W^2 w_4^j, w_bot = W^4 for ONE W = w_N'^a' per thread (load_tw), and
    // w_8 = 2^24, w_4 = 2^48 are shifts in this field, so the twelve butterflies with twelve general products are
    //     y = DFT_8(x) with shift twiddles only (compute_pow2: 5 shift-multiplies, 7 plain butterflies),
    //     then  slot j *= W^bitrev3(j)   (7 general products; slot j of the in-place DIF holds output bitrev3(j))
    // -- the same field elements, 7 general products + 5 shifts instead of 12 general products per 8 points and round.
    // f[k - 1] = W^k, k = 1 .. 7 (factored_tw below).
    template <bool INV>
    __device__ static __forceinline__ void compute_fact(gl_t (&x)[8], const gl_t (&f)[7]) {
        compute_pow2<INV>(x);
        x[1] = gl_mul_loose(x[1], f[3]); x[2] = gl_mul_loose(x[2], f[1]);
        __builtin_amdgcn_sched_barrier(0);
        x[3] = gl_mul_loose(x[3], f[5]); x[4] = gl_mul_loose(x[4], f[0]);
        __builtin_amdgcn_sched_barrier(0);
        x[5] = gl_mul_loose(x[5], f[4]); x[6] = gl_mul_loose(x[6], f[2]);
        __builtin_amdgcn_sched_barrier(0);
        x[7] = gl_mul_loose(x[7], f[6]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // stage twiddles of a full round (load_tw: w[0] = W, w[4] = W^2, w[6] = W^4) -> the powers W^1 .. W^7
    __device__ static __forceinline__ void factored_tw(gl_t (&f)[7], const gl_t (&w)[7]) {
        static_assert(top == 2, "factored_tw: a full three-stage round");
        f[0] = w[0]; f[1] = w[4]; f[3] = w[6];
        f[2] = gl_mul(w[0], w[4]);
        f[4] = gl_mul(w[0], w[6]);
        f[5] = gl_mul(w[4], w[6]);
        f[6] = gl_mul(f[2], w[6]);
    }
    // First round of a x4 zero-padded transform (coset LDE): only x[0], x[1] are non-zero, so the first two stages are
    // plain twiddle multiplications -- bfly(u, 0) = (u, u w) -- 6 products instead of 8 butterflies.
    __device__ static __forceinline__ void compute_zero_padded(gl_t (&x)[8], const gl_t (&w)[7]) {
        x[4] = gl_mul_loose(x[0], w[0]); x[5] = gl_mul_loose(x[1], w[1]);
        __builtin_amdgcn_sched_barrier(0);
        x[2] = gl_mul_loose(x[0], w[4]); x[3] = gl_mul_loose(x[1], w[5]);
        __builtin_amdgcn_sched_barrier(0);
        x[6] = gl_mul_loose(x[4], w[4]); x[7] = gl_mul_loose(x[5], w[5]);
        __builtin_amdgcn_sched_barrier(0);
        bfly(x[0], x[1], w[6]); bfly(x[2], x[3], w[6]);
        __builtin_amdgcn_sched_barrier(0);
        bfly(x[4], x[5], w[6]); bfly(x[6], x[7], w[6]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ static __forceinline__ void lds_read(const gl_t* lds, int tp, int b, int rg, gl_t (&x)[8]) {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = lds[row(rg, j) * tp + b];
    }
    __device__ static __forceinline__ void lds_write(gl_t* lds, int tp, int b, int rg, const gl_t (&x)[8]) {
#pragma unroll
        for (int j = 0; j < 8; j++) lds[row(rg, j) * tp + b] = x[j];
    }
};

// the last round of a pass: with the shift twiddles where the pass allows them (pow2_last != 0 implies m == 0)
template <int S, int K>
__device__ __forceinline__ void ntt_last_round(gl_t (&x)[8], const gl_t (&w)[7], uint32_t pow2_last) {
    using R = ntt_round<S, K>;
    if constexpr (S % 3 == 0 && K == (S + 2) / 3 - 1) {
        if (pow2_last == 1) { R::template compute_pow2<false>(x); return; }
        if (pow2_last == 2) { R::template compute_pow2<true>(x); return; }
    }
    R::compute(x, w);
}

// IN_A / OUT_A: the butterfly (row) dimension is the contiguous one in HBM on the input / output side.
// When it is not, the thread's 8 register-resident rows of the first (last) round are exactly what it
// loads (stores) -- T*8 B contiguous per row across the wave -- so the tile never passes through LDS on
// that side; LDS is then only the exchange buffer between rounds.
// PRE / POST: 0 = never, 1 = always, 2 = decided at run time (keeps the scale paths out of the register
// allocation of the hot variants)
// ZP: the input is shorter than the transform (rows beyond n_in read as zero): only then are loads bounds-checked.
// PF: software-prefetch the next column (16 VGPRs); off where the kernel must stay within 128 VGPRs for two workgroups per CU.
template <int S, bool IN_A, bool OUT_A, int PRE, int POST, bool ZP = false, bool PF = true>
__global__ __launch_bounds__(512, 2) void k_ntt_pass(ntt_pass_args p) {
    extern __shared__ __attribute__((aligned(16))) gl_t lds[];
    constexpr int R = 1 << S, NR = (S + 2) / 3;
    using R0 = ntt_round<S, 0>;
    using R1 = ntt_round<S, (NR > 1 ? 1 : 0)>;
    using R2 = ntt_round<S, (NR > 2 ? 2 : 0)>;
    using RL = ntt_round<S, NR - 1>;  // last round: q == 0, rows (rg << 3) | j
    const bool do_pre = PRE == 1 || (PRE == 2 && (p.pre_tab != nullptr || p.ncoset > 1));
    static_assert(PRE != 3 || (!IN_A && !ZP), "factored pre-scale: strided input only");
    const bool do_post = POST == 1 || (POST == 2 && (p.post_scale != 1 || p.post_tab != nullptr));
    const int logT = p.log_T, T = 1 << logT, tp = p.tp;
    const int nthreads = (R >> 3) << logT;
    const int tid = threadIdx.x;
    const int b = tid & (T - 1), rg = tid >> logT;
    // coset-fused launches: consecutive workgroup ids go round-robin over the 8 XCDs, so the ncoset workgroups of one tile get ids
    // 8 apart -- same XCD, same L2: the input tile comes from HBM once and from L2 ncoset - 1 times
    uint32_t tile = blockIdx.x, coset = 0;
    if (p.ncoset > 1) {
        const uint32_t g = blockIdx.x >> 3;
        coset = g % p.ncoset;
        tile = (g / p.ncoset) * 8 + (blockIdx.x & 7);
    }
    const gl_t* const pre_tab = p.ncoset > 1 ? p.pre_tab_k[coset] : p.pre_tab;
    const size_t out_off = p.ncoset > 1 ? p.out_off_k[coset] : 0;
    const uint32_t t_hi = tile / p.n_lo, t_lo = tile % p.n_lo;
    const size_t base_in = (size_t)t_hi * p.bi_hi + (size_t)t_lo * p.bi_lo;
    const size_t base_out = (size_t)t_hi * p.bo_hi + (size_t)t_lo * p.bo_lo;
    const size_t i_low = p.m ? (((size_t)t_lo << logT) + b) : 0;

    gl_t w0[7], w1[7], w2[7];
    R0::load_tw(w0, p.tw, rg, p.m, i_low);
    if (NR > 1) R1::load_tw(w1, p.tw, rg, p.m, i_low);
    if (NR > 2) R2::load_tw(w2, p.tw, rg, p.m, i_low);

    // column-independent offsets.  Every global access is (wave-uniform base) + (one 32-bit lane offset): the 8 row bases of a
    // thread differ by uniform multiples of the row step, so the addresses cost scalar adds, not a VGPR pair per row.
    // direct input: rows rg + j*(R/8) (== R0::row(rg, j)), column b
    const uint32_t lane_in = (uint32_t)((size_t)rg * p.sa_in + (size_t)b * p.sb_in);
    const size_t in_step = (size_t)(R >> 3) * p.sa_in;
    // direct output: rows of the last round, optionally bit-reversed: bitrev((rg<<3)|j) = brev3(j)*(R/8) + bitrev(rg, S-3)
    const uint32_t lane_out = (uint32_t)((size_t)(p.rev_rows ? bitrev32((uint32_t)rg, S - 3) : (uint32_t)(rg << 3)) * p.sa_out + (size_t)b * p.sb_out);
    const size_t out_step = p.rev_rows ? (size_t)(R >> 3) * p.sa_out : p.sa_out;
    // (the coset pre-scale table and the bounds check index by element: 32-bit element offsets, a column holds < 2^29 elements)
    const uint32_t in0 = (uint32_t)base_in + lane_in, in_step32 = (uint32_t)in_step;

    const uint32_t col0 = blockIdx.y * p.cpb;
    const uint32_t col1 = col0 + p.cpb < p.ncols ? col0 + p.cpb : p.ncols;
    // Software pipeline over the workgroup's columns: the 8 words of column c + 1 are requested before column c is
    // transformed (16 VGPRs), so the HBM latency of the next loads is covered by ~800 VALU instructions instead of
    // being exposed at the top of every iteration.  nx[] holds raw words; the coset pre-scale is applied when they are consumed.
    gl_t gpre[8];
    if (PRE == 3) {
        const gl_t B = pow_lookup(pre_tab, p.pre_log, in0);
#pragma unroll
        for (int j = 0; j < 8; j++) gpre[j] = j ? gl_mul(B, p.pre_dj[coset][j]) : gl_canon(B);
    }
    gl_t nx[8];
    auto fetch = [&](uint32_t col) {
        const gl_t* __restrict__ src = p.in + (size_t)col * p.cs_in + base_in;   // wave-uniform
        if (!IN_A) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const gl_t* row = src + (size_t)j * in_step;                       // still uniform
                nx[j] = (!ZP || in0 + (uint32_t)j * in_step32 < p.n_in) ? row[lane_in] : 0;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                int idx = tid + e * nthreads;
                int a = idx & (R - 1), bb = idx >> S;
                size_t off = (size_t)a * p.sa_in + (size_t)bb * p.sb_in;
                nx[e] = (!ZP || base_in + off < p.n_in) ? src[off] : 0;
            }
        }
    };
    if (PF && col0 < col1) fetch(col0);
    for (uint32_t col = col0; col < col1; col++) {
        if (!PF) fetch(col);
        gl_t* __restrict__ dst = p.out + (size_t)col * p.cs_out + out_off + (OUT_A ? 0 : base_out);   // wave-uniform
        gl_t x[8];
        if (!IN_A) {
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = nx[j];
            __builtin_amdgcn_sched_barrier(0);
            if (PRE == 3) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    x[j] = gl_mul_loose(x[j], gpre[j]);
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);
                }
            } else if (do_pre) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t off = in0 + (uint32_t)j * in_step32;
                    if (!ZP || off < p.n_in) x[j] = gl_mul_loose(x[j], pow_lookup(pre_tab, p.pre_log, off));  // (the table only covers the unpadded input)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // transposed staging: lanes run along the contiguous row dimension
#pragma unroll 2
            for (int e = 0; e < 8; e++) {
                int idx = tid + e * nthreads;
                int a = idx & (R - 1), bb = idx >> S;
                size_t off = base_in + (size_t)a * p.sa_in + (size_t)bb * p.sb_in;
                gl_t v = nx[e];
                if (do_pre && (!ZP || off < p.n_in)) v = gl_mul_loose(v, pow_lookup(pre_tab, p.pre_log, off));
                lds[a * tp + bb] = v;
            }
            __syncthreads();
            R0::lds_read(lds, tp, b, rg, x);
        }
        if (PF && col + 1 < col1) fetch(col + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!IN_A && S >= 3 && p.zero_padded) R0::compute_zero_padded(x, w0);
        else if (NR == 1) ntt_last_round<S, 0>(x, w0, p.pow2_last);
        else R0::compute(x, w0);
        if (NR > 1) {
            R0::lds_write(lds, tp, b, rg, x);
            __syncthreads();
            R1::lds_read(lds, tp, b, rg, x);
            if (NR == 2) ntt_last_round<S, (NR > 1 ? 1 : 0)>(x, w1, p.pow2_last);
            else R1::compute(x, w1);
        }
        if (NR > 2) {
            R1::lds_write(lds, tp, b, rg, x);
            __syncthreads();
            R2::lds_read(lds, tp, b, rg, x);
            if (NR == 3) ntt_last_round<S, (NR > 2 ? 2 : 0)>(x, w2, p.pow2_last);
            else R2::compute(x, w2);
        }
        if (!OUT_A) {
            if (do_post) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int jj = p.rev_rows ? (int)(((j & 1) << 2) | (j & 2) | ((j >> 2) & 1)) : j;
                    if (p.post_scale != 1) x[j] = gl_mul(x[j], p.post_scale);
                    if (p.post_tab) x[j] = gl_mul(x[j], pow_lookup(p.post_tab, p.post_log, (uint32_t)base_out + lane_out + (uint32_t)((size_t)jj * out_step)));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (p.canon_out) {
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = gl_canon(x[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int jj = p.rev_rows ? (int)(((j & 1) << 2) | (j & 2) | ((j >> 2) & 1)) : j;
                (dst + (size_t)jj * out_step)[lane_out] = x[j];
            }
        } else {
            RL::lds_write(lds, tp, b, rg, x);
            __syncthreads();
#pragma unroll 2
            for (int e = 0; e < 8; e++) {
                int idx = tid + e * nthreads;
                int a = idx & (R - 1), bb = idx >> S;
                gl_t v = lds[a * tp + bb];
                int a_out = p.rev_rows ? (int)bitrev32((uint32_t)a, S) : a;
                size_t off = base_out + (size_t)a_out * p.sa_out + (size_t)bb * p.sb_out;
                if (do_post) {
                    if (p.post_scale != 1) v = gl_mul(v, p.post_scale);
                    if (p.post_tab) v = gl_mul(v, pow_lookup(p.post_tab, p.post_log, off));
                }
                if (p.canon_out) v = gl_canon(v);
                dst[off] = v;
            }
        }
        if (IN_A || OUT_A || NR > 1) __syncthreads();  // LDS is reused by the next column
    }
}

// ------------------------------------------------------------------ whole transforms of 2^9 .. 2^13 points in ONE launch
// The pass kernels take at most 8 stages per launch, so a transform of 2^9 .. 2^15 points was two launches -- and a 2^16-cycle segment
// is mostly such transforms (nine of its twelve tables have <= 2^13 rows; every table's quotient chunks and FRI layers shrink into that
// range): the inverse transform and the LDE of a commitment cost four launches, whatever their size (VERDICT r03 missing #4).  Here a
// workgroup holds one column (LDE: one of the four coset blocks of one column, the coset split of lde_coset_split) in LDS and runs
// all its stages: radix-2 decimation in frequency, one barrier per stage, loose arithmetic, canonical on the way out.  natural_out
// reads the bit-reversed result back through LDS, so both sides of HBM see lane-contiguous accesses.
struct ntt_small_args {
    const gl_t* in;
    gl_t* out;
    size_t cs_in, cs_out;
    uint32_t log_n, natural_out;
    const gl_t* tw;            // stage tables, forward or inverse roots
    gl_t post_scale;           // 1 = none
    const gl_t* post_tab;      // natural_out: out[k] *= table(k) (two-level power table over log_n), or null
    const gl_t* pre_tab_k[4];  // per coset (blockIdx.y): in[i] *= table(i), or null
    size_t out_off_k[4];
};
#define ZKM_WAVE_SYNC()                                      \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
// n / 8 threads, four butterflies per thread and stage.  The stages whose butterflies span 512 words or more exchange across waves
// (a workgroup barrier each: L - 9 of them); after those every wave owns 512 consecutive words and runs the remaining nine stages
// on its own -- a wave's LDS instructions execute in order, no workgroup barrier.  The stage tables (tw[h + j], all of [1, n)) are
// copied to LDS once: a global load per stage was a round trip to L2 between two barriers, eleven times per workgroup.
__global__ __launch_bounds__(1024) void k_ntt_small(ntt_small_args p) {
    extern __shared__ __attribute__((aligned(16))) gl_t lds[];
    const unsigned L = p.log_n, n = 1u << L, tid = threadIdx.x, T = blockDim.x;   // T = n / 8
    gl_t* const twl = lds + n;
    const gl_t* __restrict__ src = p.in + (size_t)blockIdx.x * p.cs_in;
    const gl_t* __restrict__ pre = p.pre_tab_k[blockIdx.y];
    {
        gl_t v[8], w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = src[tid + T * k]; w[k] = p.tw[tid + T * k]; }
        if (pre) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = gl_mul_loose(v[k], pow_lookup(pre, L, tid + T * k));
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { lds[tid + T * k] = v[k]; twl[tid + T * k] = w[k]; }
    }
    __syncthreads();
    auto stage = [&](unsigned s, unsigned b0, unsigned bstep, unsigned base) {
        const unsigned h = 1u << s;
        unsigned idx[4];
        uint64_t u[4], v[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned b = b0 + bstep * k, j = b & (h - 1);
            idx[k] = base + (((b >> s) << (s + 1)) | j);
            u[k] = lds[idx[k]];
            v[k] = lds[idx[k] + h];
            w[k] = twl[h + j];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            lds[idx[k]] = gl_add_rr(u[k], v[k]);
            lds[idx[k] + h] = s ? gl_mul_loose(gl_sub_rr(u[k], v[k]), w[k]) : gl_sub_rr(u[k], v[k]);   // (the last stage's twiddle is 1)
        }
    };
    unsigned s = L;
#pragma unroll 1
    while (s > 9) {
        s--;
        stage(s, tid, T, 0);
        __syncthreads();
    }
    const unsigned lane = tid & 63, base = (tid >> 6) << 9, wstep = T < 64 ? T : 64;   // (transforms below 2^9 points: part of one wave)
#pragma unroll 1
    while (s > 0) {
        s--;
        stage(s, lane, wstep, base);
        ZKM_WAVE_SYNC();
    }
    __syncthreads();
    gl_t* __restrict__ dst = p.out + (size_t)blockIdx.x * p.cs_out + p.out_off_k[blockIdx.y];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const unsigned e = tid + T * k;
        gl_t v = lds[p.natural_out ? bitrev32(e, L) : e];
        if (p.post_scale != 1) v = gl_mul_loose(v, p.post_scale);
        if (p.post_tab) v = gl_mul_loose(v, pow_lookup(p.post_tab, L, e));
        dst[e] = gl_canon(v);
    }
}
static bool ntt_small_ok(unsigned log_n) { return log_n >= 9 && log_n <= 13; }
template <typename K>
static void ntt_allow_big_lds(zkm_ctx* c, K kernel, std::atomic<uint64_t>& done);
static void launch_ntt_small(zkm_ctx* c, const ntt_small_args& a, size_t ncols, unsigned ncoset, const char* name) {
    static std::atomic<uint64_t> lds_ok{0};
    const unsigned n = 1u << a.log_n;
    zkm_prof_scope ps(c, name);
    if (2 * (size_t)n * sizeof(gl_t) > 64 * 1024) ntt_allow_big_lds(c, k_ntt_small, lds_ok);
    hipLaunchKernelGGL(k_ntt_small, dim3((unsigned)ncols, ncoset), dim3(n >> 3), 2 * (size_t)n * sizeof(gl_t), c->stream, a);
    ZKM_HIP_CHECK(hipGetLastError());
}

struct ntt_plan {
    int np;
    int S[4];
};
static ntt_plan make_plan(unsigned L) {
    ntt_plan pl{};
    pl.np = (int)((L + 7) / 8);
    int base = (int)L / pl.np, extra = (int)L % pl.np;
    // the larger radices go to the LAST passes: a strided pass of 2^S rows only has 4096 / 2^S contiguous columns per row, and
    // 128 B row segments (S = 8) reach about half the HBM rate of 256 B ones (S = 7); the last pass is contiguous along rows
    for (int i = 0; i < pl.np; i++) pl.S[i] = base + (i >= pl.np - extra ? 1 : 0);
    // A last pass of exactly 6 stages ends on a full three-stage round whose twiddles are powers of two (compute_pow2): take it
    // whenever the other passes can absorb the remaining stages (3..8 each).
    if (pl.np >= 2) {
        int rest = (int)L - 6, k = pl.np - 1;
        if (rest >= 3 * k && rest <= 8 * k) {
            int b2 = rest / k, e2 = rest % k;
            for (int i = 0; i < k; i++) pl.S[i] = b2 + (i >= k - e2 ? 1 : 0);
            pl.S[k] = 6;
        }
    }
    return pl;
}

// hipFuncSetAttribute is per device: remember which devices have the dynamic-LDS limit of a kernel raised (one bit per device; a
// process may hold contexts on several GPUs, and several host threads may get here at once -- setting it twice is harmless)
template <typename K>
static void ntt_allow_big_lds(zkm_ctx* c, K kernel, std::atomic<uint64_t>& done) {
    const uint64_t bit = (uint64_t)1 << (c->device & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    ZKM_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.fetch_or(bit, std::memory_order_release);
}

template <int S, bool IN_A, bool OUT_A, int PRE, int POST, bool ZP = false, bool PF = true>
static void launch_pass_t(zkm_ctx* c, const ntt_pass_args& a, size_t ntiles) {
    static std::atomic<uint64_t> lds_ok{0};
    ntt_allow_big_lds(c, k_ntt_pass<S, IN_A, OUT_A, PRE, POST, ZP, PF>, lds_ok);
    int T = 1 << a.log_T;
    size_t shmem = ((size_t)1 << S) * a.tp * sizeof(gl_t);
    dim3 grid((unsigned)(ntiles * (a.ncoset > 1 ? a.ncoset : 1)), (a.ncols + a.cpb - 1) / a.cpb), block((unsigned)(((1 << S) >> 3) * T));
    hipLaunchKernelGGL((k_ntt_pass<S, IN_A, OUT_A, PRE, POST, ZP, PF>), grid, block, shmem, c->stream, a);
}

template <int S>
static void launch_pass_s(zkm_ctx* c, const ntt_pass_args& a, size_t ntiles) {
    const bool pre = a.pre_tab != nullptr || a.ncoset > 1, post = a.post_scale != 1 || a.post_tab != nullptr;
    const bool zp = a.n_in != ~(size_t)0;  // the input is shorter than the transform (zero-padded LDE): bounds-checked loads
    if (!a.in_contig_a && !a.out_contig_a) {
        if (post) throw std::runtime_error("ntt pass: strided pass cannot post-scale");
        if (a.ncoset > 1) launch_pass_t<S, false, false, 3, 0, false, false>(c, a, ntiles);
        else if (pre && zp) launch_pass_t<S, false, false, 1, 0, true>(c, a, ntiles);
        else if (pre) launch_pass_t<S, false, false, 1, 0>(c, a, ntiles);
        else if (zp) launch_pass_t<S, false, false, 0, 0, true>(c, a, ntiles);
        else launch_pass_t<S, false, false, 0, 0>(c, a, ntiles);
    } else if (a.in_contig_a && a.out_contig_a) {
        if (zp) launch_pass_t<S, true, true, 2, 2, true>(c, a, ntiles);
        else launch_pass_t<S, true, true, 2, 2>(c, a, ntiles);
    } else if (a.in_contig_a && !a.out_contig_a) {
        if (pre) throw std::runtime_error("ntt pass: transposing pass cannot pre-scale");
        if (post) launch_pass_t<S, true, false, 0, 1>(c, a, ntiles);
        else launch_pass_t<S, true, false, 0, 0>(c, a, ntiles);
    } else {
        throw std::runtime_error("ntt pass: unsupported layout combination");
    }
}

static void launch_pass(zkm_ctx* c, int S, ntt_pass_args a, size_t ntiles, const char* name) {
    // columns per workgroup: keep >= ~2048 workgroups in flight, amortise twiddle loads
    size_t want = ((size_t)a.ncols * ntiles) / 2048;
    a.cpb = (uint32_t)(want < 1 ? 1 : (want > 16 ? 16 : want));
    if (a.cpb > a.ncols) a.cpb = a.ncols;
    zkm_prof_scope ps(c, name);
    switch (S) {
        case 3: launch_pass_s<3>(c, a, ntiles); break;
        case 4: launch_pass_s<4>(c, a, ntiles); break;
        case 5: launch_pass_s<5>(c, a, ntiles); break;
        case 6: launch_pass_s<6>(c, a, ntiles); break;
        case 7: launch_pass_s<7>(c, a, ntiles); break;
        case 8: launch_pass_s<8>(c, a, ntiles); break;
        default: throw std::runtime_error("ntt pass: unsupported stage count");
    }
    ZKM_HIP_CHECK(hipGetLastError());
}

static uint32_t pick_log_T(int S, size_t bdim) {
    constexpr size_t TILE = 2048;  // elements per workgroup tile (threads = tile / 8): measured best of 512 .. 4096 (profiles/r02_ubench_strided_tiles.txt)
    uint32_t lt = 6;  // 64 lanes along the tile's column dimension
    while (lt > 0 && (((size_t)1 << S) << lt) > TILE) lt--;
    while (lt > 0 && ((size_t)1 << lt) > bdim) lt--;
    return lt;
}

// natural -> bit-reversed, `np` passes over `out`; pass 1 reads `in` (zero-padded beyond n_in, optionally
// scaled by shift^i), later passes are in place.
static void ntt_dif_bitrev_fast(zkm_ctx* c, const gl_t* in, size_t cs_in, gl_t* out, size_t cs_out, size_t ncols, unsigned L,
                                bool inverse, size_t n_in, unsigned log_n_in, uint64_t shift) {
    c->ensure_twiddles(L);
    ntt_plan pl = make_plan(L);
    const gl_t* tw = inverse ? c->tw.inv : c->tw.fwd;
    unsigned m = L;
    for (int p = 0; p < pl.np; p++) {
        int S = pl.S[p];
        m -= S;
        ntt_pass_args a{};
        a.in = p == 0 ? in : out; a.out = out;
        a.cs_in = p == 0 ? cs_in : cs_out; a.cs_out = cs_out;
        a.ncols = (uint32_t)ncols;
        a.tw = tw; a.m = m; a.post_scale = 1; a.n_in = ~(size_t)0;
        if (p == 0) {
            a.n_in = n_in;
            if (shift > 1) { a.pre_tab = c->pow_table(shift, log_n_in); a.pre_log = log_n_in; }
            a.zero_padded = m > 0 && S >= 3 && n_in * 4 == ((size_t)1 << L);
        }
        size_t ntiles;
        if (m > 0) {  // strided pass: rows at stride 2^m, tile columns contiguous
            a.log_T = pick_log_T(S, (size_t)1 << m);
            a.tp = 1u << a.log_T;
            a.n_lo = (uint32_t)(((size_t)1 << m) >> a.log_T);
            a.bi_hi = a.bo_hi = (size_t)1 << (S + m);
            a.bi_lo = a.bo_lo = (size_t)1 << a.log_T;
            a.sa_in = a.sa_out = (size_t)1 << m; a.sb_in = a.sb_out = 1;
            ntiles = ((size_t)1 << (L - S - m)) * a.n_lo;
        } else {  // last pass: rows contiguous, tile columns = consecutive chunks of R
            size_t nchunks = (size_t)1 << (L - S);
            a.log_T = pick_log_T(S, nchunks);
            a.tp = (1u << a.log_T) + 1;
            a.n_lo = (uint32_t)(nchunks >> a.log_T);
            a.bi_lo = a.bo_lo = ((size_t)1 << a.log_T) << S;
            a.sa_in = a.sa_out = 1; a.sb_in = a.sb_out = (size_t)1 << S;
            a.in_contig_a = a.out_contig_a = 1;
            ntiles = a.n_lo;
        }
        a.canon_out = m == 0;
        a.pow2_last = m == 0 ? (inverse ? 2u : 1u) : 0u;
        launch_pass(c, S, a, ntiles, m > 0 ? "ntt_pass_strided" : "ntt_pass_contig");
    }
}

// natural -> natural (np <= 3).  in may equal scratch.  forward: in * shift^i first; inverse: scaled by
// n^-1 * shift^-k on the way out.
static void ntt_natural_fast(zkm_ctx* c, const gl_t* in, size_t cs_in, gl_t* scratch, size_t cs_s, gl_t* out, size_t cs_out,
                             size_t ncols, unsigned L, bool inverse, uint64_t shift) {
    c->ensure_twiddles(L);
    ntt_plan pl = make_plan(L);
    const gl_t* tw = inverse ? c->tw.inv : c->tw.fwd;
    size_t n = (size_t)1 << L;
    const gl_t* pre = (!inverse && shift > 1) ? c->pow_table(shift, L) : nullptr;
    const gl_t* post = (inverse && shift > 1) ? c->pow_table(gl_inv(shift), L) : nullptr;
    gl_t post_scale = inverse ? gl_inv((gl_t)(n % GL_P)) : 1;
    if (ntt_small_ok(L) && c->small_ntt) {   // one launch, a column per workgroup (k_ntt_small)
        ntt_small_args a{};
        a.in = in; a.out = out; a.cs_in = cs_in; a.cs_out = cs_out; a.log_n = L; a.natural_out = 1; a.tw = tw;
        a.post_scale = post_scale; a.post_tab = post; a.pre_tab_k[0] = pre;
        launch_ntt_small(c, a, ncols, 1, "ntt_small");
        return;
    }
    int np = pl.np;
    size_t N1 = (size_t)1 << pl.S[0];
    unsigned m = L;
    for (int p = 0; p < np; p++) {
        int S = pl.S[p];
        m -= S;
        bool last = p == np - 1;
        ntt_pass_args a{};
        a.in = p == 0 ? in : scratch; a.cs_in = p == 0 ? cs_in : cs_s;
        a.out = last ? out : scratch; a.cs_out = last ? cs_out : cs_s;
        a.ncols = (uint32_t)ncols; a.tw = tw; a.m = m; a.post_scale = 1; a.n_in = ~(size_t)0; a.rev_rows = 1;
        if (p == 0 && pre) { a.pre_tab = pre; a.pre_log = L; }
        if (last) { a.post_scale = post_scale; a.post_tab = post; a.post_log = L; }
        size_t ntiles;
        if (!last) {
            a.log_T = pick_log_T(S, (size_t)1 << m);
            a.tp = 1u << a.log_T;
            a.n_lo = (uint32_t)(((size_t)1 << m) >> a.log_T);
            a.bi_hi = a.bo_hi = (size_t)1 << (S + m);
            a.bi_lo = a.bo_lo = (size_t)1 << a.log_T;
            a.sa_in = a.sa_out = (size_t)1 << m; a.sb_in = a.sb_out = 1;
            ntiles = ((size_t)1 << (L - S - m)) * a.n_lo;
        } else if (np == 1) {
            a.log_T = 0; a.tp = 2; a.n_lo = 1;
            a.sa_in = a.sa_out = 1; a.in_contig_a = a.out_contig_a = 1;
            ntiles = 1;
        } else {
            // transposing last pass: rows = contiguous d_last, tile columns = consecutive k1 (stride n/N1),
            // t_hi = middle digit (np == 3).  Output index k = k1 + N1*(k_mid + N_mid*k_last).
            size_t Nl = (size_t)1 << S, M1 = n >> pl.S[0], Nmid = np == 3 ? (size_t)1 << pl.S[1] : 1;
            a.log_T = pick_log_T(S, N1);
            a.tp = (1u << a.log_T) + 1;
            a.n_lo = (uint32_t)(N1 >> a.log_T);
            a.bi_hi = Nl; a.bi_lo = ((size_t)1 << a.log_T) * M1;
            a.sa_in = 1; a.sb_in = M1; a.in_contig_a = 1;
            a.bo_hi = N1; a.bo_lo = (size_t)1 << a.log_T;
            a.sa_out = N1 * Nmid; a.sb_out = 1; a.out_contig_a = 0;
            ntiles = Nmid * a.n_lo;
        }
        a.canon_out = last;
        a.pow2_last = last ? (inverse ? 2u : 1u) : 0u;
        launch_pass(c, S, a, ntiles, last ? "ntt_pass_transpose" : "ntt_pass_strided");
    }
}

// ------------------------------------------------------------------ large contiguous pass
// The LAST pass of a bit-reversed-output transform: SA + 6 radix-2 DIF stages on blocks of E = 2^(SA+6) CONTIGUOUS elements (64 KB
// for SA = 7), one block per workgroup iteration, 8 elements per thread.  Phase 1 treats the block as [2^SA rows][64 columns] and
// runs the SA upper stages like a strided pass (lanes along the 64 contiguous elements, rows at stride 64); phase 2 runs the six
// lower stages on the 64-element rows themselves with the lanes along the 2^SA rows (the LDS image [a * 65 + b] is conflict-free
// in both directions: 65 words = 130 banks == 2 mod 64).  Global accesses are word tid + NT * j on the way in and, after a last
// pass through LDS, on the way out: 512 B per wave instruction.  Phase-2 twiddles only depend on the row group (tid >> SA), which
// is wave-uniform for SA >= 6: they live in SGPRs.  The next block is prefetched while the current one is transformed.
struct ntt_big_args {
    gl_t* data;
    size_t cs;            // column stride (elements)
    uint32_t ncols;
    uint32_t blocks_per_col;
    uint32_t canon_out;
    const gl_t* tw;
    gl_t scale;           // k_ntt_blk12<INV = true>: n^-1, applied on the way out
    uint32_t s1;          // k_ntt_blk12<.., PERM = true>: stages of the strided pass before this one (the digit layout, zkm_internal.h)
};

template <int S, int K>
struct ntt_round_t : ntt_round<S, K> {  // the same row / twiddle scheme with explicit LDS strides
    using B = ntt_round<S, K>;
    __device__ static __forceinline__ void lds_read(const gl_t* lds, int sr, int sb, int b, int rg, gl_t (&x)[8]) {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = lds[B::row(rg, j) * sr + b * sb];
    }
    __device__ static __forceinline__ void lds_write(gl_t* lds, int sr, int sb, int b, int rg, const gl_t (&x)[8]) {
#pragma unroll
        for (int j = 0; j < 8; j++) lds[B::row(rg, j) * sr + b * sb] = x[j];
    }
};

template <int SA>
__global__ __launch_bounds__((1 << SA) * 8) void k_ntt_big(ntt_big_args p) {
    extern __shared__ __attribute__((aligned(16))) gl_t lds[];
    constexpr int RA = 1 << SA, NT = RA * 8, LP = 65, NR1 = (SA + 2) / 3;
    constexpr bool UNI = SA >= 6;  // tid >> SA is uniform over a wave
    using A0 = ntt_round_t<SA, 0>;
    using A1 = ntt_round_t<SA, (NR1 > 1 ? 1 : 0)>;
    using A2 = ntt_round_t<SA, (NR1 > 2 ? 2 : 0)>;
    using B0 = ntt_round_t<6, 0>;
    using B1 = ntt_round_t<6, 1>;
    const int tid = threadIdx.x;
    const int b = tid & 63, rg = tid >> 6;                  // phase 1: column b, row group rg
    const int a2 = tid & (RA - 1);                          // phase 2: row a2 of the phase-1 output ...
    const int rg2 = UNI ? __builtin_amdgcn_readfirstlane(tid >> SA) : (tid >> SA);  // ... group rg2 of its 64 elements

    gl_t wa0[7], wa1[7], wa2[7], wb0[7];
    A0::load_tw(wa0, p.tw, rg, 6, (size_t)b);
    if (NR1 > 1) A1::load_tw(wa1, p.tw, rg, 6, (size_t)b);
    if (NR1 > 2) A2::load_tw(wa2, p.tw, rg, 6, (size_t)b);
    B0::load_tw(wb0, p.tw, rg2, 0, 0);   // (the last round's twiddles are powers of two: B1::compute_pow2)

    const uint32_t total = p.ncols * p.blocks_per_col;
    auto block_ptr = [&](uint32_t blk) { return p.data + (size_t)(blk / p.blocks_per_col) * p.cs + ((size_t)(blk % p.blocks_per_col) << (SA + 6)); };
    gl_t nx[8];
    uint32_t blk = blockIdx.x;
    if (blk < total) {
        const gl_t* src = block_ptr(blk);
#pragma unroll
        for (int j = 0; j < 8; j++) nx[j] = src[tid + NT * j];
    }
    for (; blk < total; blk += gridDim.x) {
        gl_t* const dst = block_ptr(blk);
        gl_t x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = nx[j];          // rows rg + (RA / 8) j == A0::row(rg, j), column b
        if (blk + gridDim.x < total) {
            const gl_t* src = block_ptr(blk + gridDim.x);
#pragma unroll
            for (int j = 0; j < 8; j++) nx[j] = src[tid + NT * j];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 1: SA stages along the rows, lanes along b
        A0::compute(x, wa0);
        if (NR1 > 1) {
            A0::lds_write(lds, LP, 1, b, rg, x);
            __syncthreads();
            A1::lds_read(lds, LP, 1, b, rg, x);
            A1::compute(x, wa1);
        }
        if (NR1 > 2) {
            __syncthreads();
            A1::lds_write(lds, LP, 1, b, rg, x);
            __syncthreads();
            A2::lds_read(lds, LP, 1, b, rg, x);
            A2::compute(x, wa2);
        }
        using AL = ntt_round_t<SA, NR1 - 1>;
        if (NR1 > 1) __syncthreads();
        AL::lds_write(lds, LP, 1, b, rg, x);
        __syncthreads();
        // ---- phase 2: the six stages inside each 64-element row, lanes along the rows
        B0::lds_read(lds, 1, LP, a2, rg2, x);
        B0::compute(x, wb0);
        __syncthreads();
        B0::lds_write(lds, 1, LP, a2, rg2, x);
        __syncthreads();
        B1::lds_read(lds, 1, LP, a2, rg2, x);
        B1::template compute_pow2<false>(x);
        __syncthreads();
        B1::lds_write(lds, 1, LP, a2, rg2, x);
        __syncthreads();
        // ---- out: word p of the block sits at lds[p + (p >> 6)]
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pidx = tid + NT * j;
            gl_t v = lds[pidx + (pidx >> 6)];
            if (p.canon_out) v = gl_canon(v);
            dst[pidx] = v;
        }
        __syncthreads();  // LDS is reused by the next block
    }
}

// ---- the 2^12-element block with ONE exchange across waves: 3 + 9 stages.
// Round 1 (stages 11, 10, 9) works on the words a thread loads anyway (tid + 512 j: 512 B per wave instruction); what is left
// are eight independent 2^9-point problems, and 2^9 words are exactly what ONE wave holds (64 lanes x 8 registers): after the
// single cross-wave exchange every wave owns a sub-block and runs stages 8 .. 0 on its own -- rounds of three stages with the
// register-resident bits (8,7,6), (5,4,3), (2,1,0), exchanged through the wave's own LDS region, which needs no workgroup
// barrier (a wave's LDS instructions execute in order); the last round's twiddles are powers of two.  Word t of a sub-block
// lives at A(t) = (t + 8 (t >> 6)) ^ ((t >> 3) & 7): conflict-free for all four access patterns, reads (32-lane groups, 64-bit
// banks mod 32) and writes (16-lane groups, mod 16) alike.  Four LDS round trips and two workgroup barriers per block (one
// before the round-1 stores: the previous block's readers must be done; one after them) against five and nine in k_ntt_big<6>.
// 78 VGPRs and 41 KB of LDS: three workgroups per CU, whose waves cover each other's memory latency -- measured better than two
// workgroups with a register prefetch of the next block and two alternating LDS images (5.17 vs 5.49 ms for 262 x 2^22).
// (ZKM_WAVE_SYNC: defined above k_ntt_small)

// INV: inverse roots (the table passed in p.tw, the negated powers of two of the last round) and the scale n^-1 on the way out.
// PERM: the block is stored in the DIGIT order of the coefficient layout (zkm_coeff_exponent, zkm_internal.h) instead of in place:
// word q of the block (holding the output whose index has the high twelve bits rev12(q)) goes to position
// rev_(12-s1)(q >> s1) * 2^s1 + rev_s1(q mod 2^s1), so that the 2^s1 coefficients t_lo + 2^12 t_hi of one t_lo are CONTIGUOUS in t_hi
// -- the runs k_lde_upper reads its tile from.  Lanes store consecutive positions (512 B per wave instruction) and fetch the word
// that belongs there from the LDS image (any wave's sub-block: one more workgroup barrier).
template <bool INV, bool PERM>
__global__ __launch_bounds__(512, 6) void k_ntt_blk12(ntt_big_args p) {
    extern __shared__ __attribute__((aligned(16))) gl_t lds[];
    constexpr int NT = 512, SB = 576;
    using R = ntt_round<3, 0>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto A = [](int t) { return (t + ((t >> 6) << 3)) ^ ((t >> 3) & 7); };
    const int a1 = A(tid);                                        // round-1 store: word tid of sub-block j

    // round-1 twiddles (one set per thread) stay in registers; those of rounds 2 and 3 only depend on the lane (64 x 7 and 8 x 7
    // words) and are re-read from LDS when needed -- 28 VGPRs
    // (all three in the FACTORED form: the powers W^1 .. W^7 of the round's one twiddle, ntt_round::compute_fact)
    gl_t w1[7];
    {
        gl_t w[7];
        R::load_tw(w, p.tw, 0, 9, (size_t)tid);
        R::factored_tw(w1, w);
    }
    gl_t* const twl = lds + 8 * SB;
    if (tid < 64) {
        gl_t w[7], f[7];
        R::load_tw(w, p.tw, 0, 6, (size_t)tid);
        R::factored_tw(f, w);
#pragma unroll
        for (int k = 0; k < 7; k++) twl[k * 64 + tid] = f[k];
    } else if (tid < 72) {
        gl_t w[7], f[7];
        R::load_tw(w, p.tw, 0, 3, (size_t)(tid - 64));
        R::factored_tw(f, w);
        // INV: the scale n^-1 rides on the third round's factors (slot 0, which has none, is multiplied there by hand): one product per
        // eight words instead of eight on the way out
#pragma unroll
        for (int k = 0; k < 7; k++) twl[448 + k * 8 + (tid - 64)] = INV ? gl_mul(f[k], p.scale) : f[k];
    }
    __syncthreads();

    const uint32_t total = p.ncols * p.blocks_per_col;
    auto block_ptr = [&](uint32_t blk) { return p.data + (size_t)(blk / p.blocks_per_col) * p.cs + ((size_t)(blk % p.blocks_per_col) << 12); };
    for (uint32_t blk = blockIdx.x; blk < total; blk += gridDim.x) {
        gl_t* const dst = block_ptr(blk);
        gl_t x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = dst[tid + NT * j];
        R::template compute_fact<INV>(x, w1);                     // stages 11, 10, 9
        __syncthreads();                                               // the previous block's readers are done with the image
#pragma unroll
        for (int j = 0; j < 8; j++) lds[j * SB + a1] = x[j];
        __syncthreads();
        // per-lane LDS offsets of the wave-local rounds, recomputed per block from an opaque copy of the lane id (kept live across
        // the loop they would cost ~20 VGPRs)
        int l = lane;
        asm volatile("" : "+v"(l));
        const int l7 = l & 7, h3 = l >> 3;
        const int a2 = A(l);                                      // rounds 2 / out: words l + 64 j at a2 + 72 j
        const int a3 = 72 * h3;                                   // round 3: words 64 h3 + 8 j + l7 at a3 + 8 j + (l7 ^ j)
        const int a4 = 8 * l + 8 * h3;                            // round 4: words 8 l + j at a4 + (j ^ l7)
        gl_t* const sub = lds + wv * SB;
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = sub[a2 + 72 * j];
        gl_t w[7];
#pragma unroll
        for (int k = 0; k < 7; k++) w[k] = twl[k * 64 + l];
        R::template compute_fact<INV>(x, w);                      // stages 8, 7, 6
#pragma unroll
        for (int j = 0; j < 8; j++) sub[a2 + 72 * j] = x[j];
        ZKM_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = sub[a3 + 8 * j + (l7 ^ j)];
#pragma unroll
        for (int k = 0; k < 7; k++) w[k] = twl[448 + k * 8 + l7];
        R::template compute_fact<INV>(x, w);                      // stages 5, 4, 3
        if (INV) x[0] = gl_mul_loose(x[0], p.scale);
#pragma unroll
        for (int j = 0; j < 8; j++) sub[a3 + 8 * j + (l7 ^ j)] = x[j];
        ZKM_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = sub[a4 + (l7 ^ j)];
        R::template compute_pow2<INV>(x);                         // stages 2, 1, 0
        if (INV || p.canon_out) {
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = gl_canon(x[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) sub[a4 + (l7 ^ j)] = x[j];
        if (PERM) {
            __syncthreads();                                      // every sub-block of the image is final
            const unsigned s1 = p.s1, cb = 12 - s1;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned pos = tid + NT * j;
                const unsigned q = (bitrev32(pos >> s1, cb) << s1) | bitrev32(pos & ((1u << s1) - 1), s1);
                dst[pos] = lds[(q >> 9) * SB + A(q & 511)];
            }
        } else {
            ZKM_WAVE_SYNC();
            gl_t* const out = dst + (wv << 9);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const gl_t v = sub[a2 + 72 * j];
                out[l + 64 * j] = v;
            }
        }
    }
}

template <bool INV, bool PERM>
static void launch_blk12(zkm_ctx* c, const ntt_big_args& a) {
    static std::atomic<uint64_t> lds_ok{0};
    const size_t shmem = (8 * 576 + 7 * 64 + 7 * 8) * sizeof(gl_t);
    ntt_allow_big_lds(c, k_ntt_blk12<INV, PERM>, lds_ok);
    uint32_t total = a.ncols * a.blocks_per_col;
    uint32_t grid = total < (uint32_t)c->num_cus * 12 ? total : (uint32_t)c->num_cus * 12;
    hipLaunchKernelGGL((k_ntt_blk12<INV, PERM>), dim3(grid), dim3(512), shmem, c->stream, a);
}

template <int SA>
static void launch_big_t(zkm_ctx* c, const ntt_big_args& a) {
    static std::atomic<uint64_t> lds_ok{0};
    size_t shmem = ((size_t)1 << SA) * 65 * sizeof(gl_t);
    ntt_allow_big_lds(c, k_ntt_big<SA>, lds_ok);
    uint32_t total = a.ncols * a.blocks_per_col;
    uint32_t grid = total < (uint32_t)c->num_cus * 8 ? total : (uint32_t)c->num_cus * 8;
    hipLaunchKernelGGL((k_ntt_big<SA>), dim3(grid), dim3((1u << SA) * 8), shmem, c->stream, a);
}

// S2 = 11, 12, 13 lower stages of every contiguous 2^S2 block of `ncols` columns of length 2^L, in place
static void ntt_big_pass(zkm_ctx* c, gl_t* data, size_t cs, size_t ncols, unsigned L, int S2, const gl_t* tw) {
    ntt_big_args a{data, cs, (uint32_t)ncols, (uint32_t)(((size_t)1 << L) >> S2), 1u, tw, 1, 0};
    zkm_prof_scope ps(c, "ntt_pass_big");
    switch (S2) {
        case 11: launch_big_t<5>(c, a); break;
        case 12: launch_blk12<false, false>(c, a); break;
        case 13: launch_big_t<7>(c, a); break;
        default: throw std::runtime_error("ntt big pass: unsupported stage count");
    }
    ZKM_HIP_CHECK(hipGetLastError());
}

// ---- Pass A of the coset-split LDE with BLOCK twiddles (stages S = 6, 7, 8 of a 2^12-column tile).
// The decimation-in-frequency recursion with a geometric weight r^t carried implicitly instead of multiplied in:
//   D(f, N, r)[i] = sum_t f(t) r^t w_N^(i t);   split t = t' + (N/2) e, i = 2 i' + eps, W = r^(N/2):
//   D[2 i']     = D(f(t') + W f(t' + N/2), N/2, r)[i'],      D[2 i' + 1] = D(f(t') - W f(t' + N/2), N/2, r w_N)[i'].
// Every butterfly is (u, v) -> (u + W v, u - W v) with ONE W per block of the stage: block q of stage k (2^k blocks) has
//   W(k, q) = shift^(n / 2^(k+1)) * w_(2^(k+1))^bitrev_k(q),
// which only depends on the ROW bits above the stage -- uniform over a tile's columns, so the twiddles of the first two rounds
// are wave-uniform and live in SGPRs (the plain DIF form keeps 21 lane-dependent twiddles = 42 VGPRs and the factored pre-scale
// 16 more, which held the kernel to one workgroup per CU).  The coset shift is inside W, so there is no pre-scale; what the
// plain form multiplies in stage by stage is applied once at the end: row p (holding output residue i_lo = bitrev_S(p)) and
// column t_lo get  (shift w_n^i_lo)^t_lo = shift^t_lo * w_n^(i_lo t_lo)  -- 8 factors per thread, fixed for the tile, kept in
// registers across the polynomial columns.  Same number of products as before (32 butterflies + 8 per column and thread), the
// same field elements on the way out (pass B is unchanged), twice the occupancy.
struct lde_upper_args {
    const gl_t* in;
    gl_t* out;
    size_t cs_in, cs_out;
    uint32_t ncols, cpb, log_n, S2;
    const gl_t* tw;            // forward table: w_n^e = tw[n/2 + e] for e < n/2
    const gl_t* ct;            // [4 cosets][2^S]: W(k, q) at 2^k + q
    const gl_t* pre_tab_k[4];  // shift_k^t (two-level power table over n)
    size_t out_off_k[4];
};

template <int S, int K>
struct ct_round : ntt_round<S, K> {
    using B = ntt_round<S, K>;
    // W of local stage q + jb for register index j: heap index 2^(S-1-q-jb) + ((rg >> q) << (2 - jb)) + (j >> (jb + 1))
    template <bool UNI>
    __device__ static __forceinline__ void load(gl_t (&w)[7], const gl_t* __restrict__ tab, int rg) {
        int rgq = rg >> B::q;
        if (UNI) rgq = __builtin_amdgcn_readfirstlane(rgq);
#pragma unroll
        for (int jb = 2; jb >= 0; jb--) {
            if (jb > B::top) continue;
            const int base = jb == 2 ? 0 : (jb == 1 ? 1 : 3);
#pragma unroll
            for (int jh = 0; jh < (4 >> jb); jh++) w[base + jh] = tab[(1 << (S - 1 - B::q - jb)) + (rgq << (2 - jb)) + jh];
        }
    }
    __device__ static __forceinline__ void bfly(gl_t& u, gl_t& v, gl_t w) {
        const uint64_t t = gl_mul_loose(v, w);
        v = gl_sub_rr(u, t);
        u = gl_add_rr(u, t);
    }
    // A full round with the block twiddles FACTORED OUT (radix-8 decimation in time).  The seven W of a round are powers of one
    // number up to shifts: with rho = W(k + 2, 4Q) they are  w[3..6] = rho {1, w_4, w_8, w_8^3},  w[1..2] = rho^2 {1, w_4},  w[0] = rho^4
    // (W(k, q)^2 = W(k - 1, q / 2) and the bit-reversed exponent of an odd block adds half a turn of the next root), so with
    // z_j = rho^j x_j every rho cancels out of the butterflies: the round is  z = x .* rho^(0..7)  (7 general products) followed by
    // an 8-point transform whose twiddles are w_4 = 2^48, w_8 = 2^24, w_8^3 = 2^72 -- 5 shift-multiplies instead of 12 general
    // products.  The same field elements as compute().  f[j - 1] = rho^j (factored_tw).
    template <int E>
    __device__ static __forceinline__ void bfly_pow2(gl_t& u, gl_t& v) {
        const uint64_t t = E ? gl_mul_pow2<(E ? E : 1)>(v) : v;
        v = gl_sub_rr(u, t);
        u = gl_add_rr(u, t);
    }
    __device__ static __forceinline__ void compute_fact(gl_t (&x)[8], const gl_t (&f)[7]) {
        static_assert(B::top == 2, "compute_fact: a full three-stage round");
        x[1] = gl_mul_loose(x[1], f[0]); x[2] = gl_mul_loose(x[2], f[1]);
        __builtin_amdgcn_sched_barrier(0);
        x[3] = gl_mul_loose(x[3], f[2]); x[4] = gl_mul_loose(x[4], f[3]);
        __builtin_amdgcn_sched_barrier(0);
        x[5] = gl_mul_loose(x[5], f[4]); x[6] = gl_mul_loose(x[6], f[5]);
        __builtin_amdgcn_sched_barrier(0);
        x[7] = gl_mul_loose(x[7], f[6]);
        __builtin_amdgcn_sched_barrier(0);
        bfly_pow2<0>(x[0], x[4]); bfly_pow2<0>(x[1], x[5]);
        __builtin_amdgcn_sched_barrier(0);
        bfly_pow2<0>(x[2], x[6]); bfly_pow2<0>(x[3], x[7]);
        __builtin_amdgcn_sched_barrier(0);
        bfly_pow2<0>(x[0], x[2]); bfly_pow2<0>(x[1], x[3]);
        __builtin_amdgcn_sched_barrier(0);
        bfly_pow2<48>(x[4], x[6]); bfly_pow2<48>(x[5], x[7]);
        __builtin_amdgcn_sched_barrier(0);
        bfly_pow2<0>(x[0], x[1]); bfly_pow2<48>(x[2], x[3]);
        __builtin_amdgcn_sched_barrier(0);
        bfly_pow2<24>(x[4], x[5]); bfly_pow2<72>(x[6], x[7]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // block twiddles of a full round (load: w[3] = rho, w[1] = rho^2, w[0] = rho^4) -> rho^1 .. rho^7
    // UNI: the inputs are wave-uniform (SGPRs): so are the products -- back to SGPRs they go
    template <bool UNI>
    __device__ static __forceinline__ void factored_tw(gl_t (&f)[7], const gl_t (&w)[7]) {
        f[0] = w[3]; f[1] = w[1]; f[3] = w[0];
        f[2] = gl_mul(w[3], w[1]);
        f[4] = gl_mul(w[3], w[0]);
        f[5] = gl_mul(w[1], w[0]);
        f[6] = gl_mul(f[2], w[0]);
        if (UNI) {
#pragma unroll
            for (int k : {2, 4, 5, 6})
                f[k] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(f[k] >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)f[k]);
        }
    }
    __device__ static __forceinline__ void compute(gl_t (&x)[8], const gl_t (&w)[7]) {
        if (B::top >= 2) {
            bfly(x[0], x[4], w[0]); bfly(x[1], x[5], w[0]);
            __builtin_amdgcn_sched_barrier(0);
            bfly(x[2], x[6], w[0]); bfly(x[3], x[7], w[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (B::top >= 1) {
            bfly(x[0], x[2], w[1]); bfly(x[1], x[3], w[1]);
            __builtin_amdgcn_sched_barrier(0);
            bfly(x[4], x[6], w[2]); bfly(x[5], x[7], w[2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        bfly(x[0], x[1], w[3]); bfly(x[2], x[3], w[4]);
        __builtin_amdgcn_sched_barrier(0);
        bfly(x[4], x[5], w[5]); bfly(x[6], x[7], w[6]);
        __builtin_amdgcn_sched_barrier(0);
    }
};

// RUNS: the coefficients come in the digit layout of the two-pass inverse transform (zkm_coeff_exponent): the 2^S coefficients
// t_lo + 2^S2 t_hi of column t_lo are a contiguous run, in t_hi order, at rev_S(t_lo mod 2^S) * 2^12 + (t_lo >> S) * 2^S -- the tile is
// 2^(12-S) runs instead of 2^S row segments.  The first round then has its lanes along the ROWS of a run (thread = (row group rgA,
// column bA), rgA fastest): each load instruction reads 2^(S-3) consecutive words of a run, and the block twiddles of the first round
// depend on no row bit a thread holds, so they stay uniform.  The first exchange goes through a column-major image ([column][row],
// stride 2^S + 1) that the row-fastest lanes write and the column-fastest lanes of the later rounds read without bank conflicts.
template <int S, bool RUNS>
__global__ __launch_bounds__(512, 4) void k_lde_upper(lde_upper_args p) {
    extern __shared__ __attribute__((aligned(16))) gl_t lds[];
    constexpr int R = 1 << S, NR = (S + 2) / 3, LT = 12 - S, T = 1 << LT;
    using R0 = ct_round<S, 0>;
    using R1 = ct_round<S, (NR > 1 ? 1 : 0)>;
    using R2 = ct_round<S, (NR > 2 ? 2 : 0)>;
    using RL = ct_round<S, NR - 1>;  // last round: q == 0, rows (rg << 3) | j
    const int tid = threadIdx.x, b = tid & (T - 1), rg = tid >> LT;
    // the four cosets of a tile on one XCD (ids 8 apart), as in k_ntt_pass
    const uint32_t g = blockIdx.x >> 3, coset = g & 3, tile = (g >> 2) * 8 + (blockIdx.x & 7);
    const uint32_t t_lo = (tile << LT) + b;                      // column of the 2^S2-wide row this lane works on
    const gl_t* const ct = p.ct + ((size_t)coset << S);

    constexpr int RG = R >> 3, IMG = R * T + T;                  // row groups; words per LDS image (the column-major one is padded)
    const int rgA = RUNS ? tid & (RG - 1) : rg, bA = RUNS ? tid / RG : b;   // first-round thread mapping
    gl_t w0[7], w1[7], w2[7];
    R0::template load<(RUNS || LT + R0::q >= 6)>(w0, ct, rgA);
    if (NR > 1) R1::template load<(LT + R1::q >= 6)>(w1, ct, rg);
    if (NR > 2) R2::template load<(LT + R2::q >= 6)>(w2, ct, rg);
    // full rounds run in the factored form (ct_round::compute_fact): their twiddle sets become the powers rho^1 .. rho^7
    constexpr bool F0 = R0::top == 2, F1 = NR > 1 && R1::top == 2, F2 = NR > 2 && R2::top == 2;
    if (F0) { gl_t f[7]; R0::template factored_tw<(RUNS || LT + R0::q >= 6)>(f, w0);
#pragma unroll
        for (int k = 0; k < 7; k++) w0[k] = f[k]; }
    if (F1) { gl_t f[7]; R1::template factored_tw<(LT + R1::q >= 6)>(f, w1);
#pragma unroll
        for (int k = 0; k < 7; k++) w1[k] = f[k]; }
    if (F2) { gl_t f[7]; R2::template factored_tw<(LT + R2::q >= 6)>(f, w2);
#pragma unroll
        for (int k = 0; k < 7; k++) w2[k] = f[k]; }

    // (shift w_n^i_lo)^t_lo for the 8 rows this thread stores
    gl_t post[8];
    {
        const gl_t Bv = pow_lookup(p.pre_tab_k[coset], p.log_n, t_lo);
        const uint64_t half = (uint64_t)1 << (p.log_n - 1);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t e = (uint64_t)bitrev32((uint32_t)RL::row(rg, j), S) * t_lo;   // < 2^S * 2^S2 = n
            gl_t wv = p.tw[half + (e & (half - 1))];
            if (e >= half) wv = GL_P - wv;                        // w_n^(n/2) = -1
            post[j] = gl_mul(Bv, wv);
        }
    }

    const size_t sa = (size_t)1 << p.S2;
    // rows rg + j R/8 (== R0::row(rg, j)) of column t_lo
    const uint32_t t_loA = (tile << LT) + bA;
    const uint32_t lane_in = RUNS ? ((bitrev32(t_loA & (R - 1), S) << 12) | ((t_loA >> S) << S)) + (uint32_t)rgA : (uint32_t)((size_t)rg * sa) + t_lo;
    const size_t in_step = RUNS ? (size_t)(R >> 3) : (size_t)(R >> 3) * sa;
    const uint32_t lane_out = (uint32_t)((size_t)(rg << 3) * sa) + t_lo;     // rows (rg << 3) | j
    const size_t out_off = p.out_off_k[coset];

    const uint32_t col0 = blockIdx.y * p.cpb;
    const uint32_t col1 = col0 + p.cpb < p.ncols ? col0 + p.cpb : p.ncols;
    gl_t nx[8];
    auto fetch = [&](uint32_t col) {
        const gl_t* __restrict__ src = p.in + (size_t)col * p.cs_in;         // wave-uniform
#pragma unroll
        for (int j = 0; j < 8; j++) nx[j] = (src + (size_t)j * in_step)[lane_in];
    };
    if (col0 < col1) fetch(col0);
    int ex = 0;                                                               // exchanges done so far: alternates the two LDS images
    for (uint32_t col = col0; col < col1; col++) {
        gl_t* __restrict__ dst = p.out + (size_t)col * p.cs_out + out_off;   // wave-uniform
        gl_t x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = nx[j];
        if (col + 1 < col1) fetch(col + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F0) R0::compute_fact(x, w0);
        else R0::compute(x, w0);
        if (NR > 1) {
            gl_t* const img = lds + (ex & 1) * IMG;
            ex++;
            if (RUNS) {
#pragma unroll
                for (int j = 0; j < 8; j++) img[bA * (R + 1) + R0::row(rgA, j)] = x[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = img[b * (R + 1) + R1::row(rg, j)];
            } else {
                R0::lds_write(img, T, b, rg, x);
                __syncthreads();
                R1::lds_read(img, T, b, rg, x);
            }
            if constexpr (F1) R1::compute_fact(x, w1);
            else R1::compute(x, w1);
        }
        if (NR > 2) {
            gl_t* const img = lds + (ex & 1) * IMG;
            ex++;
            R1::lds_write(img, T, b, rg, x);
            __syncthreads();
            R2::lds_read(img, T, b, rg, x);
            if constexpr (F2) R2::compute_fact(x, w2);
            else R2::compute(x, w2);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x[j] = gl_mul_loose(x[j], post[j]);
            if (j & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            (dst + (size_t)j * sa)[lane_out] = x[j];
        }
    }
}

template <int S, bool RUNS = false>
static void launch_lde_upper_t(zkm_ctx* c, const lde_upper_args& a) {
    static std::atomic<uint64_t> lds_ok{0};
    ntt_allow_big_lds(c, k_lde_upper<S, RUNS>, lds_ok);
    const size_t shmem = 2 * (4096 + (4096 >> S)) * sizeof(gl_t);             // two images of the 2^S x 2^(12-S) tile (+ one pad word per column)
    const uint32_t ntiles = (1u << a.S2) >> (12 - S);
    dim3 grid(ntiles * 4, (a.ncols + a.cpb - 1) / a.cpb);
    hipLaunchKernelGGL((k_lde_upper<S, RUNS>), grid, dim3(512), shmem, c->stream, a);
}

// W(k, q) = shift_c^(n / 2^(k+1)) * w_(2^(k+1))^bitrev_k(q) at [c][2^k + q], c = 0..3, shift_c = shift * w_4n^c
static const gl_t* lde_ct_table(zkm_ctx* c, uint64_t shift, unsigned log_n, unsigned S) {
    auto key = std::make_tuple(shift, log_n, S);
    auto it = c->lde_ct_tables.find(key);
    if (it != c->lde_ct_tables.end()) return it->second;
    const size_t R = (size_t)1 << S;
    std::vector<gl_t> host(4 * R, 0);
    const gl_t w4n = gl_root_of_unity(log_n + 2);
    gl_t sk = shift;
    for (unsigned cs = 0; cs < 4; cs++) {
        for (unsigned k = 0; k < S; k++) {
            const gl_t base = gl_pow(sk, (uint64_t)1 << (log_n - k - 1));
            const gl_t wk = gl_root_of_unity(k + 1);
            for (uint32_t q = 0; q < (1u << k); q++)
                host[cs * R + ((size_t)1 << k) + q] = gl_mul(base, gl_pow(wk, k ? bitrev32(q, k) : 0));
        }
        sk = gl_mul(sk, w4n);
    }
    gl_t* d = (gl_t*)c->alloc(host.size() * sizeof(gl_t));
    ZKM_HIP_CHECK(hipMemcpyAsync(d, host.data(), host.size() * sizeof(gl_t), hipMemcpyHostToDevice, c->stream));
    ZKM_HIP_CHECK(hipStreamSynchronize(c->stream));  // host vector goes out of scope
    c->lde_ct_tables[key] = d;
    c->resident_bytes += host.size() * sizeof(gl_t);
    return d;
}

// Coset-split LDE (rate 4): the evaluations on g<w_4n> in bit-reversed order are four blocks of n -- block bitrev2(k) holds the
// size-n DIF transform of c_t (g w_4n^k)^t (position bitrev(k + 4 i) = bitrev2(k) n + bitrev(i)).  Two launches: (A) the upper
// S1 = log_n - S2 stages of all four transforms, strided, coset-fused (the coefficients are read once from HBM, three more times from
// L2; 4n words written); (B) the lower S2 stages on contiguous 2^S2 blocks, in place.  HBM traffic per column 8n + 32n + 64n =
// 104n bytes against 168n of the three-pass zero-padded transform.
//   log_n      14   15..17   18..20    21    22
//   S1 / S2   3/11  3..5/12  6..8/12  8/13  9/13      S1 >= 6: block-twiddle kernel k_lde_upper<S1>; below: plain DIF pass, factored pre-scale
// S2 = 12 wherever pass A can take the other log_n - 12 stages: the 2^12-element block kernel runs three workgroups per CU, the 2^13
// one a single one (measured at 262 x 2^20: 6.8 ms / 12 stages against 8.4 ms / 13 stages, profiles/r02_ntt_split_ab.txt).  2^22-row
// polynomials (BASELINE config 4) need nine upper stages: a 512-row x 8-column tile, i.e. 64 B row segments -- slower per byte than
// the 128 B segments of S1 = 8, still 104n instead of 168n bytes.
static bool lde_coset_split(zkm_ctx* c, const gl_t* coeffs, gl_t* out, size_t ncols, unsigned log_n, uint64_t shift, unsigned coeff_s1) {
    if (log_n < 14 || log_n > 22 || shift <= 1) return false;
    if (coeff_s1 && (coeff_s1 != log_n - 12 || coeff_s1 < 6 || coeff_s1 > 8)) throw std::runtime_error("internal: LDE of an unexpected coefficient layout");
    const size_t n = (size_t)1 << log_n, N = n << 2;
    const int S2 = log_n == 14 ? 11 : (log_n <= 20 ? 12 : 13);
    const int S1 = (int)log_n - S2;   // 3..9
    c->ensure_twiddles(log_n + 2);
    const gl_t w4n = gl_root_of_unity(log_n + 2);
    const gl_t* pre_tab_k[4];
    size_t out_off_k[4];
    gl_t sk = shift;
    for (unsigned k = 0; k < 4; k++) {
        pre_tab_k[k] = c->pow_table(sk, log_n);
        out_off_k[k] = (size_t)bitrev32(k, 2) * n;
        sk = gl_mul(sk, w4n);
    }
    if (S1 >= 6) {
        lde_upper_args u{};
        u.in = coeffs; u.out = out; u.cs_in = n; u.cs_out = N; u.ncols = (uint32_t)ncols; u.log_n = log_n; u.S2 = (uint32_t)S2;
        u.tw = c->tw.fwd; u.ct = lde_ct_table(c, shift, log_n, (unsigned)S1);
        for (unsigned k = 0; k < 4; k++) { u.pre_tab_k[k] = pre_tab_k[k]; u.out_off_k[k] = out_off_k[k]; }
        const size_t ntiles = ((size_t)1 << S2) >> (12 - S1);
        // columns per workgroup (as launch_pass).  Two things were measured against the L2 misses of the four cosets' re-reads (1.4 GB per
        // proof) and lost, profiles/r03_ntt_variants.txt: 8 columns per workgroup (43.0 GB, +7 % time), two cosets per 1024-thread
        // workgroup (41.7 GB, +12 % time).
        size_t want = (ncols * ntiles) / 2048;
        u.cpb = (uint32_t)(want < 1 ? 1 : (want > 16 ? 16 : want));
        if (u.cpb > u.ncols) u.cpb = u.ncols;
        zkm_prof_scope ps(c, "ntt_pass_strided");
        if (coeff_s1) {
            switch (S1) {
                case 6: launch_lde_upper_t<6, true>(c, u); break;
                case 7: launch_lde_upper_t<7, true>(c, u); break;
                default: launch_lde_upper_t<8, true>(c, u); break;
            }
        } else {
            switch (S1) {
                case 6: launch_lde_upper_t<6>(c, u); break;
                case 7: launch_lde_upper_t<7>(c, u); break;
                case 8: launch_lde_upper_t<8>(c, u); break;
                default: launch_lde_upper_t<9>(c, u); break;
            }
        }
        ZKM_HIP_CHECK(hipGetLastError());
    } else {
        ntt_pass_args a{};
        a.in = coeffs; a.out = out; a.cs_in = n; a.cs_out = N; a.ncols = (uint32_t)ncols;
        a.tw = c->tw.fwd; a.m = (uint32_t)S2; a.post_scale = 1; a.n_in = ~(size_t)0; a.pre_log = log_n;  // (every offset of the tile is < n)
        a.log_T = pick_log_T(S1, (size_t)1 << S2);
        a.tp = 1u << a.log_T;
        a.n_lo = (uint32_t)(((size_t)1 << S2) >> a.log_T);
        a.bi_hi = a.bo_hi = n; a.bi_lo = a.bo_lo = (size_t)1 << a.log_T;  // (one tile row of blocks: t_hi is always 0)
        a.sa_in = a.sa_out = (size_t)1 << S2; a.sb_in = a.sb_out = 1;
        const size_t ntiles = a.n_lo;                                    // 2^S2 / 64 tiles of 64 columns: a multiple of 8
        a.ncoset = 4;
        const uint64_t row_step = (((uint64_t)1 << S1) >> 3) << S2;       // elements between the 8 rows a thread holds
        sk = shift;
        for (unsigned k = 0; k < 4; k++) {
            a.pre_tab_k[k] = pre_tab_k[k];
            a.out_off_k[k] = out_off_k[k];
            const gl_t D = gl_pow(sk, row_step);
            gl_t dj = 1;
            for (int j = 0; j < 8; j++) { a.pre_dj[k][j] = dj; dj = gl_mul(dj, D); }
            sk = gl_mul(sk, w4n);
        }
        a.canon_out = 0;
        launch_pass(c, S1, a, ntiles, "ntt_pass_strided");
    }
    ntt_big_pass(c, out, N, ncols, log_n + 2, S2, c->tw.fwd);
    return true;
}

// ------------------------------------------------------------------ radix-2 kernels for the sizes the pass kernels do not take
// (transforms of fewer than 8 points, natural-order transforms beyond 2^24): one global-memory DIF stage per launch (span h = 2^s) ...
__global__ __launch_bounds__(256) void k_dif_stage(gl_t* __restrict__ data, size_t col_stride, unsigned log_n, unsigned s,
                                                   const gl_t* __restrict__ tw, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t half = (size_t)1 << (log_n - 1);
    size_t col = idx >> (log_n - 1), k = idx & (half - 1);
    size_t h = (size_t)1 << s, j = k & (h - 1);
    size_t i = ((k >> s) << (s + 1)) | j;
    gl_t* p = data + col * col_stride;
    gl_t u = p[i], v = p[i + h];
    p[i] = gl_add(u, v);
    p[i + h] = gl_mul(gl_sub(u, v), tw[h + j]);
}

// ... and the lowest `lows` stages of every contiguous 2^lows chunk, staged through LDS.
template <int LOWS>
__global__ __launch_bounds__(256) void k_dif_low(gl_t* __restrict__ data, size_t col_stride, unsigned log_n,
                                                 const gl_t* __restrict__ tw) {
    __shared__ gl_t sh[1 << LOWS];
    constexpr int CH = 1 << LOWS;
    size_t chunks_per_col = (size_t)1 << (log_n - LOWS);
    size_t col = blockIdx.x / chunks_per_col, chunk = blockIdx.x % chunks_per_col;
    gl_t* p = data + col * col_stride + chunk * CH;
    for (int i = threadIdx.x; i < CH; i += 256) sh[i] = p[i];
    __syncthreads();
#pragma unroll 1
    for (int s = LOWS - 1; s >= 0; s--) {
        int h = 1 << s;
        for (int k = threadIdx.x; k < CH / 2; k += 256) {
            int j = k & (h - 1);
            int i = ((k >> s) << (s + 1)) | j;
            gl_t u = sh[i], v = sh[i + h];
            sh[i] = gl_add(u, v);
            sh[i + h] = gl_mul(gl_sub(u, v), tw[h + j]);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < CH; i += 256) p[i] = sh[i];
}

template <int LOWS>
static void launch_dif_low(zkm_ctx* c, gl_t* data, size_t ncols, size_t col_stride, unsigned log_n, const gl_t* tw) {
    size_t blocks = ncols << (log_n - LOWS);
    hipLaunchKernelGGL(k_dif_low<LOWS>, dim3(blocks), dim3(256), 0, c->stream, data, col_stride, log_n, tw);
}

void zkm_ntt_dif_bitrev(zkm_ctx* c, gl_t* data, size_t ncols, size_t col_stride, unsigned log_n, bool inverse) {
    if (log_n == 0 || ncols == 0) return;
    c->ensure_twiddles(log_n);
    const gl_t* tw = inverse ? c->tw.inv : c->tw.fwd;
    unsigned lows = log_n < 10 ? log_n : 10;
    size_t total = ncols << (log_n - 1);
    for (unsigned s = log_n; s-- > lows;) {
        zkm_prof_scope ps(c, "ntt_dif_stage");
        hipLaunchKernelGGL(k_dif_stage, dim3((total + 255) / 256), dim3(256), 0, c->stream, data, col_stride, log_n, s, tw, total);
    }
    {
        zkm_prof_scope ps(c, "ntt_dif_low");
        switch (lows) {
            case 1: launch_dif_low<1>(c, data, ncols, col_stride, log_n, tw); break;
            case 2: launch_dif_low<2>(c, data, ncols, col_stride, log_n, tw); break;
            case 3: launch_dif_low<3>(c, data, ncols, col_stride, log_n, tw); break;
            case 4: launch_dif_low<4>(c, data, ncols, col_stride, log_n, tw); break;
            case 5: launch_dif_low<5>(c, data, ncols, col_stride, log_n, tw); break;
            case 6: launch_dif_low<6>(c, data, ncols, col_stride, log_n, tw); break;
            case 7: launch_dif_low<7>(c, data, ncols, col_stride, log_n, tw); break;
            case 8: launch_dif_low<8>(c, data, ncols, col_stride, log_n, tw); break;
            case 9: launch_dif_low<9>(c, data, ncols, col_stride, log_n, tw); break;
            default: launch_dif_low<10>(c, data, ncols, col_stride, log_n, tw); break;
        }
    }
    ZKM_HIP_CHECK(hipGetLastError());
}

// out[c][k] = in[c][bitrev(k)] * scale * shift_inv^k   (shift table may be null; scale canonical)
__global__ __launch_bounds__(256) void k_bitrev_scale(const gl_t* __restrict__ in, size_t col_stride_in, gl_t* __restrict__ out,
                                                      size_t col_stride_out, unsigned log_n, gl_t scale,
                                                      const gl_t* __restrict__ pow_tab, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t col = idx >> log_n, k = idx & (((size_t)1 << log_n) - 1);
    gl_t v = in[col * col_stride_in + bitrev32((uint32_t)k, log_n)];
    if (scale != 1) v = gl_mul(v, scale);
    if (pow_tab) v = gl_mul(v, pow_lookup(pow_tab, log_n, k));
    out[col * col_stride_out + k] = v;
}

// (in may alias out: every thread reads and writes only its own element)
__global__ __launch_bounds__(256) void k_scale_pad(const gl_t* in, size_t col_stride_in, gl_t* out,
                                                   size_t col_stride_out, unsigned log_n_in, unsigned log_n_out,
                                                   const gl_t* __restrict__ pow_tab, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t col = idx >> log_n_out, i = idx & (((size_t)1 << log_n_out) - 1);
    gl_t v = 0;
    if (i < ((size_t)1 << log_n_in)) {
        v = in[col * col_stride_in + i];
        if (pow_tab) v = gl_mul(v, pow_lookup(pow_tab, log_n_in, i));
    }
    out[col * col_stride_out + i] = v;
}

void zkm_launch_scale_pad(zkm_ctx* c, const gl_t* in, size_t col_stride_in, gl_t* out, size_t col_stride_out, size_t ncols,
                          unsigned log_n_in, unsigned log_n_out, uint64_t shift) {
    const gl_t* tab = shift > 1 ? c->pow_table(shift, log_n_in) : nullptr;
    size_t total = ncols << log_n_out;
    zkm_prof_scope ps(c, "ntt_scale_pad");
    hipLaunchKernelGGL(k_scale_pad, dim3((total + 255) / 256), dim3(256), 0, c->stream, in, col_stride_in, out, col_stride_out,
                       log_n_in, log_n_out, tab, total);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ values -> coefficients in the digit layout: TWO passes
// from_values needs an inverse transform natural -> natural, and a self-sorting transform of 2^20 points takes three passes of <= 8
// stages (the last one transposing).  Nothing forces the coefficients to be stored in natural order, though: their consumers are this
// library's own kernels.  A plain decimation-in-frequency transform is the strided pass of S1 = log_n - 12 stages followed by the
// 2^12-element block kernel (twelve stages per pass), 32n instead of 48n bytes per column; what it leaves in place is bit-reversed,
// and the block kernel stores its block in the order the first LDE pass reads best (k_ntt_blk12<.., PERM>): position
//     P = rev_S1(e mod 2^S1) * 2^12 + ((e >> S1) mod 2^(12-S1)) * 2^S1 + (e >> 12)          holds the coefficient of X^e
// (zkm_coeff_exponent is the inverse map).  Used for 2^18 .. 2^20 rows (S1 = 6, 7, 8: where k_lde_upper takes the first LDE pass);
// other heights keep natural order.  Consumers: k_lde_upper<S, RUNS>, the openings (exponent-indexed power table), the FRI
// combination (position-wise, then zkm_coeffs_to_natural on the composite polynomials), zkm_batch_coeffs (converts on the way out).
void zkm_intt_digit(zkm_ctx* c, const gl_t* values, size_t cs_in, gl_t* coeffs, size_t cs_out, size_t ncols, unsigned log_n) {
    const unsigned s1 = zkm_coeff_layout_s1(log_n);
    if (!s1) throw std::runtime_error("internal: no digit layout for this height");
    const size_t n = (size_t)1 << log_n;
    c->ensure_twiddles(log_n);
    ntt_pass_args a{};
    a.in = values; a.out = coeffs; a.cs_in = cs_in; a.cs_out = cs_out; a.ncols = (uint32_t)ncols;
    a.tw = c->tw.inv; a.m = 12; a.post_scale = 1; a.n_in = ~(size_t)0;
    a.log_T = 12 - s1;   // a 4096-element tile (512 threads): 2^(12 - s1) contiguous columns per row, >= 128 B segments (the 2048-element
                         // tile of pick_log_T gives an eight-stage pass 64 B rows: every cache line fetched by two workgroups)
    a.tp = 1u << a.log_T;
    a.n_lo = (uint32_t)(((size_t)1 << 12) >> a.log_T);
    a.bi_hi = a.bo_hi = n; a.bi_lo = a.bo_lo = (size_t)1 << a.log_T;
    a.sa_in = a.sa_out = (size_t)1 << 12; a.sb_in = a.sb_out = 1;
    a.canon_out = 0;
    launch_pass(c, (int)s1, a, a.n_lo, "ntt_pass_strided");
    ntt_big_args b{coeffs, cs_out, (uint32_t)ncols, (uint32_t)(n >> 12), 1u, c->tw.inv, gl_inv((gl_t)(n % GL_P)), s1};
    zkm_prof_scope ps(c, "ntt_pass_big");
    launch_blk12<true, true>(c, b);
    ZKM_HIP_CHECK(hipGetLastError());
}

// Layout conversion of whole columns, both directions (to_natural: out[e(P)] = in[P]; else out[P] = in[e(P)]).  A workgroup moves the
// tile k_lde_upper would read: 2^(12-s1) runs of 2^s1 coefficients on the digit side == 2^s1 rows of 2^(12-s1) consecutive exponents
// on the natural side, through LDS, so both sides move contiguous segments (>= 128 B).  in != out.
__global__ __launch_bounds__(256) void k_coeff_layout(const gl_t* __restrict__ in, size_t cs_in, gl_t* __restrict__ out, size_t cs_out,
                                                      unsigned log_n, unsigned s1, int to_natural) {
    __shared__ gl_t tile[4096 + 64];
    const unsigned lt = 12 - s1, T = 1u << lt, R = 1u << s1;
    const uint32_t t0 = blockIdx.x << lt;                         // first exponent-column (t_lo) of the tile
    const gl_t* src = in + (size_t)blockIdx.y * cs_in;
    gl_t* dst = out + (size_t)blockIdx.y * cs_out;
    // tile[(b * R + t_hi) with a pad of one word per 64]: element of column t_lo = t0 + b, row t_hi
    auto slot = [](unsigned i) { return i + (i >> 6); };
    for (unsigned i = threadIdx.x; i < 4096; i += 256) {
        const unsigned b = to_natural ? i >> s1 : i & (T - 1), t_hi = to_natural ? i & (R - 1) : i >> lt;
        const uint32_t t_lo = t0 + b;
        const uint32_t P = (bitrev32(t_lo & (R - 1), s1) << 12) | ((t_lo >> s1) << s1) | t_hi, e = (t_hi << 12) | t_lo;
        tile[slot(b * R + t_hi)] = src[to_natural ? P : e];
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 4096; i += 256) {
        const unsigned b = to_natural ? i & (T - 1) : i >> s1, t_hi = to_natural ? i >> lt : i & (R - 1);
        const uint32_t t_lo = t0 + b;
        const uint32_t P = (bitrev32(t_lo & (R - 1), s1) << 12) | ((t_lo >> s1) << s1) | t_hi, e = (t_hi << 12) | t_lo;
        dst[to_natural ? e : P] = tile[slot(b * R + t_hi)];
    }
}

void zkm_coeff_layout_convert(zkm_ctx* c, const gl_t* in, size_t cs_in, gl_t* out, size_t cs_out, size_t ncols, unsigned log_n, bool to_natural) {
    const unsigned s1 = zkm_coeff_layout_s1(log_n);
    if (!s1) throw std::runtime_error("internal: no digit layout for this height");
    if (!ncols) return;
    zkm_prof_scope ps(c, "coeff_layout");
    hipLaunchKernelGGL(k_coeff_layout, dim3(((size_t)1 << log_n) >> 12, (unsigned)ncols), dim3(256), 0, c->stream, in, cs_in, out, cs_out, log_n, s1,
                       to_natural ? 1 : 0);
    ZKM_HIP_CHECK(hipGetLastError());
}

void zkm_lde_bitrev(zkm_ctx* c, const gl_t* coeffs, gl_t* out, size_t ncols, unsigned log_n, unsigned rate_bits, uint64_t shift,
                    unsigned coeff_s1) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    if (rate_bits == 2 && lde_coset_split(c, coeffs, out, ncols, log_n, shift, coeff_s1)) return;
    if (coeff_s1) throw std::runtime_error("internal: the digit coefficient layout is only produced where the coset-split LDE reads it");
    if (rate_bits == 2 && shift > 1 && log_n >= 7 && log_n <= 13 && c->small_ntt) {   // (2^7, 2^8 rows: LDEs of 2^9, 2^10 points were two passes too)
        // the coset split of lde_coset_split with the whole size-n transform of a block in one workgroup: block bitrev2(k) of a column
        // = DIF transform of c_t (shift w_4n^k)^t, left bit-reversed -- one launch of 4 x ncols workgroups
        c->ensure_twiddles(log_n + 2);
        ntt_small_args a{};
        a.in = coeffs; a.out = out; a.cs_in = n; a.cs_out = N; a.log_n = log_n; a.natural_out = 0; a.tw = c->tw.fwd; a.post_scale = 1;
        const gl_t w4n = gl_root_of_unity(log_n + 2);
        gl_t sk = shift;
        for (unsigned k = 0; k < 4; k++) {
            a.pre_tab_k[k] = c->pow_table(sk, log_n);
            a.out_off_k[k] = (size_t)bitrev32(k, 2) * n;
            sk = gl_mul(sk, w4n);
        }
        launch_ntt_small(c, a, ncols, 4, "ntt_small");
        return;
    }
    if (log_n + rate_bits >= 3) {
        ntt_dif_bitrev_fast(c, coeffs, n, out, N, ncols, log_n + rate_bits, false, n, log_n, shift);
        return;
    }
    zkm_launch_scale_pad(c, coeffs, n, out, N, ncols, log_n, log_n + rate_bits, shift);
    zkm_ntt_dif_bitrev(c, out, ncols, N, log_n + rate_bits, false);
}

void zkm_ntt_natural(zkm_ctx* c, gl_t* in_scratch, gl_t* out, size_t ncols, size_t col_stride_in, size_t col_stride_out,
                     unsigned log_n, bool inverse, uint64_t shift);
// natural -> natural with a read-only input: `in` is only read by the first pass, `scratch` (same column stride as `in` is not
// required) holds the intermediate passes.  in == scratch is allowed.
void zkm_ntt_natural_ex(zkm_ctx* c, const gl_t* in, size_t cs_in, gl_t* scratch, size_t cs_s, gl_t* out, size_t cs_out, size_t ncols,
                        unsigned log_n, bool inverse, uint64_t shift) {
    if (log_n >= 3 && log_n <= 24) {
        ntt_natural_fast(c, in, cs_in, scratch, cs_s, out, cs_out, ncols, log_n, inverse, shift);
        return;
    }
    if (in != scratch)
        ZKM_HIP_CHECK(hipMemcpy2DAsync(scratch, cs_s * sizeof(gl_t), in, cs_in * sizeof(gl_t), sizeof(gl_t) << log_n, ncols,
                                       hipMemcpyDeviceToDevice, c->stream));
    zkm_ntt_natural(c, scratch, out, ncols, cs_s, cs_out, log_n, inverse, shift);
}

void zkm_ntt_natural(zkm_ctx* c, gl_t* in_scratch, gl_t* out, size_t ncols, size_t col_stride_in, size_t col_stride_out,
                     unsigned log_n, bool inverse, uint64_t shift) {
    size_t n = (size_t)1 << log_n, total = ncols << log_n;
    if (log_n >= 3 && log_n <= 24) {
        ntt_natural_fast(c, in_scratch, col_stride_in, in_scratch, col_stride_in, out, col_stride_out, ncols, log_n, inverse, shift);
        return;
    }
    if (!inverse && shift > 1) zkm_launch_scale_pad(c, in_scratch, col_stride_in, in_scratch, col_stride_in, ncols, log_n, log_n, shift);
    zkm_ntt_dif_bitrev(c, in_scratch, ncols, col_stride_in, log_n, inverse);
    gl_t scale = inverse ? gl_inv((gl_t)(n % GL_P)) : 1;
    const gl_t* tab = (inverse && shift > 1) ? c->pow_table(gl_inv(shift), log_n) : nullptr;
    zkm_prof_scope ps(c, "ntt_bitrev_scale");
    hipLaunchKernelGGL(k_bitrev_scale, dim3((total + 255) / 256), dim3(256), 0, c->stream, in_scratch, col_stride_in, out,
                       col_stride_out, log_n, scale, tab, total);
    ZKM_HIP_CHECK(hipGetLastError());
}
