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
    if (tw.fwd) { release(tw.fwd); release(tw.inv); }
    size_t total = (size_t)1 << lm;
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
    return d;
}

__device__ __forceinline__ gl_t pow_lookup(const gl_t* __restrict__ tab, unsigned log_n, size_t i) {
    unsigned h = (log_n + 1) / 2;
    return gl_mul_loose(tab[i & (((size_t)1 << h) - 1)], tab[((size_t)1 << h) + (i >> h)]);
}

// ------------------------------------------------------------------ baseline radix-2 kernels
// One global-memory DIF stage (span h = 2^s): used for the strides that do not fit one workgroup.
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

// The lowest `lows` stages of every contiguous 2^lows chunk, staged through LDS.
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

void zkm_lde_bitrev(zkm_ctx* c, const gl_t* coeffs, gl_t* out, size_t ncols, unsigned log_n, unsigned rate_bits, uint64_t shift) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    zkm_launch_scale_pad(c, coeffs, n, out, N, ncols, log_n, log_n + rate_bits, shift);
    zkm_ntt_dif_bitrev(c, out, ncols, N, log_n + rate_bits, false);
}

void zkm_ntt_natural(zkm_ctx* c, gl_t* in_scratch, gl_t* out, size_t ncols, size_t col_stride_in, size_t col_stride_out,
                     unsigned log_n, bool inverse, uint64_t shift) {
    size_t n = (size_t)1 << log_n, total = ncols << log_n;
    if (!inverse && shift > 1) zkm_launch_scale_pad(c, in_scratch, col_stride_in, in_scratch, col_stride_in, ncols, log_n, log_n, shift);
    zkm_ntt_dif_bitrev(c, in_scratch, ncols, col_stride_in, log_n, inverse);
    gl_t scale = inverse ? gl_inv((gl_t)(n % GL_P)) : 1;
    const gl_t* tab = (inverse && shift > 1) ? c->pow_table(gl_inv(shift), log_n) : nullptr;
    zkm_prof_scope ps(c, "ntt_bitrev_scale");
    hipLaunchKernelGGL(k_bitrev_scale, dim3((total + 255) / 256), dim3(256), 0, c->stream, in_scratch, col_stride_in, out,
                       col_stride_out, log_n, scale, tab, total);
    ZKM_HIP_CHECK(hipGetLastError());
}
