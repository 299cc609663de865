// gl_dev.h -- Goldilocks field arithmetic for gfx950 device code (and the host side of the library).
//
// F = Z/p, p = 2^64 - 2^32 + 1; F2 = F[X]/(X^2 - 7).  The reference gets this arithmetic from the
// un-vendored plonky2_field crate (call sites prover/src/prover.rs:595, 690-696); results are
// mathematically determined, so bit-exactness only requires canonical outputs.
//
// Representation contract used throughout the kernels:
//   * "canonical"  : value in [0, p).  Everything stored to HBM is canonical.
//   * "loose"      : any uint64_t (represents its residue mod p).  gl_mul / gl_reduce128 return loose
//                    values; gl_add / gl_sub take one loose and one canonical operand where noted.
// Every function documents its pre/post-conditions; no function relies on "unlikely" overflow cases.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

typedef uint64_t gl_t;

static constexpr uint64_t GL_P = 0xFFFFFFFF00000001ULL;
static constexpr uint64_t GL_EPS = 0xFFFFFFFFULL;  // 2^32 - 1 == 2^64 mod p
static constexpr uint64_t GL_GENERATOR = 14293326489335486720ULL;
static constexpr uint64_t GL_POW2_GENERATOR = 7277203076849721926ULL;

// loose -> canonical
GL_HD gl_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

// canonical + canonical -> canonical
GL_HD gl_t gl_add(gl_t a, gl_t b) {
    uint64_t s = a + b;
    // a + b < 2p < 2^65.  If the 64-bit add wrapped, the true sum is s + 2^64 and s + 2^64 - p == s + EPS (mod 2^64).
    return (s < a || s >= GL_P) ? s - GL_P : s;
}
// canonical - canonical -> canonical
GL_HD gl_t gl_sub(gl_t a, gl_t b) { return a >= b ? a - b : a - b + GL_P; }
GL_HD gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }

// loose + loose -> loose.  s = a + b may wrap once; 2^64 == EPS, and (s + EPS) may wrap once more, in
// which case s >= 2^64 - EPS so the second wrapped value is < EPS and adding EPS again cannot wrap.
GL_HD uint64_t gl_add_loose(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) {
        uint64_t t = s + GL_EPS;
        s = t < s ? t + GL_EPS : t;
    }
    return s;
}

// loose + canonical -> loose (one correction: a + b < 2^64 + p, so after a wrap s <= p - 2 and s + EPS < 2^64)
GL_HD uint64_t gl_add_lc(uint64_t a, gl_t b) {
    uint64_t s = a + b;
    s += (s < a) ? GL_EPS : 0;
    return s;
}
// loose - canonical -> loose (one correction: after a borrow d = a - b + 2^64 >= 2^64 - (p - 1) = 2^32 > EPS)
GL_HD uint64_t gl_sub_lc(uint64_t a, gl_t b) {
    uint64_t d = a - b;
    d -= (a < b) ? GL_EPS : 0;
    return d;
}

// loose +/- loose -> loose for the NTT butterflies.  a + b wraps once with probability ~1/2 (corrected branch-free from the add's own
// carry-out) and a SECOND time only when both operands are non-canonical (a, b >= p: probability ~2^-64 on field data); likewise a
// second borrow of a - b needs b >= p and a < 2^32.  The second correction is therefore a wave-uniform branch on the carry mask
// that is never taken in practice instead of a canonicalisation of one operand on every butterfly (4 VALU instructions, ~17 issue
// cycles): correctness does not depend on it being rare.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t gl_add_rr(uint64_t a, uint64_t b) {
    uint32_t rl, rh, e;
    uint64_t c2;
    asm("v_add_co_u32 %0, vcc, %4, %5\n\t"
        "v_addc_co_u32 %1, vcc, %6, %7, vcc\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "s_mov_b64 %3, vcc"
        : "=&v"(rl), "=&v"(rh), "=&v"(e), "=s"(c2)
        : "v"((uint32_t)a), "v"((uint32_t)b), "v"((uint32_t)(a >> 32)), "v"((uint32_t)(b >> 32))
        : "vcc");
    uint64_t r = ((uint64_t)rh << 32) | rl;
    if (__builtin_expect(c2 != 0, 0)) {              // c2 is the wave's carry mask: a scalar branch
        asm volatile("; ZKM_COLD (tools/isa_histogram.py: never-taken block)");
        const bool mine = (c2 >> (__lane_id() & 63)) & 1;
        if (mine) r += GL_EPS;                       // after a second wrap r < 2^32: cannot wrap again
    }
    return r;
}
__device__ __forceinline__ uint64_t gl_sub_rr(uint64_t a, uint64_t b) {
    uint32_t rl, rh, e;
    uint64_t c2;
    asm("v_sub_co_u32 %0, vcc, %4, %5\n\t"
        "v_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "s_mov_b64 %3, vcc"
        : "=&v"(rl), "=&v"(rh), "=&v"(e), "=s"(c2)
        : "v"((uint32_t)a), "v"((uint32_t)b), "v"((uint32_t)(a >> 32)), "v"((uint32_t)(b >> 32))
        : "vcc");
    uint64_t r = ((uint64_t)rh << 32) | rl;
    if (__builtin_expect(c2 != 0, 0)) {
        asm volatile("; ZKM_COLD");
        const bool mine = (c2 >> (__lane_id() & 63)) & 1;
        if (mine) r -= GL_EPS;                       // after a second borrow r > 2^64 - 2^32: cannot borrow again
    }
    return r;
}
#else
GL_HD uint64_t gl_add_rr(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) { uint64_t t = s + GL_EPS; s = t < s ? t + GL_EPS : t; }
    return s;
}
GL_HD uint64_t gl_sub_rr(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    if (a < b) { uint64_t t = d - GL_EPS; d = t > d ? t - GL_EPS : t; }
    return d;
}
#endif

// 128-bit (hi:lo) -> loose.  Standard Goldilocks reduction: 2^64 == 2^32 - 1, 2^96 == -1, written for gfx950 issue costs
// (tools/ubench3.hip: compare + select chains are the expensive part of a modmul, moves and plain 32-bit ops are cheap):
//   t0 = lo - h1           borrow <=> lo < h1 < 2^32 <=> the high word went from 0 to 0xFFFFFFFF; the borrow mask
//                          (0xFFFFFFFF == EPS) is taken from the sign of (t0_hi & ~lo_hi) and subtracted: t0 in (p - 2^32, p)
//   r  = h0 * EPS + t0     one multiply-add; if it wrapped, r < 2^64 - 2^33 + 1 and adding EPS cannot wrap again
GL_HD uint64_t gl_reduce128(uint64_t lo, uint64_t hi) {
    uint32_t h1 = (uint32_t)(hi >> 32), h0 = (uint32_t)hi;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_REDUCE_BRANCHFREE)
    // t0 = lo - h1 borrows only if lo < h1 < 2^32, i.e. with probability < 2^-32 on products of field data: the borrow mask comes out
    // of the subtract itself and the correction (t0 -= EPS: the wrapped value is > 2^64 - 2^32, it cannot borrow again) sits behind a
    // wave-uniform branch that is practically never taken -- 2 VALU instructions instead of 6 (the r01 form builds the mask from
    // the sign of t0_hi & ~lo_hi and subtracts it unconditionally: kept below for GL_REDUCE_BRANCHFREE).  A modular multiply is
    // 12 VALU instructions (was 17): Poseidon 14.6k -> 13.3k per permutation, leaf hashing 48.8 -> 44.5 ms.  The branch is a
    // scheduling barrier, so kernels that live on interleaving independent products at low occupancy (quotient, openings:
    // +30 % / +70 % with it) define GL_REDUCE_BRANCHFREE.  tests: zkm_field_selftest feeds the borrowing products.
    uint32_t tl, th;
    uint64_t bm;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\tv_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\ts_mov_b64 %2, vcc"
        : "=&v"(tl), "=&v"(th), "=s"(bm)
        : "v"((uint32_t)lo), "v"((uint32_t)(lo >> 32)), "v"(h1)
        : "vcc");
    uint64_t t0 = ((uint64_t)th << 32) | tl;
    if (__builtin_expect(bm != 0, 0)) {
        asm volatile("; ZKM_COLD");
        if ((bm >> (__lane_id() & 63)) & 1) t0 -= GL_EPS;
    }
#else
    uint64_t t0 = lo - h1;
    uint32_t m = (uint32_t)((int32_t)((uint32_t)(t0 >> 32) & ~(uint32_t)(lo >> 32)) >> 31);
    t0 -= m;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // The wrap of h0 * EPS + t0 is the multiply-add's own carry-out (an SGPR pair): mad, select, add -- no 64-bit compare.
    // The compiler does not form this (it adds separately and compares); tools/ubench3.hip: +14 % x^7 s-boxes/s.
    uint64_t r, carry;
    uint32_t wrap;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, %1" : "=&v"(r), "=&s"(carry), "=v"(wrap) : "v"(h0), "v"(t0));
    return r + wrap;
#else
    uint64_t r = (uint64_t)h0 * 0xFFFFFFFFu + t0;
    r += (r < t0) ? GL_EPS : 0;
    return r;
#endif
}

// 64 x 64 -> 128.  Device: four chained 32x32+64 multiply-adds sharing their partial products (no compares; the one possible
// overflow is read from the multiply-add's carry-out).  Measured on MI355X (tools/ubench3.hip): 18% more x^7 s-boxes/s than `a * b` next to
// `__umul64hi(a, b)` (which evaluates the low product twice), and faster than carry-propagating forms with fewer
// instructions -- v_mov is cheap on gfx950, carry/compare/select chains are not.
GL_HD void gl_mul_wide(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t p00 = (uint64_t)al * bl;
    uint64_t m1 = (uint64_t)al * bh + (p00 >> 32);          // <= (2^32-1)^2 + 2^32 - 1 < 2^64
    // m2 = ah * bl + m1 may exceed 64 bits; its overflow is the multiply-add's carry-out (weight 2^32 in the high word).
    // One 64-bit accumulator for both middle terms instead of splitting m1 into halves: two instructions less per product
    // (tools/ubench3.hip: +8 % x^7 s-boxes/s on top of the carry-out reduction).
    uint64_t m2, c;
    uint32_t cw;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, 1, %1" : "=&v"(m2), "=&s"(c), "=v"(cw) : "v"(ah), "v"(bl), "v"(m1));
    hi = (uint64_t)ah * bh + ((m2 >> 32) | ((uint64_t)cw << 32));   // the true high word, < 2^64
    lo = (m2 << 32) | (uint32_t)p00;
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (uint64_t)p;
    hi = (uint64_t)(p >> 64);
#endif
}

// loose * 2^K -> loose for the small roots of unity of this field: w_8 = 2^24, w_4 = 2^48 (and w_16 = 2^12, w_32 = 2^6, w_64 = 2^3),
// i.e. the twiddles of the last transform stages are shifts: x * 2^K = (x << K) reduced, one multiply-add (K < 32) or the plain
// 128-bit reduction (K < 64) instead of the four multiply-adds of a general product, and no twiddle register.
template <int K>
GL_HD uint64_t gl_mul_pow2(uint64_t x) {
    static_assert(K > 0 && K < 96, "gl_mul_pow2: 0 < K < 96");
    if constexpr (K < 64) {
        // x * 2^K = lo + 2^64 hi with hi = x >> (64 - K); for K < 32 the high word h1 of hi is zero and the compiler drops its terms
        return gl_reduce128(x << K, x >> (64 - K));
    } else {
        return gl_mul_pow2<K - 48>(gl_mul_pow2<48>(x));
    }
}

// loose * loose -> loose
GL_HD uint64_t gl_mul_loose(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_REDUCE_BRANCHFREE)
    // the branch-free form of the fused product below (kernels that interleave independent products at low occupancy): the same
    // borrow-in, then the r01 borrow mask -- lo < h1' + c <= 2^32 still means "high word went from 0 to 0xFFFFFFFF"
    const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)al * bl;
    const uint64_t m1 = (uint64_t)al * bh + (p00 >> 32);
    uint64_t m2, c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(m2), "=&s"(c) : "v"(ah), "v"(bl), "v"(m1));
    const uint64_t hi = (uint64_t)ah * bh + (m2 >> 32);
    uint32_t tl, th;
    asm("v_subb_co_u32 %0, vcc, %2, %4, %5\n\tv_subbrev_co_u32 %1, vcc, 0, %3, vcc"
        : "=&v"(tl), "=&v"(th)
        : "v"((uint32_t)p00), "v"((uint32_t)m2), "v"((uint32_t)(hi >> 32)), "s"(c)
        : "vcc");
    uint64_t t0 = ((uint64_t)th << 32) | tl;
    t0 -= (uint32_t)((int32_t)(th & ~(uint32_t)m2) >> 31);
    uint64_t r, carry;
    uint32_t wrap;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, %1" : "=&v"(r), "=&s"(carry), "=v"(wrap) : "v"((uint32_t)hi), "v"(t0));
    return r + wrap;
#elif defined(__HIP_DEVICE_COMPILE__)
    // gl_mul_wide + gl_reduce128 fused around the ONE carry of the product: m2 = ah bl + m1 may exceed 64 bits, and that bit
    // weighs 2^96 == -1.  Instead of materialising it (v_cndmask) and adding it into the high word, it stays in the SGPR pair the
    // multiply-add wrote it to and enters the reduction's first subtract as its borrow-in:
    //   a b = lo + 2^64 hi' + 2^96 c,  hi' = ah bh + (m2 >> 32) < 2^64   ==>   t0 = lo - h1' - c,  r = h0' EPS + t0
    // t0 borrows only if lo < h1' + c <= 2^32 (probability ~2^-32): same never-taken branch as in gl_reduce128.  11 VALU instructions.
    const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)al * bl;
    const uint64_t m1 = (uint64_t)al * bh + (p00 >> 32);
    uint64_t m2, c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(m2), "=&s"(c) : "v"(ah), "v"(bl), "v"(m1));
    const uint64_t hi = (uint64_t)ah * bh + (m2 >> 32);
    uint32_t tl, th;
    uint64_t bm;
    asm("v_subb_co_u32 %0, vcc, %3, %5, %6\n\tv_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\ts_mov_b64 %2, vcc"
        : "=&v"(tl), "=&v"(th), "=s"(bm)
        : "v"((uint32_t)p00), "v"((uint32_t)m2), "v"((uint32_t)(hi >> 32)), "s"(c)
        : "vcc");
    uint64_t t0 = ((uint64_t)th << 32) | tl;
    if (__builtin_expect(bm != 0, 0)) {
        asm volatile("; ZKM_COLD");
        if ((bm >> (__lane_id() & 63)) & 1) t0 -= GL_EPS;
    }
    uint64_t r, carry;
    uint32_t wrap;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, %1" : "=&v"(r), "=&s"(carry), "=v"(wrap) : "v"((uint32_t)hi), "v"(t0));
    return r + wrap;
#else
    uint64_t lo, hi;
    gl_mul_wide(a, b, lo, hi);
    return gl_reduce128(lo, hi);
#endif
}
// loose * loose -> canonical
GL_HD gl_t gl_mul(uint64_t a, uint64_t b) { return gl_canon(gl_mul_loose(a, b)); }
GL_HD gl_t gl_sqr(uint64_t a) { return gl_mul(a, a); }

GL_HD gl_t gl_pow(gl_t b, uint64_t e) {
    gl_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_sqr(b);
        e >>= 1;
    }
    return r;
}
GL_HD gl_t gl_exp_pow2(gl_t a, unsigned k) {
    while (k--) a = gl_sqr(a);
    return a;
}
// a^(p - 2), p - 2 = 2^64 - 2^32 - 1 = (31 one bits)(0)(32 one bits): the addition chain through a^(2^k - 1), k = 2, 3, 6, 12, 24, 30, 31,
// then (a^(2^31 - 1))^(2^32) * a^(2^31 - 1), squared, times a -- 63 squarings and 9 products instead of the 64 + 63 of square-and-multiply
// (the inverse is unique: the same canonical word; 0 -> 0).  The CTL / lookup column kernels invert once per row and column set.
GL_HD gl_t gl_inv(gl_t a) {
    const gl_t t2 = gl_mul(gl_sqr(a), a);
    const gl_t t3 = gl_mul(gl_sqr(t2), a);
    const gl_t t6 = gl_mul(gl_exp_pow2(t3, 3), t3);
    const gl_t t12 = gl_mul(gl_exp_pow2(t6, 6), t6);
    const gl_t t24 = gl_mul(gl_exp_pow2(t12, 12), t12);
    const gl_t t30 = gl_mul(gl_exp_pow2(t24, 6), t6);
    const gl_t t31 = gl_mul(gl_sqr(t30), a);
    const gl_t t63 = gl_mul(gl_exp_pow2(t31, 32), t31);
    return gl_mul(gl_sqr(t63), a);
}
GL_HD gl_t gl_root_of_unity(unsigned k) { return gl_exp_pow2(GL_POW2_GENERATOR, 32 - k); }

// ---- F2 (all canonical) ----
struct gl2_t {
    gl_t c0, c1;
};
GL_HD gl2_t gl2_make(gl_t a, gl_t b) { return gl2_t{a, b}; }
GL_HD gl2_t gl2_add(gl2_t a, gl2_t b) { return gl2_t{gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)}; }
GL_HD gl2_t gl2_sub(gl2_t a, gl2_t b) { return gl2_t{gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)}; }
GL_HD gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl_t a0b0 = gl_mul(a.c0, b.c0), a1b1 = gl_mul(a.c1, b.c1);
    return gl2_t{gl_add(a0b0, gl_mul(7, a1b1)), gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0))};
}
GL_HD gl2_t gl2_scalar_mul(gl2_t a, gl_t s) { return gl2_t{gl_mul(a.c0, s), gl_mul(a.c1, s)}; }
GL_HD bool gl2_eq(gl2_t a, gl2_t b) { return a.c0 == b.c0 && a.c1 == b.c1; }
GL_HD gl2_t gl2_inv(gl2_t a) {
    gl_t norm = gl_sub(gl_sqr(a.c0), gl_mul(7, gl_sqr(a.c1)));
    gl_t ni = gl_inv(norm);
    return gl2_t{gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni)};
}
GL_HD gl2_t gl2_pow(gl2_t b, uint64_t e) {
    gl2_t r{1, 0};
    while (e) {
        if (e & 1) r = gl2_mul(r, b);
        b = gl2_mul(b, b);
        e >>= 1;
    }
    return r;
}
GL_HD gl2_t gl2_exp_pow2(gl2_t a, unsigned k) {
    while (k--) a = gl2_mul(a, a);
    return a;
}

GL_HD uint32_t bitrev32(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}
