/* oracle/gl.h -- Goldilocks field F (p = 2^64 - 2^32 + 1) and its quadratic extension F2 = F[X]/(X^2-7).
 *
 * TEST INFRASTRUCTURE ONLY (the CPU oracle).  Nothing under zkm_amd/ may include this.
 *
 * The arithmetic lives in the un-vendored dependency plonky2_field 0.1.1 (zkMIPS/plonky2 @ zkm_dev,
 * rev f1e28a6d, /root/reference/prover/examples/Cargo.lock:3233-3283); the reference only calls it
 * (e.g. prover/src/prover.rs:595, 690-696).  Results are mathematically determined, so any correct
 * implementation is bit-exact after canonicalisation.  Every value here is kept canonical (< p).
 * Domain constants (generator, 2^32-th root, non-residue 7) are the published Goldilocks parameters
 * and are checked for self-consistency in tests/test_oracle_field.py.
 */
#ifndef ZKM_ORACLE_GL_H
#define ZKM_ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t gl_t;
typedef unsigned __int128 u128_t;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GENERATOR 14293326489335486720ULL      /* multiplicative generator == coset_shift() */
#define GL_POW2_GENERATOR 7277203076849721926ULL  /* order 2^32 */
#define GL_TWO_ADICITY 32
#define GL_EXT_W 7ULL                             /* X^2 = 7 */

static inline gl_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

/* (branch-free: whether a sum wraps is a coin flip on field data, and a mispredicted branch costs more than the addition -- the
 * sparse partial rounds of the permutation alone are ~480 of these per permutation) */
static inline gl_t gl_add(gl_t a, gl_t b) {
    uint64_t s = a + b;
    return s - ((0 - (uint64_t)((s < a) | (s >= GL_P))) & GL_P);
}
static inline gl_t gl_sub(gl_t a, gl_t b) {
    uint64_t d = a - b;
    return d + ((0 - (uint64_t)(a < b)) & GL_P);
}
static inline gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }

static inline gl_t gl_reduce128(u128_t x) {
    /* 2^64 == 2^32 - 1 and 2^96 == -1 (mod p).  Branch-free: the two corrections are data dependent with probability ~1/2
     * (mispredicted branches cost more than the arithmetic), so they are applied through masks. */
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0, r;
    uint64_t borrow = __builtin_sub_overflow(lo, hi_hi, &t0);
    t0 -= (0 - borrow) & GL_EPS;                 /* lo - hi_hi + 2^64 == ... - EPS (mod p); cannot borrow again: t0 >= 2^64 - 2^32 */
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t carry = __builtin_add_overflow(t0, t1, &r);
    r += (0 - carry) & GL_EPS;                   /* cannot carry again: after a wrap r < 2^64 - 2^32 */
    return gl_canon(r);
}
static inline gl_t gl_mul(gl_t a, gl_t b) { return gl_reduce128((u128_t)a * b); }
static inline gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }

static inline gl_t gl_pow(gl_t b, uint64_t e) {
    gl_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_sqr(b);
        e >>= 1;
    }
    return r;
}
static inline gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }
static inline gl_t gl_exp_pow2(gl_t a, unsigned k) {
    while (k--) a = gl_sqr(a);
    return a;
}
/* primitive 2^k-th root of unity: POWER_OF_TWO_GENERATOR^(2^(32-k)) */
static inline gl_t gl_root_of_unity(unsigned k) { return gl_exp_pow2(GL_POW2_GENERATOR, GL_TWO_ADICITY - k); }

/* ---- F2 ---- */
typedef struct { gl_t c[2]; } gl2_t;

static inline gl2_t gl2_make(gl_t a, gl_t b) { gl2_t r = {{a, b}}; return r; }
static inline gl2_t gl2_from_base(gl_t a) { return gl2_make(a, 0); }
static inline int gl2_eq(gl2_t a, gl2_t b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) { return gl2_make(gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])); }
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) { return gl2_make(gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])); }
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl_t a0b0 = gl_mul(a.c[0], b.c[0]), a1b1 = gl_mul(a.c[1], b.c[1]);
    gl_t x = gl_add(a0b0, gl_mul(GL_EXT_W, a1b1));
    gl_t y = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return gl2_make(x, y);
}
static inline gl2_t gl2_scalar_mul(gl2_t a, gl_t s) { return gl2_make(gl_mul(a.c[0], s), gl_mul(a.c[1], s)); }
static inline gl2_t gl2_inv(gl2_t a) {
    gl_t norm = gl_sub(gl_sqr(a.c[0]), gl_mul(GL_EXT_W, gl_sqr(a.c[1])));
    gl_t ni = gl_inv(norm);
    return gl2_make(gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni));
}
static inline gl2_t gl2_pow(gl2_t b, uint64_t e) {
    gl2_t r = gl2_from_base(1);
    while (e) {
        if (e & 1) r = gl2_mul(r, b);
        b = gl2_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline gl2_t gl2_exp_pow2(gl2_t a, unsigned k) {
    while (k--) a = gl2_mul(a, a);
    return a;
}

static inline size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
#endif
