// constraints_dev.h -- per-table STARK constraints as device functions (one thread = one point of the quotient domain).
//
// Each eval_<table>_constraints restates the table's eval_packed_generic in the reference, emitting constraints in the
// SAME ORDER (the order defines the alpha powers, constraint_consumer.rs:57-62).  `lv` points at the thread's element of
// column 0 of the trace LDE, columns are `cs` words apart, the next row's element of a column sits `dnext` words after the
// local one.  All values canonical.  Used by k_quotient in stark.hip; the CPU oracle has the same functions in
// oracle/constraints_tmpl.h (test infrastructure).
#pragma once
#include "poseidon_dev.h"
#include "hash_constants_dev.h"
#include "zkm_internal.h"

// ------------------------------------------------------------------ constraint consumer, Poseidon table
// Constraint order = alpha-power order (constraint_consumer.rs:57-62): table constraints
// (poseidon_stark.rs:554-594), then CTL checks (cross_table_lookup.rs:1067-1118), vanishing_poly.rs:30-45.
template <int NA>
struct consumer_t {
    gl_t alpha[NA], acc[NA];
    gl_t z_last, l_first, l_last;
    __device__ __forceinline__ void constraint(gl_t c) {
#pragma unroll
        for (int j = 0; j < NA; j++) acc[j] = gl_add(gl_mul(acc[j], alpha[j]), c);
    }
    __device__ __forceinline__ void transition(gl_t c) { constraint(gl_mul(c, z_last)); }
    __device__ __forceinline__ void last_row(gl_t c) { constraint(gl_mul(c, l_last)); }
    __device__ __forceinline__ void first_row(gl_t c) { constraint(gl_mul(c, l_first)); }
};

template <int NA>
__device__ __forceinline__ void sbox_constraints(consumer_t<NA>& k, gl_t in, gl_t inter, gl_t out) {
    k.constraint(gl_sub(gl_mul(gl_mul(in, in), in), inter));
    k.constraint(gl_sub(gl_mul(gl_mul(in, inter), inter), out));
}

__device__ __forceinline__ void mds_canon(gl_t s[12]) {
    poseidon_mds(s);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

// lv(c) = trace LDE value of column c at this thread's row
template <int NA>
__device__ void eval_poseidon_constraints(const gl_t* __restrict__ lv, size_t cs, consumer_t<NA>& k) {
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = lv[(size_t)(1 + i) * cs];
    int rc = 0;
#pragma unroll 1
    for (int r = 0; r < 4; r++, rc++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            gl_t x = gl_add(s[i], gl_canon(PC::ZKM_POSEIDON_RC[rc * 12 + i]));
            gl_t tmp = lv[(size_t)(26 + 24 * r + 2 * i) * cs], out = lv[(size_t)(26 + 24 * r + 2 * i + 1) * cs];
            sbox_constraints(k, x, tmp, out);
            s[i] = out;
        }
        mds_canon(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PC::ZKM_POSEIDON_FAST_FIRST_RC[i]);
    {
        gl_t t[12];
        t[0] = s[0];
#pragma unroll
        for (int c = 1; c < 12; c++) {
            uint64_t acc = 0;
#pragma unroll
            for (int r = 1; r < 12; r++) acc = gl_add_loose(acc, gl_mul_loose(s[r], PC::ZKM_POSEIDON_FAST_INIT[r - 1][c - 1]));
            t[c] = gl_canon(acc);
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        gl_t inter = lv[(size_t)(122 + 2 * r) * cs], out = lv[(size_t)(122 + 2 * r + 1) * cs];
        sbox_constraints(k, s[0], inter, out);
        gl_t s0 = r < 21 ? gl_add(out, PC::ZKM_POSEIDON_FAST_RC[r]) : out;
        uint64_t d = gl_mul_loose(s0, 25);
#pragma unroll
        for (int i = 1; i < 12; i++) d = gl_add_loose(d, gl_mul_loose(s[i], PC::ZKM_POSEIDON_FAST_W_HATS[r][i - 1]));
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = gl_add(s[i], gl_mul(s0, PC::ZKM_POSEIDON_FAST_VS[r][i - 1]));
        s[0] = gl_canon(d);
    }
    rc += 22;
#pragma unroll 1
    for (int r = 0; r < 4; r++, rc++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            gl_t x = gl_add(s[i], gl_canon(PC::ZKM_POSEIDON_RC[rc * 12 + i]));
            gl_t tmp = lv[(size_t)(166 + 24 * r + 2 * i) * cs], out = lv[(size_t)(166 + 24 * r + 2 * i + 1) * cs];
            sbox_constraints(k, x, tmp, out);
            s[i] = out;
        }
        mds_canon(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) k.constraint(gl_sub(s[i], lv[(size_t)(13 + i) * cs]));
}

// LogicStark (logic.rs:199-248; columns :25-50).  64 booleanity constraints then the result constraint.
template <int NA>
__device__ void eval_logic_constraints(const gl_t* __restrict__ lv, size_t cs, consumer_t<NA>& k) {
    gl_t is_and = lv[0], is_or = lv[cs], is_xor = lv[2 * cs], is_nor = lv[3 * cs];
    gl_t sum_coeff = gl_sub(gl_add(is_or, is_xor), is_nor);
    gl_t and_coeff = gl_add(gl_sub(gl_sub(is_and, is_or), gl_add(is_xor, is_xor)), is_nor);
    gl_t x = 0, y = 0, x_land_y = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        gl_t b = lv[(size_t)(4 + i) * cs];
        k.constraint(gl_mul(b, gl_sub(b, 1)));
        x = gl_add(x, gl_mul(b, (gl_t)1 << i));
    }
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        gl_t b = lv[(size_t)(36 + i) * cs];
        k.constraint(gl_mul(b, gl_sub(b, 1)));
        y = gl_add(y, gl_mul(b, (gl_t)1 << i));
        x_land_y = gl_add(x_land_y, gl_mul(gl_mul(lv[(size_t)(4 + i) * cs], b), (gl_t)1 << i));
    }
    gl_t x_op_y = gl_add(gl_add(gl_mul(sum_coeff, gl_add(x, y)), gl_mul(and_coeff, x_land_y)), gl_mul(is_nor, 0xFFFFFFFFULL));
    k.constraint(gl_sub(lv[(size_t)68 * cs], x_op_y));
}

// KeccakSpongeStark (keccak_sponge_stark.rs:456-567; columns keccak_sponge/columns.rs:19-70).  nv = lv + dnext.
template <int NA>
__device__ void eval_keccak_sponge_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    const gl_t* __restrict__ nv = lv + dnext;
    enum { FULL = 0, CONTEXT = 1, SEGMENT = 2, TIMESTAMP = 37, LEN = 38, ABSORBED = 39, FINAL_LEN = 40, ORIG_RATE = 176,
           ORIG_CAP = 210, PARTIAL = 396, DIGEST = 438 };
    gl_t full = lv[FULL];
    k.constraint(gl_mul(full, gl_sub(full, 1)));
    gl_t is_final = 0, next_final = 0;
#pragma unroll 4
    for (int i = 0; i < 136; i++) {
        is_final = gl_add(is_final, lv[(size_t)(FINAL_LEN + i) * cs]);
        next_final = gl_add(next_final, nv[(size_t)(FINAL_LEN + i) * cs]);
    }
    k.constraint(gl_mul(is_final, gl_sub(is_final, 1)));
#pragma unroll 4
    for (int i = 0; i < 136; i++) {
        gl_t f = lv[(size_t)(FINAL_LEN + i) * cs];
        k.constraint(gl_mul(f, gl_sub(f, 1)));
    }
    k.constraint(gl_mul(is_final, full));
    gl_t absorbed = lv[(size_t)ABSORBED * cs];
    k.first_row(absorbed);
#pragma unroll 2
    for (int i = 0; i < 50; i++) k.first_row(lv[(size_t)(ORIG_RATE + i) * cs]);  // original_rate then original_capacity
    // both scaled by (x - last) inside transition(): fold is_final / full into it once
    gl_t fin_t = gl_mul(is_final, k.z_last), full_t = gl_mul(full, k.z_last);
    k.constraint(gl_mul(fin_t, nv[(size_t)ABSORBED * cs]));
#pragma unroll 2
    for (int i = 0; i < 50; i++) k.constraint(gl_mul(fin_t, nv[(size_t)(ORIG_RATE + i) * cs]));
    k.constraint(gl_mul(full_t, gl_sub(lv[(size_t)CONTEXT * cs], nv[(size_t)CONTEXT * cs])));
    k.constraint(gl_mul(full_t, gl_sub(lv[(size_t)SEGMENT * cs], nv[(size_t)SEGMENT * cs])));
    k.constraint(gl_mul(full_t, gl_sub(lv[(size_t)TIMESTAMP * cs], nv[(size_t)TIMESTAMP * cs])));
#pragma unroll 2
    for (int l = 0; l < 8; l++) {
        gl_t cur = lv[(size_t)(DIGEST + 4 * l) * cs];
#pragma unroll
        for (int i = 1; i < 4; i++) cur = gl_add(cur, gl_mul(lv[(size_t)(DIGEST + 4 * l + i) * cs], (gl_t)1 << (8 * i)));
        k.constraint(gl_mul(full_t, gl_sub(nv[(size_t)(ORIG_RATE + l) * cs], cur)));
    }
#pragma unroll 2
    for (int i = 0; i < 42; i++)  // rate u32s 8..33 then the 16 capacity u32s are contiguous in both views
        k.constraint(gl_mul(full_t, gl_sub(nv[(size_t)(ORIG_RATE + 8 + i) * cs], lv[(size_t)(PARTIAL + i) * cs])));
    k.constraint(gl_mul(full_t, gl_sub(gl_add(absorbed, 136), nv[(size_t)ABSORBED * cs])));
    gl_t is_dummy = gl_sub(gl_sub(1, full), is_final);
    k.transition(gl_mul(is_dummy, gl_add(nv[FULL], next_final)));
    gl_t offset = gl_sub(lv[(size_t)LEN * cs], absorbed);
#pragma unroll 4
    for (int i = 0; i < 136; i++) k.constraint(gl_mul(lv[(size_t)(FINAL_LEN + i) * cs], gl_sub(offset, (gl_t)i)));
}

// KeccakStark (keccak/keccak_stark.rs:256-413: 3 + 320 + 50 + 320 + 50 + 4 + 50 = 797 constraints over 2431 columns;
// register map keccak/columns.rs:7-134).  Columns are streamed from HBM in constraint order (each thread owns one
// row; a wavefront reads 64 consecutive rows of one column = 512 contiguous bytes per load).
namespace kk {
enum { TIMESTAMP = 24, A = 25, C = 75, CP = 395, AP = 715, APP = 2315, APP00_BITS = 2365, APPP00 = 2429 };
__device__ const uint8_t ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
__device__ __forceinline__ int reg_a(int x, int y) { return A + (x * 5 + y) * 2; }
__device__ __forceinline__ int reg_c(int x, int z) { return C + x * 64 + z; }
__device__ __forceinline__ int reg_cp(int x, int z) { return CP + x * 64 + z; }
__device__ __forceinline__ int reg_ap(int x, int y, int z) { return AP + x * 320 + y * 64 + z; }
__device__ __forceinline__ int reg_app(int x, int y) { return APP + x * 10 + y * 2; }
__device__ __forceinline__ int reg_appp(int x, int y) { return (x == 0 && y == 0) ? (int)APPP00 : reg_app(x, y); }
__device__ __forceinline__ int mod5(int v) { return v >= 5 ? v - 5 : v; }
__device__ __forceinline__ gl_t xor_gen(gl_t x, gl_t y) { return gl_sub(gl_add(x, y), gl_mul(x, gl_add(y, y))); }
}  // namespace kk

template <int NA>
__device__ void eval_keccak_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    using namespace kk;
    const gl_t* __restrict__ nv = lv + dnext;
#define LV(c) lv[(size_t)(c) * cs]
#define NV(c) nv[(size_t)(c) * cs]
    gl_t final_step = LV(23);
    k.constraint(gl_mul(final_step, gl_sub(final_step, 1)));
    gl_t not_final = gl_sub(1, final_step);
    k.constraint(gl_mul(not_final, final_step));
    // round flags: their sum, and the round-constant bit sum_r step_r * RC_r[z] for the 7 bit positions RC uses
    gl_t sum_flags = 0, rc0 = 0, rc1 = 0, rc3 = 0, rc7 = 0, rc15 = 0, rc31 = 0, rc63 = 0;
    {
        constexpr const uint64_t (&RC)[24] = KECCAK_RC_DEV;
#pragma unroll
        for (int r = 0; r < 24; r++) {
            gl_t f = LV(r);
            sum_flags = gl_add(sum_flags, f);
            if (RC[r] & 1) rc0 = gl_add(rc0, f);
            if (RC[r] >> 1 & 1) rc1 = gl_add(rc1, f);
            if (RC[r] >> 3 & 1) rc3 = gl_add(rc3, f);
            if (RC[r] >> 7 & 1) rc7 = gl_add(rc7, f);
            if (RC[r] >> 15 & 1) rc15 = gl_add(rc15, f);
            if (RC[r] >> 31 & 1) rc31 = gl_add(rc31, f);
            if (RC[r] >> 63 & 1) rc63 = gl_add(rc63, f);
        }
    }
    k.constraint(gl_mul(gl_mul(sum_flags, not_final), gl_sub(NV(TIMESTAMP), LV(TIMESTAMP))));
    // C'[x, z] = xor(C[x, z], C[x - 1, z], C[x + 1, z - 1])
#pragma unroll 1
    for (int x = 0; x < 5; x++)
#pragma unroll 2
        for (int z = 0; z < 64; z++) {
            gl_t v = xor_gen(LV(reg_c(x, z)), xor_gen(LV(reg_c(mod5(x + 4), z)), LV(reg_c(mod5(x + 1), (z + 63) & 63))));
            k.constraint(gl_sub(LV(reg_cp(x, z)), v));
        }
    // A[x, y] limbs from xor(A'[x, y, z], C[x, z], C'[x, z])
#pragma unroll 1
    for (int x = 0; x < 5; x++)
#pragma unroll 1
        for (int y = 0; y < 5; y++)
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                gl_t acc = 0;
#pragma unroll 2
                for (int z = 32 * half + 31; z >= 32 * half; z--)
                    acc = gl_add(gl_add(acc, acc), xor_gen(LV(reg_ap(x, y, z)), xor_gen(LV(reg_c(x, z)), LV(reg_cp(x, z)))));
                k.constraint(gl_sub(acc, LV(reg_a(x, y) + half)));
            }
    // diff = sum_y A'[x, y, z] - C'[x, z] in {0, 2, 4}
#pragma unroll 1
    for (int x = 0; x < 5; x++)
#pragma unroll 2
        for (int z = 0; z < 64; z++) {
            gl_t sum = LV(reg_ap(x, 0, z));
#pragma unroll
            for (int i = 1; i < 5; i++) sum = gl_add(sum, LV(reg_ap(x, i, z)));
            gl_t diff = gl_sub(sum, LV(reg_cp(x, z)));
            k.constraint(gl_mul(gl_mul(diff, gl_sub(diff, 2)), gl_sub(diff, 4)));
        }
    // A''[x, y] = xor(B[x, y], andn(B[x + 1, y], B[x + 2, y])), B[x, y, z] = A'[(x + 3y) % 5, x, z - r] (columns.rs:91-105)
#pragma unroll 1
    for (int x = 0; x < 5; x++)
#pragma unroll 1
        for (int y = 0; y < 5; y++) {
            int x1 = mod5(x + 1), x2 = mod5(x + 2);
            int a0 = (x + 3 * y) % 5, a1 = (x1 + 3 * y) % 5, a2 = (x2 + 3 * y) % 5;
            int base0 = reg_ap(a0, x, 0), base1 = reg_ap(a1, x1, 0), base2 = reg_ap(a2, x2, 0);
            int r0 = 64 - ROT[a0][x], r1 = 64 - ROT[a1][x1], r2 = 64 - ROT[a2][x2];
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                gl_t acc = 0;
#pragma unroll 2
                for (int z = 32 * half + 31; z >= 32 * half; z--) {
                    gl_t b0 = LV(base0 + ((z + r0) & 63)), b1 = LV(base1 + ((z + r1) & 63)), b2 = LV(base2 + ((z + r2) & 63));
                    acc = gl_add(gl_add(acc, acc), xor_gen(b0, gl_mul(gl_sub(1, b1), b2)));
                }
                k.constraint(gl_sub(acc, LV(reg_app(x, y) + half)));
            }
        }
    // A''[0, 0] bit decomposition, then the iota output A'''[0, 0] = A''[0, 0] xor RC
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        gl_t acc = 0;
#pragma unroll 4
        for (int z = 32 * half + 31; z >= 32 * half; z--) acc = gl_add(gl_add(acc, acc), LV(APP00_BITS + z));
        k.constraint(gl_sub(acc, LV(reg_app(0, 0) + half)));
    }
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        gl_t acc = 0;
#pragma unroll 1
        for (int z = 32 * half + 31; z >= 32 * half; z--) {
            gl_t rc = z == 0 ? rc0 : z == 1 ? rc1 : z == 3 ? rc3 : z == 7 ? rc7 : z == 15 ? rc15 : z == 31 ? rc31 : z == 63 ? rc63 : 0;
            acc = gl_add(gl_add(acc, acc), xor_gen(LV(APP00_BITS + z), rc));
        }
        k.constraint(gl_sub(acc, LV(APPP00 + half)));
    }
    // this round's output is the next row's input unless this is the last round
    gl_t not_last_t = gl_mul(not_final, k.z_last);
#pragma unroll 1
    for (int x = 0; x < 5; x++)
#pragma unroll 1
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int half = 0; half < 2; half++)
                k.constraint(gl_mul(not_last_t, gl_sub(LV(reg_appp(x, y) + half), NV(reg_a(x, y) + half))));
#undef LV
#undef NV
}

// The same constraints for SHORT Keccak tables (a 2^16-cycle segment has 2^11 rows: 4096 quotient points = 64 waves for 797 constraints
// over 2431 columns, 2.9 ms of one wave per sixteenth SIMD): 25 threads per point.  Part p = 5x + y evaluates, in emission order, C'[x, z]
// and the diff check for 13 (12) values of z, and the A, A'' and transition constraints of lane (x, y); part 0 also takes the three flag
// constraints and the four A''[0, 0] / iota ones.  A part runs the same Horner recurrence acc = acc * alpha + c over the constraints it
// owns and jumps over the others by multiplying with alpha^gap (apw[a * (KECCAK_NUM_CONSTRAINTS + 1) + e] = alpha_a^e): the 25 partial
// values of a point add up to what eval_keccak_constraints leaves in acc -- the same polynomial in alpha, term by term, and field addition
// is exact.
#define KECCAK_NUM_CONSTRAINTS 797
#define KECCAK_CONSTRAINT_PARTS 25
template <int NA>
__device__ void eval_keccak_constraints_part(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k, int part,
                                             const gl_t* __restrict__ apw) {
    using namespace kk;
    const gl_t* __restrict__ nv = lv + dnext;
#define LV(c) lv[(size_t)(c) * cs]
#define NV(c) nv[(size_t)(c) * cs]
    const int x = part / 5, y = part % 5, z0 = 13 * y, z1 = z0 + 13 < 64 ? z0 + 13 : 64;
    int pos = 0;   // emission index of the next constraint
    auto jump = [&](int to) {
        const int gap = to - pos;
        if (gap) {
#pragma unroll
            for (int a = 0; a < NA; a++) k.acc[a] = gl_mul(k.acc[a], apw[a * (KECCAK_NUM_CONSTRAINTS + 1) + gap]);
        }
        pos = to;
    };
    const gl_t final_step = LV(23);
    const gl_t not_final = gl_sub(1, final_step);
    gl_t rc0 = 0, rc1 = 0, rc3 = 0, rc7 = 0, rc15 = 0, rc31 = 0, rc63 = 0;
    if (part == 0) {
        k.constraint(gl_mul(final_step, gl_sub(final_step, 1)));
        k.constraint(gl_mul(not_final, final_step));
        gl_t sum_flags = 0;
        constexpr const uint64_t (&RC)[24] = KECCAK_RC_DEV;
#pragma unroll
        for (int r = 0; r < 24; r++) {
            gl_t f = LV(r);
            sum_flags = gl_add(sum_flags, f);
            if (RC[r] & 1) rc0 = gl_add(rc0, f);
            if (RC[r] >> 1 & 1) rc1 = gl_add(rc1, f);
            if (RC[r] >> 3 & 1) rc3 = gl_add(rc3, f);
            if (RC[r] >> 7 & 1) rc7 = gl_add(rc7, f);
            if (RC[r] >> 15 & 1) rc15 = gl_add(rc15, f);
            if (RC[r] >> 31 & 1) rc31 = gl_add(rc31, f);
            if (RC[r] >> 63 & 1) rc63 = gl_add(rc63, f);
        }
        k.constraint(gl_mul(gl_mul(sum_flags, not_final), gl_sub(NV(TIMESTAMP), LV(TIMESTAMP))));
        pos = 3;
    }
    // C'[x, z], z0 <= z < z1: constraints 3 + 64 x + z
    jump(3 + 64 * x + z0);
#pragma unroll 1
    for (int z = z0; z < z1; z++) {
        gl_t v = xor_gen(LV(reg_c(x, z)), xor_gen(LV(reg_c(mod5(x + 4), z)), LV(reg_c(mod5(x + 1), (z + 63) & 63))));
        k.constraint(gl_sub(LV(reg_cp(x, z)), v));
    }
    pos += z1 - z0;
    // A[x, y] limbs: constraints 323 + 2 (5 x + y) + half
    jump(323 + 2 * part);
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        gl_t acc = 0;
#pragma unroll 2
        for (int z = 32 * half + 31; z >= 32 * half; z--)
            acc = gl_add(gl_add(acc, acc), xor_gen(LV(reg_ap(x, y, z)), xor_gen(LV(reg_c(x, z)), LV(reg_cp(x, z)))));
        k.constraint(gl_sub(acc, LV(reg_a(x, y) + half)));
    }
    pos += 2;
    // diff check: constraints 373 + 64 x + z
    jump(373 + 64 * x + z0);
#pragma unroll 1
    for (int z = z0; z < z1; z++) {
        gl_t sum = LV(reg_ap(x, 0, z));
#pragma unroll
        for (int i = 1; i < 5; i++) sum = gl_add(sum, LV(reg_ap(x, i, z)));
        gl_t diff = gl_sub(sum, LV(reg_cp(x, z)));
        k.constraint(gl_mul(gl_mul(diff, gl_sub(diff, 2)), gl_sub(diff, 4)));
    }
    pos += z1 - z0;
    // A''[x, y]: constraints 693 + 2 (5 x + y) + half
    jump(693 + 2 * part);
    {
        int x1 = mod5(x + 1), x2 = mod5(x + 2);
        int a0 = (x + 3 * y) % 5, a1 = (x1 + 3 * y) % 5, a2 = (x2 + 3 * y) % 5;
        int base0 = reg_ap(a0, x, 0), base1 = reg_ap(a1, x1, 0), base2 = reg_ap(a2, x2, 0);
        int r0 = 64 - ROT[a0][x], r1 = 64 - ROT[a1][x1], r2 = 64 - ROT[a2][x2];
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            gl_t acc = 0;
#pragma unroll 2
            for (int z = 32 * half + 31; z >= 32 * half; z--) {
                gl_t b0 = LV(base0 + ((z + r0) & 63)), b1 = LV(base1 + ((z + r1) & 63)), b2 = LV(base2 + ((z + r2) & 63));
                acc = gl_add(gl_add(acc, acc), xor_gen(b0, gl_mul(gl_sub(1, b1), b2)));
            }
            k.constraint(gl_sub(acc, LV(reg_app(x, y) + half)));
        }
    }
    pos += 2;
    if (part == 0) {
        // A''[0, 0] bit decomposition and the iota output: constraints 743 .. 746
        jump(743);
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            gl_t acc = 0;
#pragma unroll 4
            for (int z = 32 * half + 31; z >= 32 * half; z--) acc = gl_add(gl_add(acc, acc), LV(APP00_BITS + z));
            k.constraint(gl_sub(acc, LV(reg_app(0, 0) + half)));
        }
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            gl_t acc = 0;
#pragma unroll 1
            for (int z = 32 * half + 31; z >= 32 * half; z--) {
                gl_t rc = z == 0 ? rc0 : z == 1 ? rc1 : z == 3 ? rc3 : z == 7 ? rc7 : z == 15 ? rc15 : z == 31 ? rc31 : z == 63 ? rc63 : 0;
                acc = gl_add(gl_add(acc, acc), xor_gen(LV(APP00_BITS + z), rc));
            }
            k.constraint(gl_sub(acc, LV(APPP00 + half)));
        }
        pos += 4;
    }
    // transitions: constraints 747 + 2 (5 x + y) + half
    jump(747 + 2 * part);
    {
        const gl_t not_last_t = gl_mul(not_final, k.z_last);
#pragma unroll
        for (int half = 0; half < 2; half++) k.constraint(gl_mul(not_last_t, gl_sub(LV(reg_appp(x, y) + half), NV(reg_a(x, y) + half))));
    }
    pos += 2;
    jump(KECCAK_NUM_CONSTRAINTS);
#undef LV
#undef NV
}

// PoseidonSpongeStark (poseidon_sponge/poseidon_sponge_stark.rs:383-478; columns poseidon_sponge/columns.rs:17-66)
template <int NA>
__device__ void eval_poseidon_sponge_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    const gl_t* __restrict__ nv = lv + dnext;
    enum { FULL = 0, CONTEXT = 1, SEGMENT = 2, TIMESTAMP = 11, LEN = 12, ABSORBED = 13, FINAL_LEN = 14, ORIG = 46, PARTIAL = 98, DIGEST = 106 };
    gl_t full = lv[FULL];
    k.constraint(gl_mul(full, gl_sub(full, 1)));
    gl_t is_final = 0, next_final = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        is_final = gl_add(is_final, lv[(size_t)(FINAL_LEN + i) * cs]);
        next_final = gl_add(next_final, nv[(size_t)(FINAL_LEN + i) * cs]);
    }
    k.constraint(gl_mul(is_final, gl_sub(is_final, 1)));
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        gl_t f = lv[(size_t)(FINAL_LEN + i) * cs];
        k.constraint(gl_mul(f, gl_sub(f, 1)));
    }
    k.constraint(gl_mul(is_final, full));
    gl_t absorbed = lv[(size_t)ABSORBED * cs];
    k.first_row(absorbed);
#pragma unroll 2
    for (int i = 0; i < 12; i++) k.first_row(lv[(size_t)(ORIG + i) * cs]);  // original_rate then original_capacity
    gl_t fin_t = gl_mul(is_final, k.z_last), full_t = gl_mul(full, k.z_last);
    k.constraint(gl_mul(fin_t, nv[(size_t)ABSORBED * cs]));
#pragma unroll 2
    for (int i = 0; i < 12; i++) k.constraint(gl_mul(fin_t, nv[(size_t)(ORIG + i) * cs]));
    k.constraint(gl_mul(full_t, gl_sub(lv[(size_t)CONTEXT * cs], nv[(size_t)CONTEXT * cs])));
    k.constraint(gl_mul(full_t, gl_sub(lv[(size_t)SEGMENT * cs], nv[(size_t)SEGMENT * cs])));
    k.constraint(gl_mul(full_t, gl_sub(lv[(size_t)TIMESTAMP * cs], nv[(size_t)TIMESTAMP * cs])));
#pragma unroll 2
    for (int i = 0; i < 4; i++) k.constraint(gl_mul(full_t, gl_sub(nv[(size_t)(ORIG + i) * cs], lv[(size_t)(DIGEST + i) * cs])));
#pragma unroll 2
    for (int i = 0; i < 8; i++)  // rate words 4..7 then the 4 capacity words are contiguous in both views
        k.constraint(gl_mul(full_t, gl_sub(nv[(size_t)(ORIG + 4 + i) * cs], lv[(size_t)(PARTIAL + i) * cs])));
    k.constraint(gl_mul(full_t, gl_sub(gl_add(absorbed, 32), nv[(size_t)ABSORBED * cs])));
    gl_t is_dummy = gl_sub(gl_sub(1, full), is_final);
    k.transition(gl_mul(is_dummy, gl_add(nv[FULL], next_final)));
    gl_t offset = gl_sub(lv[(size_t)LEN * cs], absorbed);
#pragma unroll 4
    for (int i = 0; i < 32; i++) k.constraint(gl_mul(lv[(size_t)(FINAL_LEN + i) * cs], gl_sub(offset, (gl_t)i)));
}

// ShaExtendStark (sha_extend/sha_extend_stark.rs:238-317; rotate_right.rs:29-62, shift_right.rs:29-60, wrapping_add_4.rs:35-78)
__device__ __forceinline__ gl_t sha_le4(const gl_t* __restrict__ b, size_t cs) {
    return gl_add(gl_add(b[0], gl_mul(b[cs], 1u << 8)), gl_add(gl_mul(b[2 * cs], 1u << 16), gl_mul(b[3 * cs], 1u << 24)));
}
template <int NA>
__device__ __forceinline__ void sha_rot_constraints(const gl_t* __restrict__ in, const gl_t* __restrict__ op, size_t cs, unsigned r,
                                                    bool is_shift, consumer_t<NA>& k) {
    gl_t out = sha_le4(op, cs), inv = sha_le4(in, cs), shift = op[4 * cs], carry = op[5 * cs];
    if (is_shift) k.constraint(gl_sub(out, shift));
    else k.constraint(gl_sub(gl_sub(out, gl_mul(carry, (gl_t)1 << (32 - r))), shift));
    k.constraint(gl_sub(gl_sub(inv, gl_mul(shift, (gl_t)1 << r)), carry));
}
template <int NA>
__device__ void eval_sha_extend_constraints(const gl_t* __restrict__ lv, size_t cs, consumer_t<NA>& k) {
    sha_rot_constraints<NA>(lv + 8 * cs, lv + 40 * cs, cs, 7, false, k);
    sha_rot_constraints<NA>(lv + 8 * cs, lv + 46 * cs, cs, 18, false, k);
    sha_rot_constraints<NA>(lv + 12 * cs, lv + 52 * cs, cs, 17, false, k);
    sha_rot_constraints<NA>(lv + 12 * cs, lv + 58 * cs, cs, 19, false, k);
    sha_rot_constraints<NA>(lv + 8 * cs, lv + 70 * cs, cs, 3, true, k);
    sha_rot_constraints<NA>(lv + 12 * cs, lv + 64 * cs, cs, 10, true, k);
    gl_t real = lv[77 * cs];
    const gl_t *a = lv + 36 * cs, *b = lv + 20 * cs, *c = lv + 28 * cs, *d = lv + 16 * cs, *cy = lv + 4 * cs;
    gl_t c0 = cy[0], c1 = cy[cs], c2 = cy[2 * cs], c3 = cy[3 * cs];
    k.constraint(gl_mul(gl_mul(c0, gl_sub(1, c0)), real));
    k.constraint(gl_mul(gl_mul(c1, gl_sub(1, c1)), real));
    k.constraint(gl_mul(gl_mul(c2, gl_sub(1, c2)), real));
    k.constraint(gl_mul(gl_mul(c3, gl_sub(1, c3)), real));
    k.constraint(gl_mul(gl_sub(gl_add(gl_add(c0, c1), gl_add(c2, c3)), 1), real));
    gl_t carry = gl_add(gl_add(c1, gl_add(c2, c2)), gl_mul(c3, 3));
    gl_t sum = 0;
#pragma unroll
    for (int i = 3; i >= 0; i--)
        sum = gl_add(gl_mul(sum, 1u << 8), gl_add(gl_add(a[(size_t)i * cs], b[(size_t)i * cs]), gl_add(c[(size_t)i * cs], d[(size_t)i * cs])));
    k.constraint(gl_mul(gl_sub(gl_sub(sum, gl_mul(carry, (gl_t)1 << 32)), sha_le4(lv, cs)), real));
}

// ShaExtendSpongeStark (sha_extend_sponge/sha_extend_sponge_stark.rs:220-330); NUM_CHANNELS = 10 (cpu/membus.rs:10-32)
template <int NA>
__device__ void eval_sha_extend_sponge_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    const gl_t* __restrict__ nv = lv + dnext;
    gl_t sum = 0, lidx = 0, nidx = 0;
#pragma unroll 4
    for (int i = 0; i < 48; i++) {
        gl_t f = lv[(size_t)i * cs];
        k.constraint(gl_mul(f, gl_sub(f, 1)));
        sum = gl_add(sum, f);
        lidx = gl_add(lidx, gl_mul(f, (gl_t)i));
        nidx = gl_add(nidx, gl_mul(nv[(size_t)i * cs], (gl_t)i));
    }
    gl_t is_final = lv[(size_t)47 * cs];
    k.constraint(gl_mul(is_final, gl_sub(is_final, 1)));
    gl_t g = gl_mul(sum, gl_sub(1, is_final));
    k.constraint(gl_mul(g, gl_sub(gl_sub(nv[(size_t)75 * cs], lv[(size_t)75 * cs]), 20)));
    k.constraint(gl_mul(g, gl_sub(gl_sub(nidx, lidx), 1)));
#pragma unroll
    for (int i = 0; i < 5; i++)  // the four input addresses, then the output address
        k.constraint(gl_mul(g, gl_sub(gl_sub(nv[(size_t)(68 + i) * cs], lv[(size_t)(68 + i) * cs]), 4)));
    gl_t a16 = lv[(size_t)70 * cs];
    k.constraint(gl_mul(sum, gl_sub(gl_sub(lv[(size_t)68 * cs], a16), 4)));
    k.constraint(gl_mul(sum, gl_sub(gl_sub(lv[(size_t)69 * cs], a16), 56)));
    k.constraint(gl_mul(sum, gl_sub(gl_sub(lv[(size_t)71 * cs], a16), 36)));
    k.constraint(gl_mul(sum, gl_sub(gl_sub(lv[(size_t)72 * cs], a16), 64)));
}

// ShaCompressStark (sha_compress/sha_compress_stark.rs:402-606) and ShaCompressSpongeStark (sha_compress_sponge_stark.rs:233-268)
// wrapping add of NIN byte quadruples (column offsets in[]) into op = value[4], carry[NC]; every constraint times gate
template <int NA, int NIN, int NC>
__device__ __forceinline__ void sha_wadd_constraints(const gl_t* __restrict__ lv, size_t cs, const int (&in)[NIN], int op, gl_t gate,
                                                     consumer_t<NA>& k) {
    gl_t csum = 0, carry = 0;
#pragma unroll
    for (int i = 0; i < NC; i++) {
        gl_t c = lv[(size_t)(op + 4 + i) * cs];
        k.constraint(gl_mul(gate, gl_mul(c, gl_sub(1, c))));
        csum = gl_add(csum, c);
        if (i) carry = gl_add(carry, gl_mul(c, (gl_t)i));
    }
    k.constraint(gl_mul(gate, gl_sub(csum, 1)));
    gl_t sum = 0;
#pragma unroll
    for (int b = 3; b >= 0; b--) {
        gl_t s = lv[(size_t)(in[0] + b) * cs];
#pragma unroll
        for (int q = 1; q < NIN; q++) s = gl_add(s, lv[(size_t)(in[q] + b) * cs]);
        sum = gl_add(gl_mul(sum, 1u << 8), s);
    }
    k.constraint(gl_mul(gate, gl_sub(gl_sub(sum, gl_mul(carry, (gl_t)1 << 32)), sha_le4(lv + (size_t)op * cs, cs))));
}
template <int NA>
__device__ void eval_sha_compress_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    const gl_t* __restrict__ nv = lv + dnext;
    enum { ROUND = 159, TIMESTAMP = 146, W_VIRT = 149 };
    gl_t is_final = lv[(size_t)(ROUND + 64) * cs];
    k.constraint(gl_mul(is_final, gl_sub(is_final, 1)));
    gl_t sum = is_final, kb[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int j = 0; j < 64; j++) {
        gl_t f = lv[(size_t)(ROUND + j) * cs];
        sum = gl_add(sum, f);
        uint32_t kc = SHA256_K_DEV[j];
#pragma unroll
        for (int i = 0; i < 4; i++) kb[i] = gl_add(kb[i], gl_mul(f, (kc >> (8 * i)) & 0xFF));
    }
    k.constraint(gl_mul(sum, gl_sub(sum, 1)));
    gl_t g = gl_mul(sum, gl_sub(1, is_final));
#pragma unroll
    for (int i = 0; i < 4; i++) k.constraint(gl_mul(g, gl_sub(lv[(size_t)(40 + i) * cs], kb[i])));
    sha_rot_constraints<NA>(lv + 16 * cs, lv + 92 * cs, cs, 6, false, k);
    sha_rot_constraints<NA>(lv + 16 * cs, lv + 98 * cs, cs, 11, false, k);
    sha_rot_constraints<NA>(lv + 16 * cs, lv + 104 * cs, cs, 25, false, k);
    sha_rot_constraints<NA>(lv, lv + 110 * cs, cs, 2, false, k);
    sha_rot_constraints<NA>(lv, lv + 116 * cs, cs, 13, false, k);
    sha_rot_constraints<NA>(lv, lv + 122 * cs, cs, 22, false, k);
#pragma unroll
    for (int i = 0; i < 4; i++) k.constraint(gl_mul(sum, gl_sub(gl_add(lv[(size_t)(16 + i) * cs], lv[(size_t)(32 + i) * cs]), 255)));
    { const int in[5] = {28, 48, 60, 40, 36}; sha_wadd_constraints<NA, 5, 5>(lv, cs, in, 150, sum, k); }  // temp1 = h + s_1 + ch + k_i + w_i
    { const int in[2] = {68, 88}; sha_wadd_constraints<NA, 2, 2>(lv, cs, in, 128, sum, k); }              // temp2 = s_0 + maj
    { const int in[2] = {12, 150}; sha_wadd_constraints<NA, 2, 2>(lv, cs, in, 134, sum, k); }             // d + temp1
    { const int in[2] = {150, 128}; sha_wadd_constraints<NA, 2, 2>(lv, cs, in, 140, sum, k); }            // temp1 + temp2
    k.constraint(gl_mul(g, gl_sub(nv[(size_t)TIMESTAMP * cs], lv[(size_t)TIMESTAMP * cs])));
    k.constraint(gl_mul(g, gl_sub(gl_sub(nv[(size_t)W_VIRT * cs], lv[(size_t)W_VIRT * cs]), 4)));
#pragma unroll
    for (int i = 0; i < 4; i++) k.constraint(gl_mul(g, gl_sub(lv[(size_t)(140 + i) * cs], nv[(size_t)i * cs])));
#pragma unroll 1
    for (int w = 0; w < 3; w++)
#pragma unroll
        for (int i = 0; i < 4; i++) k.constraint(gl_mul(g, gl_sub(lv[(size_t)(4 * w + i) * cs], nv[(size_t)(4 * (w + 1) + i) * cs])));
#pragma unroll
    for (int i = 0; i < 4; i++) k.constraint(gl_mul(g, gl_sub(lv[(size_t)(134 + i) * cs], nv[(size_t)(16 + i) * cs])));
#pragma unroll 1
    for (int w = 4; w < 7; w++)
#pragma unroll
        for (int i = 0; i < 4; i++) k.constraint(gl_mul(g, gl_sub(lv[(size_t)(4 * w + i) * cs], nv[(size_t)(4 * (w + 1) + i) * cs])));
}
template <int NA>
__device__ void eval_sha_compress_sponge_constraints(const gl_t* __restrict__ lv, size_t cs, consumer_t<NA>& k) {
    gl_t real = lv[(size_t)126 * cs];
    k.constraint(gl_mul(real, gl_sub(real, 1)));
#pragma unroll
    for (int i = 0; i < 7; i++) k.constraint(gl_mul(real, gl_sub(gl_sub(lv[(size_t)(113 + i) * cs], lv[(size_t)(112 + i) * cs]), 4)));
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        const int in[2] = {4 * i, 32 + 4 * i};
        sha_wadd_constraints<NA, 2, 2>(lv, cs, in, 64 + 6 * i, real, k);
    }
}

// MemoryStark (memory/memory_stark.rs:253-341; columns memory/columns.rs, VALUE_LIMBS = 1)
template <int NA>
__device__ void eval_memory_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    const gl_t* __restrict__ nv = lv + dnext;
    enum { FILTER = 0, TIMESTAMP = 1, IS_READ = 2, CONTEXT = 3, SEGMENT = 4, VIRTUAL = 5, VALUE = 6, CFC = 7, SFC = 8, VFC = 9, RANGE_CHECK = 10 };
    gl_t filter = lv[FILTER];
    k.constraint(gl_mul(filter, gl_sub(filter, 1)));
    gl_t cfc = lv[CFC * cs], sfc = lv[SFC * cs], vfc = lv[VFC * cs];
    gl_t unchanged = gl_sub(gl_sub(gl_sub(1, cfc), sfc), vfc);
    k.constraint(gl_mul(cfc, gl_sub(1, cfc)));
    k.constraint(gl_mul(sfc, gl_sub(1, sfc)));
    k.constraint(gl_mul(vfc, gl_sub(1, vfc)));
    k.constraint(gl_mul(unchanged, gl_sub(1, unchanged)));
    gl_t dctx = gl_sub(nv[CONTEXT * cs], lv[CONTEXT * cs]), dseg = gl_sub(nv[SEGMENT * cs], lv[SEGMENT * cs]);
    gl_t dvirt = gl_sub(nv[VIRTUAL * cs], lv[VIRTUAL * cs]);
    k.transition(gl_mul(sfc, dctx));
    k.transition(gl_mul(vfc, dctx));
    k.transition(gl_mul(vfc, dseg));
    k.transition(gl_mul(unchanged, dctx));
    k.transition(gl_mul(unchanged, dseg));
    k.transition(gl_mul(unchanged, dvirt));
    gl_t computed = gl_add(gl_add(gl_mul(cfc, gl_sub(dctx, 1)), gl_mul(sfc, gl_sub(dseg, 1))),
                           gl_add(gl_mul(vfc, gl_sub(dvirt, 1)), gl_mul(unchanged, gl_sub(nv[TIMESTAMP * cs], lv[TIMESTAMP * cs]))));
    k.transition(gl_sub(lv[RANGE_CHECK * cs], computed));
    k.transition(gl_mul(gl_mul(nv[IS_READ * cs], unchanged), gl_sub(nv[VALUE * cs], lv[VALUE * cs])));
}

// A table's own logUp lookups (eval_packed_lookups_generic lookup.rs:138-198; helper-column checks eval_helper_columns
// cross_table_lookup.rs:1006-1058 with beta = 1, gamma = challenge, no filters).  Passed by value: a few words.
struct lookup_dev {
    uint32_t nlookups, nch;
    struct { uint32_t ncols, col_off, table_col, freq_col; } lk[2];
    uint32_t cols[24];
};
template <int NA>
__device__ void eval_lookup_constraints(const lookup_dev& d, const gl_t* __restrict__ challenges /* nch: this segment's */,
                                        const gl_t* __restrict__ lv, size_t N, const gl_t* __restrict__ aux, size_t j, size_t jn,
                                        consumer_t<NA>& k) {
    uint32_t start = 0;
    for (uint32_t l = 0; l < d.nlookups; l++) {
        const uint32_t ncols = d.lk[l].ncols, nh = (ncols + 1) / 2;
        const uint32_t* cols = d.cols + d.lk[l].col_off;
        for (uint32_t c = 0; c < d.nch; c++) {
            const gl_t ch = challenges[c];
            gl_t hsum = 0;
            for (uint32_t q = 0; q < nh; q++) {
                gl_t h = aux[(size_t)(start + q) * N + j];
                gl_t combin0 = gl_add(lv[(size_t)cols[2 * q] * N], ch);
                if (2 * q + 1 < ncols) {
                    gl_t combin1 = gl_add(lv[(size_t)cols[2 * q + 1] * N], ch);
                    k.constraint(gl_sub(gl_sub(gl_mul(gl_mul(combin1, combin0), h), combin1), combin0));
                } else {
                    k.constraint(gl_sub(gl_mul(combin0, h), 1));
                }
                hsum = gl_add(hsum, h);
            }
            gl_t z = aux[(size_t)(start + nh) * N + j], next_z = aux[(size_t)(start + nh) * N + jn];
            gl_t table_ch = gl_add(lv[(size_t)d.lk[l].table_col * N], ch);
            gl_t y = gl_sub(gl_mul(hsum, table_ch), lv[(size_t)d.lk[l].freq_col * N]);
            k.first_row(z);
            k.constraint(gl_sub(gl_mul(gl_sub(next_z, z), table_ch), y));
            start += nh + 1;
        }
    }
}

// ArithmeticStark (arithmetic/arithmetic_stark.rs:214-240 and the nine operation modules it calls; N_LIMBS = 2).  The oracle's
// TNAME(eval_arithmetic) documents the column map; this is the same sequence of constraints on LDE columns.
namespace arith {
enum { ADD = 0, ADDU, ADDI, ADDIU, SUB, SUBU, MULT, MULTU, MUL, DIV, DIVU, SLLV, SRLV, SRAV, SLL, SRL, SRA, SLT, SLTU, SLTI, SLTIU, LUI, MFHI, MTHI,
       MFLO, MTLO, IN0 = 26, IN1 = 28, IN2 = 30, OUT = 32, AUX0 = 34, AUX1 = 36, AUX2 = 38, QUOT_ABS = 40, REM_ABS = 42, MULT_AUX_LO = 36,
       MULT_AUX_HI = 40, RANGE_COUNTER = 44, RC_FREQ = 45, AUX_EXTRA = 46, NV_RED = 26, NV_MOD_IS_ZERO = 28, NV_AUX_LO = 29, NV_AUX_HI = 32,
       NV_DENOM_IS_ZERO = 35 };
constexpr gl_t INV_65536 = 18446462594437939201ULL;  // addcy.rs:41
#undef ZKM_CONST
#define ZKM_CONST static __device__ const
#include "arith_constants.inc"
#undef ZKM_CONST

// local / next row accessors of one thread
struct frame {
    const gl_t* __restrict__ lv;
    const gl_t* __restrict__ nv;
    size_t cs;
    __device__ __forceinline__ gl_t L(int c) const { return lv[(size_t)c * cs]; }
    __device__ __forceinline__ gl_t N(int c) const { return nv[(size_t)c * cs]; }
};
template <int NA>
__device__ __forceinline__ void emit(consumer_t<NA>& k, gl_t c, bool two_row) {
    if (two_row) k.transition(c); else k.constraint(c);
}
// eval_packed_generic_addcy addcy.rs:43-93 (x + y == z + cy 2^32)
template <int NA>
__device__ void addcy(consumer_t<NA>& k, gl_t filter, const gl_t (&x)[2], const gl_t (&y)[2], const gl_t (&z)[2], const gl_t (&given_cy)[2],
                      bool two_row) {
    gl_t cy = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        gl_t t = gl_sub(gl_add(gl_add(cy, x[i]), y[i]), z[i]);
        emit(k, gl_mul(gl_mul(filter, t), gl_sub(65536, t)), two_row);
        cy = gl_mul(t, INV_65536);
    }
    if (two_row) {
        k.transition(gl_mul(filter, gl_sub(cy, given_cy[0])));
        k.transition(gl_mul(filter, given_cy[1]));
    } else {
        k.constraint(gl_mul(gl_mul(filter, given_cy[0]), gl_sub(given_cy[0], 1)));
        k.constraint(gl_mul(filter, gl_sub(cy, given_cy[0])));
        k.constraint(gl_mul(filter, given_cy[1]));
    }
}
// eval_packed_generic_mul mul.rs:109-135
template <int NA>
__device__ void mul(const frame& f, gl_t filter, int l, int r, consumer_t<NA>& k) {
    gl_t aux[2];
#pragma unroll
    for (int i = 0; i < 2; i++) aux[i] = gl_sub(gl_add(f.L(AUX0 + i), gl_mul(f.L(AUX1 + i), 65536)), 1 << 20);
    gl_t l0 = f.L(l), l1 = f.L(l + 1), r0 = f.L(r), r1 = f.L(r + 1);
    gl_t c0 = gl_add(gl_sub(gl_mul(l0, r0), f.L(OUT)), gl_mul(aux[0], 65536));
    gl_t c1 = gl_sub(gl_sub(gl_add(gl_mul(l0, r1), gl_mul(l1, r0)), f.L(OUT + 1)), gl_sub(aux[0], gl_mul(aux[1], 65536)));
    k.constraint(gl_mul(filter, c0));
    k.constraint(gl_mul(filter, c1));
}
// eval_packed_generic_mult_helper mult.rs:236-262
template <int NA>
__device__ void mult_helper(const frame& f, gl_t filter, const gl_t (&l)[4], const gl_t (&r)[4], consumer_t<NA>& k) {
    gl_t aux[4], cp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) aux[i] = gl_sub(gl_add(f.L(MULT_AUX_LO + i), gl_mul(f.L(MULT_AUX_HI + i), 65536)), 1 << 20);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        gl_t s = 0;
#pragma unroll
        for (int i = 0; i <= d; i++) s = gl_add(s, gl_mul(l[i], r[d - i]));
        cp[d] = gl_sub(s, f.L(OUT + d));
    }
    cp[0] = gl_add(cp[0], gl_mul(aux[0], 65536));
#pragma unroll
    for (int d = 1; d < 4; d++) cp[d] = gl_sub(cp[d], gl_sub(aux[d - 1], gl_mul(aux[d], 65536)));
#pragma unroll
    for (int d = 0; d < 4; d++) k.constraint(gl_mul(filter, cp[d]));
}
// eval_packed_div_helper div.rs:509-541 (modular_constr_poly :325-380, check_reduced :300-323)
template <int NA>
__device__ void div_helper(const frame& f, consumer_t<NA>& k, gl_t filter, int num, int den, int quo, int rem) {
    k.last_row(filter);
    gl_t miz = f.N(NV_MOD_IS_ZERO);
    k.transition(gl_mul(filter, gl_sub(gl_mul(miz, miz), miz)));
    gl_t modulus[2] = {f.L(den), f.L(den + 1)}, output[2] = {f.L(rem), f.L(rem + 1)};
    k.transition(gl_mul(gl_mul(filter, gl_add(modulus[0], modulus[1])), miz));
    modulus[0] = gl_add(modulus[0], miz);
    gl_t ddz = f.N(NV_DENOM_IS_ZERO);
    gl_t shr_div = gl_add(gl_add(gl_add(f.L(DIV), f.L(DIVU)), gl_add(f.L(SRL), f.L(SRLV))), gl_add(f.L(SRA), f.L(SRAV)));
    k.transition(gl_mul(filter, gl_sub(gl_mul(miz, shr_div), ddz)));
    output[0] = gl_add(output[0], ddz);
    {
        const gl_t less[2] = {gl_sub(1, gl_mul(miz, shr_div)), 0}, red[2] = {f.N(NV_RED), f.N(NV_RED + 1)};
        addcy(k, filter, modulus, red, output, less, true);
    }
    output[0] = gl_sub(output[0], ddz);
    const gl_t q0 = f.L(quo), q1 = f.L(quo + 1);  // quot = [q0, q1, 0, 0]
    gl_t prod[4] = {gl_mul(q0, modulus[0]), gl_add(gl_mul(q0, modulus[1]), gl_mul(q1, modulus[0])), gl_mul(q1, modulus[1]), 0};
    k.transition(0);  // filter * prod[4], which is identically zero for a two-limb quotient
    gl_t cp[4] = {gl_add(prod[0], output[0]), gl_add(prod[1], output[1]), prod[2], prod[3]}, aux[4];
#pragma unroll
    for (int i = 0; i < 3; i++) aux[i] = gl_add(gl_sub(f.N(NV_AUX_LO + i), 1 << 20), gl_mul(f.N(NV_AUX_HI + i), 65536));
    aux[3] = 0;
    cp[0] = gl_sub(cp[0], gl_mul(aux[0], 65536));
#pragma unroll
    for (int d = 1; d < 4; d++) cp[d] = gl_add(cp[d], gl_sub(aux[d - 1], gl_mul(aux[d], 65536)));
    cp[0] = gl_sub(cp[0], f.L(num));
    cp[1] = gl_sub(cp[1], f.L(num + 1));
#pragma unroll
    for (int d = 0; d < 4; d++) k.transition(gl_mul(filter, cp[d]));
}
// check_abs of eval_packed_div div.rs:400-430
template <int NA>
__device__ gl_t check_abs(const frame& f, consumer_t<NA>& k, gl_t filter, int input, int abs_col, int sum_col, int neg_col, int borrow_col) {
    gl_t is_neg = f.N(neg_col);
    k.transition(gl_mul(gl_mul(filter, is_neg), gl_sub(1, is_neg)));
    k.transition(gl_mul(filter, gl_sub(gl_sub(gl_add(f.L(input + 1), 32768), f.N(sum_col)), gl_mul(is_neg, 65536))));
    gl_t b = f.N(borrow_col);
    k.transition(gl_mul(gl_mul(filter, b), gl_sub(1, b)));
    const gl_t neg_in[2] = {gl_sub(gl_mul(b, 65536), f.L(input)), gl_sub(gl_sub(65536, f.L(input + 1)), b)};
#pragma unroll
    for (int i = 0; i < 2; i++)
        k.transition(gl_mul(filter, gl_sub(gl_add(gl_mul(is_neg, neg_in[i]), gl_mul(gl_sub(1, is_neg), f.L(input + i))), f.L(abs_col + i))));
    return is_neg;
}
}  // namespace arith

template <int NA>
__device__ void eval_arithmetic_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    using namespace arith;
    const frame f{lv, lv + dnext, cs};
    gl_t rc1 = f.L(RANGE_COUNTER), incr = gl_sub(f.N(RANGE_COUNTER), rc1);
    k.first_row(rc1);
    k.transition(gl_sub(gl_mul(incr, incr), incr));
    k.last_row(gl_sub(rc1, 65535));
    mul(f, f.L(MUL), IN0, IN1, k);
    {   // mult.rs:115-234: signed (sign-extended operands), then unsigned
        gl_t ff = f.L(MULT), l[4], r[4];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            gl_t is_neg = f.L(AUX_EXTRA + s), in_hi = f.L((s ? IN1 : IN0) + 1), sum = f.L(IN2 + s);
            k.constraint(gl_mul(gl_mul(ff, is_neg), gl_sub(1, is_neg)));
            k.constraint(gl_mul(ff, gl_sub(gl_sub(gl_add(in_hi, 32768), sum), gl_mul(is_neg, 65536))));
            gl_t pad = gl_mul(is_neg, 65535);
            if (s) { r[0] = f.L(IN1); r[1] = in_hi; r[2] = r[3] = pad; }
            else { l[0] = f.L(IN0); l[1] = in_hi; l[2] = l[3] = pad; }
        }
        mult_helper(f, ff, l, r, k);
        const gl_t lu[4] = {l[0], l[1], 0, 0}, ru[4] = {r[0], r[1], 0, 0};
        mult_helper(f, f.L(MULTU), lu, ru, k);
    }
    {   // addcy.rs:142-160 (ADDU / SUBU rows are not constrained by the reference)
        const gl_t in0[2] = {f.L(IN0), f.L(IN0 + 1)}, in1[2] = {f.L(IN1), f.L(IN1 + 1)}, out[2] = {f.L(OUT), f.L(OUT + 1)};
        const gl_t aux[2] = {f.L(AUX0), f.L(AUX0 + 1)};
        addcy(k, f.L(ADD), in0, in1, out, aux, false);
        addcy(k, f.L(SUB), in1, out, in0, aux, false);
        addcy(k, f.L(ADDI), in0, in1, out, aux, false);
        addcy(k, f.L(ADDIU), in0, in1, out, aux, false);
        // slt.rs:50-120: x = in1, y = diff (AUX0), z = in0, given_cy = AUX1, rd = OUT
        gl_t fl = gl_add(gl_add(f.L(SLT), f.L(SLTU)), gl_add(f.L(SLTI), f.L(SLTIU))), sign = gl_add(f.L(SLT), f.L(SLTI));
        const gl_t gc[2] = {f.L(AUX1), f.L(AUX1 + 1)};
        gl_t cy = 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            gl_t t = gl_sub(gl_add(gl_add(cy, in1[i]), aux[i]), in0[i]);
            k.constraint(gl_mul(gl_mul(fl, t), gl_sub(65536, t)));
            cy = gl_mul(t, INV_65536);
        }
        k.constraint(gl_mul(gl_mul(fl, gc[0]), gl_sub(gc[0], 1)));
        k.constraint(gl_mul(gl_mul(fl, gl_sub(cy, gc[0])), gl_sub(1, sign)));
        k.constraint(gl_mul(gl_mul(fl, gc[1]), gl_sub(gl_sub(1, cy), gc[0])));
        k.transition(gl_mul(fl, gl_sub(out[0], gc[0])));
        k.constraint(gl_mul(gl_mul(fl, gc[1]), gl_sub(1, sign)));
        k.transition(gl_mul(fl, out[1]));
    }
    mul(f, f.L(LUI), IN0, IN1, k);
    div_helper(f, k, f.L(DIVU), IN0, IN1, OUT, AUX0);
    {   // signed division div.rs:387-507
        gl_t ff = f.L(DIV);
        gl_t n0 = check_abs(f, k, ff, IN0, IN2, NV_DENOM_IS_ZERO + 1, NV_DENOM_IS_ZERO + 5, NV_DENOM_IS_ZERO + 6);
        gl_t n1 = check_abs(f, k, ff, IN1, AUX2, NV_DENOM_IS_ZERO + 2, NV_DENOM_IS_ZERO + 7, NV_DENOM_IS_ZERO + 8);
        gl_t nq = check_abs(f, k, ff, OUT, QUOT_ABS, NV_DENOM_IS_ZERO + 3, RC_FREQ + 1, RC_FREQ + 2);
        gl_t nr = check_abs(f, k, ff, AUX0, REM_ABS, NV_DENOM_IS_ZERO + 4, RC_FREQ + 3, RC_FREQ + 4);
        gl_t same = f.N(RC_FREQ + 5);
        k.transition(gl_mul(ff, gl_sub(gl_sub(gl_add(n0, n1), gl_mul(gl_mul(n0, n1), 2)), same)));
        k.transition(gl_mul(gl_mul(ff, gl_sub(nq, same)), gl_add(f.L(OUT), f.L(OUT + 1))));
        k.transition(gl_mul(gl_mul(ff, gl_sub(nr, n0)), gl_add(f.L(AUX0), f.L(AUX0 + 1))));
        div_helper(f, k, ff, IN2, AUX2, QUOT_ABS, REM_ABS);
    }
    mul(f, gl_add(f.L(SLL), f.L(SLLV)), IN1, IN2, k);
    div_helper(f, k, gl_add(f.L(SRL), f.L(SRLV)), IN1, IN2, OUT, AUX0);
    {   // sra.rs:66-133
        gl_t ff = gl_add(f.L(SRA), f.L(SRAV)), shift = f.L(IN0);
        k.transition(gl_mul(ff, f.L(IN0 + 1)));
        gl_t is_neg = f.L(AUX2 + 3);
        k.transition(gl_mul(gl_mul(ff, is_neg), gl_sub(1, is_neg)));
        k.transition(gl_mul(ff, gl_sub(gl_sub(gl_add(f.L(IN1 + 1), 32768), f.L(AUX2 + 2)), gl_mul(is_neg, 65536))));
        gl_t shift_sq = f.N(AUX2 + 2);
        k.transition(gl_mul(ff, gl_sub(shift_sq, gl_mul(shift, shift))));
        gl_t acc = 0;
#pragma unroll 1
        for (int i = 0; i < 16; i++) {
            gl_t w = i < 8 ? f.L(AUX_EXTRA + i) : f.N(AUX_EXTRA + i - 8);
            gl_t v = gl_add(gl_add(gl_mul(acc, shift_sq), gl_mul(shift, ZKM_ARITH_SIGN_EXTEND_POLY[31 - 2 * i])), ZKM_ARITH_SIGN_EXTEND_POLY[30 - 2 * i]);
            k.transition(gl_mul(ff, gl_sub(v, w)));
            acc = w;
        }
        gl_t acc_lo = f.N(AUX2), acc_hi = f.N(AUX2 + 1);
        k.transition(gl_mul(ff, gl_sub(gl_add(gl_mul(acc_hi, 65536), acc_lo), acc)));
        div_helper(f, k, ff, IN1, IN2, AUX2, AUX0);
        k.transition(gl_mul(ff, gl_sub(gl_add(f.L(AUX2), gl_mul(acc_lo, is_neg)), f.L(OUT))));
        k.transition(gl_mul(ff, gl_sub(gl_add(f.L(AUX2 + 1), gl_mul(acc_hi, is_neg)), f.L(OUT + 1))));
    }
    {   // lo_hi.rs:23-36
        gl_t ff = gl_add(gl_add(f.L(MFHI), f.L(MTHI)), gl_add(f.L(MFLO), f.L(MTLO)));
        k.constraint(gl_mul(ff, gl_sub(f.L(IN0), f.L(OUT))));
        k.constraint(gl_mul(ff, gl_sub(f.L(IN0 + 1), f.L(OUT + 1))));
    }
}

// CpuStark (cpu/cpu_stark.rs:260-285).  Emission order: bootstrap_kernel.rs:308-353, decode.rs:66-100, jumps.rs (jump / jumpi /
// jumpdirect, then branch), membus.rs:35-48, memio.rs (load :175-437, store :738-961), shift.rs:18-124, count.rs:10-75,
// syscall.rs:12-232, bits.rs:9-64, misc.rs (rdhwr, condmov, teq, ext, ror, ins, maddu).  Columns: cpu/columns/mod.rs:62-96,
// ops.rs:10-45, general.rs (union of the syscall / misc / io / logic views).  nv = lv + dnext.
// Bit-string recombinations are built from byte / prefix sums instead of one 32-term sum per candidate value.
namespace cpu {
enum { IS_BOOT = 0, CONTEXT = 2, CODE_CONTEXT = 3, PROGRAM_COUNTER = 4, NEXT_PC = 5, KERNEL = 6, BR = 40, OPC = 50, RS = 56, RT = 61, RD = 66, SHAMT = 71,
       FUNC = 76, GEN = 86, MEMIO = 188, CH = 205 };
enum { BINARY = 7, BINARY_IMM, EQ_ISZERO, LOGIC, LOGIC_IMM, MOVZ, MOVN, CLZ, CLO, SHIFT, SHIFT_IMM, KECCAK_GENERAL, JUMPS, JUMPI, JUMPDIRECT,
       BRANCH, PC_OP, GET_CONTEXT, SET_CONTEXT, EXIT_KERNEL, M_OP_LOAD, M_OP_STORE, NOP, EXT, INS, MADDU, RDHWR, SIGNEXT8, SIGNEXT16, SWAPHALF,
       TEQ, ROR, SYSCALL };
enum { USED = 0, IS_READ = 1, CTX = 2, SEG = 3, VIRT = 4, VAL = 5 };
constexpr gl_t P32 = 1ULL << 32;
constexpr gl_t INV_2_32 = 18446744065119617026ULL;  // 2^-32 (jumps.rs:15)
constexpr gl_t INV_2 = 9223372034707292161ULL;      // (p + 1) / 2

struct row {
    const gl_t* __restrict__ p;
    size_t cs;
    __device__ __forceinline__ gl_t operator()(int c) const { return p[(size_t)c * cs]; }
    __device__ __forceinline__ gl_t ch(int i, int f) const { return p[(size_t)(CH + 6 * i + f) * cs]; }
    // sum_{i < n} col[base + i] << (i + shift)
    __device__ __forceinline__ gl_t le(int base, int n, int shift = 0) const {
        gl_t acc = 0;
        for (int i = 0; i < n; i++) acc = gl_add(acc, gl_mul((*this)(base + i), (gl_t)1 << (i + shift)));
        return acc;
    }
};
__device__ __forceinline__ gl_t word(gl_t b0, gl_t b1, gl_t b2, gl_t b3) {
    return gl_add(gl_add(b0, gl_mul(b1, 1ULL << 8)), gl_add(gl_mul(b2, 1ULL << 16), gl_mul(b3, 1ULL << 24)));
}

// memio.rs enforce_half_word :65-77 / enforce_byte :106-129
template <int NA>
__device__ __forceinline__ void half_word(consumer_t<NA>& k, gl_t op, gl_t rs1, gl_t mem, gl_t v1, gl_t v0) {
    k.constraint(gl_mul(op, gl_add(gl_mul(gl_sub(rs1, 1), gl_sub(mem, v0)), gl_mul(rs1, gl_sub(mem, v1)))));
}
struct byte_sel_t { gl_t rs0, rs1, aux, w00, w10, w01; };
template <int NA>
__device__ __forceinline__ void byte_sel(consumer_t<NA>& k, const byte_sel_t& s, gl_t op, gl_t mem, gl_t v00, gl_t v10, gl_t v01, gl_t v11) {
    k.constraint(gl_mul(op, gl_sub(gl_mul(s.rs0, s.rs1), s.aux)));
    gl_t sum = gl_add(gl_add(gl_mul(gl_sub(mem, v00), s.w00), gl_mul(gl_sub(mem, v10), s.w10)),
                      gl_add(gl_mul(gl_sub(mem, v01), s.w01), gl_mul(gl_sub(mem, v11), s.aux)));
    k.constraint(gl_mul(sum, op));
}
}  // namespace cpu

template <int NA>
__device__ void eval_cpu_constraints(const gl_t* __restrict__ lvp, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    using namespace cpu;
    const row lv{lvp, cs}, nv{lvp + dnext, cs};
    // ---- bootstrap
    {
        gl_t boot = lv(IS_BOOT), d = gl_sub(nv(IS_BOOT), boot);
        k.first_row(gl_sub(boot, 1));
        k.last_row(boot);
        k.transition(gl_mul(d, gl_add(d, 1)));
#pragma unroll 1
        for (int i = 0; i < 9; i++) {
            gl_t f = gl_mul(boot, lv.ch(i, USED));
            k.constraint(gl_mul(f, lv.ch(i, CTX)));
            k.constraint(gl_mul(f, lv.ch(i, SEG)));  // Segment::Code = 0
        }
#pragma unroll 1
        for (int i = 0; i < 9; i++) k.transition(gl_mul(d, lv.ch(i, USED)));
    }
    // ---- decode
    {
        gl_t km = lv(KERNEL);
        k.constraint(gl_mul(km, gl_sub(km, 1)));
#pragma unroll 1
        for (int i = 0; i < 6; i++) {
            gl_t b = lv(OPC + i);
            k.constraint(gl_mul(b, gl_sub(b, 1)));
        }
        const int flags[15] = {EQ_ISZERO, KECCAK_GENERAL, JUMPS, BRANCH, PC_OP, GET_CONTEXT, SET_CONTEXT, EXIT_KERNEL,
                               LOGIC, BINARY, BINARY_IMM, SHIFT, SHIFT_IMM, M_OP_LOAD, M_OP_STORE};
        gl_t sum = 0;
#pragma unroll
        for (int i = 0; i < 15; i++) {
            gl_t f = lv(flags[i]);
            k.constraint(gl_mul(f, gl_sub(f, 1)));
            sum = gl_add(sum, f);
        }
        k.constraint(gl_mul(sum, gl_sub(sum, 1)));
    }
    // shared instruction fields
    const gl_t rs_f = lv.le(RS, 5), rt_f = lv.le(RT, 5), rd_f = lv.le(RD, 5), sa_f = lv.le(SHAMT, 5), fn_f = lv.le(FUNC, 6);
    const gl_t imm16 = gl_add(gl_add(fn_f, gl_mul(sa_f, 1ULL << 6)), gl_mul(rd_f, 1ULL << 11));  // insn[15:0]
    const gl_t sign16 = lv(RD + 4);                                                               // insn[15]
    const gl_t pc = lv(PROGRAM_COUNTER), npc_next = nv(NEXT_PC);
    // sign_extend(imm16 << 2): 18 low bits + 14 copies of the sign
    const gl_t off4 = gl_add(gl_mul(imm16, 4), gl_mul(sign16, 0xFFFC0000ULL));
    // ---- jumps
    {
        gl_t is_jump = lv(JUMPS), is_jumpi = lv(JUMPI), is_jd = lv(JUMPDIRECT);
        gl_t is_link = gl_mul(is_jump, lv(FUNC)), is_linki = gl_mul(is_jumpi, lv(OPC));
        k.constraint(gl_mul(is_jump, gl_sub(npc_next, lv.ch(0, VAL))));
        k.constraint(gl_mul(is_jump, gl_sub(rs_f, lv.ch(0, VIRT))));
        gl_t index26 = gl_add(gl_add(imm16, gl_mul(rt_f, 1ULL << 16)), gl_mul(rs_f, 1ULL << 21));
        gl_t aux = lv.ch(2, VAL);
        k.constraint(gl_mul(is_jumpi, gl_sub(npc_next, gl_add(aux, gl_mul(index26, 4)))));
        k.constraint(gl_mul(is_jd, gl_sub(aux, off4)));
        gl_t dst = gl_add(gl_add(pc, 4), aux);
        k.constraint(gl_mul(gl_mul(is_jd, gl_sub(npc_next, dst)), gl_sub(gl_add(npc_next, P32), dst)));
        k.constraint(gl_mul(gl_add(gl_add(is_link, is_linki), is_jd), gl_sub(gl_add(pc, 8), lv.ch(1, VAL))));
        gl_t link_reg = lv.ch(1, VIRT);
        k.constraint(gl_mul(is_link, gl_sub(link_reg, rd_f)));
        k.constraint(gl_mul(gl_add(is_linki, is_jd), gl_sub(link_reg, 31)));
    }
    // ---- branch
    {
        gl_t f = lv(BRANCH), sj = lv(BR + 0), fgt = lv(BR + 1), flt = lv(BR + 2), feq = lv(BR + 3);
        gl_t is_gt = lv(BR + 4), is_lt = lv(BR + 5), is_eq = lv(BR + 6), is_ge = lv(BR + 7), is_le = lv(BR + 8), is_ne = lv(BR + 9);
        gl_t norm = gl_add(gl_add(is_eq, is_ne), gl_add(is_le, is_gt)), special = gl_add(is_ge, is_lt);
        gl_t src1 = lv.ch(0, VAL), src2 = lv.ch(1, VAL), aux1 = lv.ch(2, VAL), aux2 = lv.ch(3, VAL), aux3 = lv.ch(4, VAL), aux4 = lv.ch(5, VAL);
        gl_t nf = gl_sub(1, f);
        k.constraint(gl_mul(sj, gl_sub(1, sj)));
        k.constraint(gl_mul(sj, nf));
        k.constraint(gl_mul(f, gl_sub(1, gl_add(norm, special))));
        k.constraint(gl_mul(f, gl_sub(1, gl_add(gl_add(flt, fgt), feq))));
        k.constraint(gl_mul(f, gl_sub(aux4, off4)));
        gl_t dst = gl_add(gl_add(pc, 4), aux4);
        k.constraint(gl_mul(gl_mul(sj, gl_sub(npc_next, dst)), gl_sub(gl_add(npc_next, P32), dst)));
        k.constraint(gl_mul(gl_mul(f, gl_sub(1, sj)), gl_sub(npc_next, gl_add(pc, 8))));
        gl_t ca = gl_sub(gl_add(aux1, src2), src1), cb = gl_sub(gl_add(aux2, src1), src2);
        gl_t fca = gl_mul(f, ca), fcb = gl_mul(f, cb);
        k.constraint(gl_mul(fca, gl_sub(ca, P32)));
        k.constraint(gl_mul(fcb, gl_sub(cb, P32)));
        k.constraint(gl_mul(gl_mul(f, aux1), gl_sub(gl_add(aux1, aux2), P32)));
        k.constraint(gl_mul(gl_mul(f, aux3), gl_sub(1, aux3)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rs_f)));
        gl_t rt_reg = lv.ch(1, VIRT);
        k.constraint(gl_mul(norm, gl_sub(rt_reg, rt_f)));
        k.constraint(gl_mul(gl_mul(special, rt_reg), gl_sub(1, rt_reg)));
        k.constraint(gl_mul(fca, gl_sub(P32, ca)));
        gl_t lt = gl_mul(ca, INV_2_32);
        k.constraint(gl_mul(flt, gl_sub(1, lt)));
        k.constraint(gl_mul(fcb, gl_sub(P32, cb)));
        gl_t gt = gl_mul(cb, INV_2_32);
        k.constraint(gl_mul(fgt, gl_sub(1, gt)));
        gl_t ne = gl_add(lt, gt);
        k.constraint(gl_mul(feq, ne));
        // x xor aux3 = x + aux3 - 2 x aux3
        gl_t lt2 = gl_sub(gl_add(flt, aux3), gl_mul(gl_add(flt, flt), aux3));
        gl_t gt2 = gl_sub(gl_add(fgt, aux3), gl_mul(gl_add(fgt, fgt), aux3));
        k.constraint(gl_mul(is_eq, nf));
        k.constraint(gl_mul(is_eq, gl_sub(sj, gl_sub(1, ne))));
        k.constraint(gl_mul(is_ne, nf));
        k.constraint(gl_mul(is_ne, gl_sub(sj, ne)));
        k.constraint(gl_mul(is_le, nf));
        k.constraint(gl_mul(is_le, gl_sub(sj, gl_sub(1, gt2))));
        k.constraint(gl_mul(is_ge, nf));
        k.constraint(gl_mul(is_ge, gl_sub(sj, gl_sub(1, lt2))));
        k.constraint(gl_mul(is_gt, nf));
        k.constraint(gl_mul(is_gt, gl_sub(sj, gt2)));
        k.constraint(gl_mul(is_lt, nf));
        k.constraint(gl_mul(is_lt, gl_sub(sj, lt2)));
    }
    // ---- membus
    k.constraint(gl_sub(lv(CODE_CONTEXT), gl_mul(gl_sub(1, lv(KERNEL)), lv(CONTEXT))));
#pragma unroll 1
    for (int i = 0; i < 9; i++) {
        gl_t u = lv.ch(i, USED);
        k.constraint(gl_mul(u, gl_sub(u, 1)));
    }
    // ---- memio: io view = rs_le (GEN..), rt_le (GEN + 32..), mem_le (GEN + 64..), aux_rs0_mul_rs1 (GEN + 96)
    {
        gl_t R[4], M[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            R[j] = lv.le(GEN + 32 + 8 * j, 8);
            M[j] = lv.le(GEN + 64 + 8 * j, 8);
        }
        const gl_t rs0 = lv(GEN), rs1 = lv(GEN + 1);
        const gl_t virt = lv.le(GEN + 2, 30, 2);
        const gl_t rs_from_bits = gl_add(virt, gl_add(rs0, gl_add(rs1, rs1)));
        const gl_t rt_word = word(R[0], R[1], R[2], R[3]), mem_word = word(M[0], M[1], M[2], M[3]);
        const gl_t m7 = lv(GEN + 64 + 7), m15 = lv(GEN + 64 + 15), m23 = lv(GEN + 64 + 23), m31 = lv(GEN + 64 + 31);
        const gl_t aux_filter = lv(MEMIO + 15), rs = lv.ch(0, VAL), rt = lv.ch(1, VAL), mem = lv.ch(3, VAL);
        const gl_t virt_raw = gl_add(rs, gl_add(imm16, gl_mul(sign16, 0xFFFF0000ULL)));
        byte_sel_t s;
        s.rs0 = rs0; s.rs1 = rs1; s.aux = lv(GEN + 96);
        s.w00 = gl_add(gl_sub(gl_sub(s.aux, rs1), rs0), 1); s.w10 = gl_sub(s.aux, rs0); s.w01 = gl_sub(s.aux, rs1);
        const gl_t addr_check = gl_mul(gl_mul(aux_filter, gl_sub(rs_from_bits, virt_raw)), gl_sub(gl_add(rs_from_bits, P32), virt_raw));
#pragma unroll 1
        for (int store = 0; store < 2; store++) {
            gl_t f = gl_mul(lv(store ? M_OP_STORE : M_OP_LOAD), lv(OPC + 5));
            k.constraint(gl_mul(f, gl_sub(1, aux_filter)));
            k.constraint(gl_mul(f, gl_sub(lv.ch(0, SEG), 4)));  // Segment::RegisterFile
            k.constraint(gl_mul(f, gl_sub(lv.ch(1, SEG), 4)));
            k.constraint(addr_check);
            k.constraint(gl_mul(f, gl_sub(rt_word, rt)));
            k.constraint(gl_mul(f, gl_sub(virt, lv.ch(2, VIRT))));
            if (!store) {
                const gl_t lo16 = gl_add(M[0], gl_mul(M[1], 256)), hi16 = gl_add(M[2], gl_mul(M[3], 256));
                half_word(k, lv(MEMIO + 0), rs1, mem, gl_add(lo16, gl_mul(m15, 0xFFFF0000ULL)), gl_add(hi16, gl_mul(m31, 0xFFFF0000ULL)));
                byte_sel(k, s, lv(MEMIO + 1), mem, mem_word, word(R[0], M[0], M[1], M[2]), word(R[0], R[1], M[0], M[1]),
                         word(R[0], R[1], R[2], M[0]));
                k.constraint(gl_mul(lv(MEMIO + 2), gl_sub(mem, mem_word)));
                byte_sel(k, s, lv(MEMIO + 3), mem, M[3], M[2], M[1], M[0]);
                half_word(k, lv(MEMIO + 4), rs1, mem, lo16, hi16);
                byte_sel(k, s, lv(MEMIO + 5), mem, word(M[3], R[1], R[2], R[3]), word(M[2], M[3], R[2], R[3]), word(M[1], M[2], M[3], R[3]),
                         mem_word);
                k.constraint(gl_mul(lv(MEMIO + 11), gl_sub(mem, mem_word)));
                byte_sel(k, s, lv(MEMIO + 14), mem, gl_add(M[3], gl_mul(m31, 0xFFFFFF00ULL)), gl_add(M[2], gl_mul(m23, 0xFFFFFF00ULL)),
                         gl_add(M[1], gl_mul(m15, 0xFFFFFF00ULL)), gl_add(M[0], gl_mul(m7, 0xFFFFFF00ULL)));
            } else {
                byte_sel(k, s, lv(MEMIO + 6), mem, word(M[0], M[1], M[2], R[0]), word(M[0], M[1], R[0], M[3]), word(M[0], R[0], M[2], M[3]),
                         word(R[0], M[1], M[2], M[3]));
                half_word(k, lv(MEMIO + 7), rs1, mem, word(R[0], R[1], M[2], M[3]), word(M[0], M[1], R[0], R[1]));
                byte_sel(k, s, lv(MEMIO + 8), mem, rt_word, word(R[1], R[2], R[3], M[3]), word(R[2], R[3], M[2], M[3]),
                         word(R[3], M[1], M[2], M[3]));
                k.constraint(gl_mul(lv(MEMIO + 9), gl_sub(mem, rt_word)));
                byte_sel(k, s, lv(MEMIO + 10), mem, word(M[0], M[1], M[2], R[0]), word(M[0], M[1], R[0], R[1]), word(M[0], R[0], R[1], R[2]),
                         rt_word);
                k.constraint(gl_mul(lv(MEMIO + 12), gl_sub(mem, rt_word)));
                k.constraint(gl_mul(lv(MEMIO + 13), mem));
            }
            k.constraint(gl_mul(f, lv.ch(6, USED)));
            k.constraint(gl_mul(f, lv.ch(7, USED)));
        }
    }
    // ---- shift: variable then immediate; the power of two comes from the shift table through channel 3
    {
        const gl_t used3 = lv.ch(3, USED), rd1 = gl_sub(lv.ch(3, IS_READ), 1), ctx3 = lv.ch(3, CTX), seg3 = gl_sub(lv.ch(3, SEG), 3),
                   virt3 = lv.ch(3, VIRT);
#pragma unroll 1
        for (int imm = 0; imm < 2; imm++) {
            gl_t f = lv(imm ? SHIFT_IMM : SHIFT), disp = imm ? sa_f : lv.ch(0, VAL);
            k.constraint(gl_mul(gl_mul(f, used3), rd1));
            k.constraint(gl_mul(f, ctx3));
            k.constraint(gl_mul(f, seg3));
            k.constraint(gl_mul(f, gl_sub(virt3, disp)));
        }
    }
    // ---- count (CLZ / CLO)
    {
        gl_t fz = lv(CLZ), fo = lv(CLO), f = gl_add(fo, fz);
        k.constraint(gl_mul(f, gl_sub(lv.le(OPC, 6), 0x1c)));
        k.constraint(gl_mul(fz, gl_sub(fn_f, 0x20)));
        k.constraint(gl_mul(fo, gl_sub(fn_f, 0x21)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rs_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(1, VIRT), rd_f)));
        gl_t sum = 0;
#pragma unroll 1
        for (int i = 0; i < 32; i++) {
            gl_t b = lv(GEN + i);
            k.constraint(gl_mul(gl_mul(f, b), gl_sub(1, b)));
            sum = gl_add(sum, gl_mul(b, (gl_t)1 << i));
        }
        gl_t rs = lv.ch(0, VAL), rd = lv.ch(1, VAL);
        k.constraint(gl_mul(fz, gl_sub(rs, sum)));
        k.constraint(gl_mul(fo, gl_sub(gl_sub(0xffffffffULL, rs), sum)));
        gl_t partial = lv(GEN + 31);  // bits[i..] as an integer, Horner from the top
        k.constraint(gl_mul(gl_mul(f, partial), rd));
#pragma unroll 1
        for (int i = 30; i >= 0; i--) {
            partial = gl_add(gl_add(partial, partial), lv(GEN + i));
            const int j = 30 - i;
            gl_t is_eq = lv(GEN + 32 + j), inv = lv(GEN + 64 + j), diff = gl_sub(partial, 1), feq = gl_mul(f, is_eq);
            k.constraint(gl_mul(feq, diff));
            k.constraint(gl_mul(f, gl_sub(gl_add(gl_mul(diff, inv), is_eq), 1)));
            k.constraint(gl_mul(feq, gl_sub(rd, (gl_t)(31 - i))));
        }
        gl_t is_eq = lv(GEN + 32 + 31), inv = lv(GEN + 64 + 31), feq = gl_mul(f, is_eq);
        k.constraint(gl_mul(feq, partial));
        k.constraint(gl_mul(f, gl_sub(gl_add(gl_mul(partial, inv), is_eq), 1)));
        k.constraint(gl_mul(feq, gl_sub(rd, 32)));
    }
    // ---- syscall: general = cond[12] sysnum[12] a0[3] a1
    {
        const gl_t f = lv(SYSCALL);
        auto cond = [&](int i) { return lv(GEN + i); };
        auto sysnum = [&](int i) { return lv(GEN + 12 + i); };
        const gl_t a0_is0 = lv(GEN + 24), a0_is12 = lv(GEN + 25), a0_else = lv(GEN + 26), sz_nz = lv(GEN + 27);
        const gl_t a0 = lv.ch(1, VAL), a1 = lv.ch(2, VAL), a2 = lv.ch(3, VAL), rv0 = lv.ch(4, VAL), rv1 = lv.ch(5, VAL);
        const gl_t c6 = lv.ch(6, VAL), rheap = lv.ch(7, VAL);
        const gl_t bad0 = gl_sub(0xFFFFFFFFULL, rv0), bad1 = gl_sub(9, rv1), zero1 = gl_sub(0, rv1);
        auto fc = [&](gl_t c, gl_t x) { k.constraint(gl_mul(gl_mul(f, c), x)); };
        auto def = [&](gl_t c, gl_t a, gl_t b) { k.constraint(gl_mul(f, gl_sub(c, gl_mul(a, b)))); };
        const gl_t is_map = sysnum(1);
        def(cond(0), is_map, a0_is0);
        def(cond(1), cond(0), sz_nz);
        fc(cond(1), gl_sub(gl_add(c6, sysnum(9)), rheap));
        def(cond(2), cond(0), sysnum(10));
        fc(cond(2), gl_sub(gl_add(c6, a1), rheap));
        fc(cond(0), gl_sub(c6, rv0));
        def(cond(3), is_map, a0_else);
        fc(cond(3), gl_sub(a0, rv0));
        const gl_t is_brk = sysnum(2);
        fc(is_brk, gl_sub(1, gl_add(cond(10), cond(11))));
        fc(cond(10), gl_sub(a0, rv0));
        fc(cond(11), gl_sub(c6, rv0));
        fc(is_brk, zero1);
        const gl_t is_clone = sysnum(3);
        fc(is_clone, gl_sub(1, rv0));
        fc(is_clone, zero1);
        const gl_t is_read = sysnum(5);
        def(cond(4), is_read, a0_else);
        fc(cond(4), bad0);
        fc(cond(4), bad1);
        def(cond(5), is_read, a0_is0);
        fc(cond(5), gl_sub(0, rv0));
        fc(cond(5), zero1);
        const gl_t is_write = sysnum(6);
        def(cond(6), is_write, a0_else);
        fc(cond(6), bad0);
        fc(cond(6), bad1);
        def(cond(7), is_write, a0_is12);
        fc(cond(7), gl_sub(a2, rv0));
        fc(cond(7), zero1);
        const gl_t is_fcntl = sysnum(7);
        def(cond(8), is_fcntl, a0_is0);
        fc(cond(8), gl_sub(0, rv0));
        fc(cond(8), zero1);
        def(cond(9), is_fcntl, a0_is12);
        fc(cond(9), gl_sub(1, rv0));
        fc(cond(9), zero1);
        const gl_t rest = gl_sub(gl_sub(is_fcntl, cond(8)), cond(9));
        def(rest, is_fcntl, a0_else);
        fc(rest, bad0);
        fc(rest, bad1);
        fc(sysnum(8), gl_sub(a0, c6));
    }
    // ---- bits (SEB / SEH / WSBH) on io().rt_le
    {
        gl_t seh = lv(SIGNEXT16), seb = lv(SIGNEXT8), wsbh = lv(SWAPHALF), f = gl_add(gl_add(seh, seb), wsbh);
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rt_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(1, VIRT), rd_f)));
        gl_t B[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (int i = 0; i < 32; i++) {
            gl_t b = lv(GEN + 32 + i);
            k.constraint(gl_mul(gl_mul(f, b), gl_sub(1, b)));
            B[i >> 3] = gl_add(B[i >> 3], gl_mul(b, (gl_t)1 << (i & 7)));
        }
        gl_t rd = lv.ch(1, VAL);
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VAL), word(B[0], B[1], B[2], B[3]))));
        k.constraint(gl_mul(seb, gl_sub(rd, gl_add(B[0], gl_mul(lv(GEN + 32 + 7), 0xFFFFFF00ULL)))));
        k.constraint(gl_mul(seh, gl_sub(rd, gl_add(gl_add(B[0], gl_mul(B[1], 256)), gl_mul(lv(GEN + 32 + 15), 0xFFFF0000ULL)))));
        k.constraint(gl_mul(wsbh, gl_sub(rd, word(B[1], B[0], B[3], B[2]))));
    }
    // ---- misc: rs_bits (GEN..), is_msb (GEN + 32..), is_lsb (GEN + 64..), auxm, auxl, auxs, rd_index, rd_index_eq_0, rd_index_eq_29
    const gl_t auxm = lv(GEN + 96), auxl = lv(GEN + 97), auxs = lv(GEN + 98);
    {  // rdhwr
        gl_t f = lv(RDHWR), rd_index = lv(GEN + 99), eq0 = lv(GEN + 100), eq29 = lv(GEN + 101), rt_val = lv.ch(0, VAL);
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rt_f)));
        k.constraint(gl_mul(f, gl_sub(rd_index, rd_f)));
        gl_t f0 = gl_mul(f, eq0), f29 = gl_mul(f, eq29);
        k.constraint(gl_mul(f0, rd_index));
        k.constraint(gl_mul(f0, gl_sub(rt_val, 1)));
        k.constraint(gl_mul(f29, gl_sub(rd_index, 29)));
        k.constraint(gl_mul(f29, gl_sub(rt_val, lv.ch(1, VAL))));
        k.constraint(gl_mul(gl_mul(f, gl_sub(gl_sub(1, eq29), eq0)), rt_val));
    }
    {  // condmov
        gl_t rs = lv.ch(0, VAL), rt = lv.ch(1, VAL), rd = lv.ch(2, VAL), out = lv.ch(3, VAL), mov = lv.ch(4, VAL);
        gl_t movn = lv(MOVN), movz = lv(MOVZ), f = gl_add(movn, movz), is_ne = gl_mul(lv(GEN), rt), no_mov = gl_sub(1, mov);
        k.constraint(gl_mul(movn, gl_sub(mov, is_ne)));
        k.constraint(gl_mul(movz, gl_sub(mov, gl_sub(1, is_ne))));
        k.constraint(gl_mul(gl_mul(f, mov), no_mov));
        k.constraint(gl_mul(f, gl_sub(out, gl_add(gl_mul(mov, rs), gl_mul(no_mov, rd)))));
    }
    {  // teq
        gl_t f = lv(TEQ);
        k.constraint(gl_mul(f, gl_sub(lv.ch(1, VIRT), rt_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rs_f)));
        k.constraint(gl_mul(f, gl_sub(1, gl_mul(gl_sub(lv.ch(0, VAL), lv.ch(1, VAL)), lv(GEN)))));
    }
    {  // ext
        gl_t f = lv(EXT);
        k.constraint(gl_mul(f, gl_sub(lv.ch(1, VIRT), rt_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rs_f)));
        gl_t msb = gl_add(sa_f, rd_f);
        k.constraint(gl_mul(f, gl_sub(gl_add(gl_mul(lv.ch(1, VAL), auxs), auxl), auxm)));
        gl_t prefix = 0;  // rs_bits[0..i)
#pragma unroll 1
        for (int i = 0; i < 32; i++) {
            gl_t lpartial = prefix;
            prefix = gl_add(prefix, gl_mul(lv(GEN + i), (gl_t)1 << i));
            gl_t fm = gl_mul(f, lv(GEN + 32 + i)), fl = gl_mul(f, lv(GEN + 64 + i));
            k.constraint(gl_mul(fm, gl_sub(msb, (gl_t)i)));
            k.constraint(gl_mul(fm, gl_sub(auxm, prefix)));
            k.constraint(gl_mul(fl, gl_sub(sa_f, (gl_t)i)));
            k.constraint(gl_mul(fl, gl_sub(auxl, lpartial)));
            k.constraint(gl_mul(fl, gl_sub(auxs, (gl_t)1 << i)));
        }
    }
    {  // ror: rotate right by i = (x >> i) + (x mod 2^i) << (32 - i)
        gl_t f = lv(ROR);
        k.constraint(gl_mul(f, gl_sub(lv.ch(1, VIRT), rd_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rt_f)));
        gl_t rd_val = lv.ch(1, VAL), hi = lv.le(GEN, 32), lo = 0;
#pragma unroll 1
        for (int i = 0; i < 32; i++) {
            gl_t fs = gl_mul(f, lv(GEN + 64 + i));
            k.constraint(gl_mul(fs, gl_sub(sa_f, (gl_t)i)));
            k.constraint(gl_mul(fs, gl_sub(rd_val, gl_add(hi, gl_mul(lo, (gl_t)1 << ((32 - i) & 63))))));
            gl_t b = lv(GEN + i);
            lo = gl_add(lo, gl_mul(b, (gl_t)1 << i));
            hi = gl_mul(gl_sub(hi, b), INV_2);
        }
    }
    {  // ins
        gl_t f = lv(INS);
        k.constraint(gl_mul(f, gl_sub(lv.ch(1, VIRT), rt_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(2, VIRT), rt_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rs_f)));
        k.constraint(gl_mul(f, gl_sub(gl_sub(lv.ch(2, VAL), auxm), gl_mul(auxl, auxs))));
        gl_t size = gl_sub(rd_f, sa_f), prefix = 0;
#pragma unroll 1
        for (int i = 0; i < 32; i++) {
            prefix = gl_add(prefix, gl_mul(lv(GEN + i), (gl_t)1 << i));
            gl_t fm = gl_mul(f, lv(GEN + 32 + i)), fl = gl_mul(f, lv(GEN + 64 + i));
            k.constraint(gl_mul(fl, gl_sub(sa_f, (gl_t)i)));
            k.constraint(gl_mul(fl, gl_sub(auxs, (gl_t)1 << i)));
            k.constraint(gl_mul(fm, gl_sub(size, (gl_t)i)));
            k.constraint(gl_mul(fm, gl_sub(auxl, prefix)));
        }
    }
    {  // maddu
        gl_t f = lv(MADDU);
        k.constraint(gl_mul(f, gl_sub(lv.ch(0, VIRT), rs_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(1, VIRT), rt_f)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(2, VIRT), 33)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(4, VIRT), 33)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(3, VIRT), 32)));
        k.constraint(gl_mul(f, gl_sub(lv.ch(5, VIRT), 32)));
        gl_t result = gl_add(gl_mul(lv.ch(4, VAL), P32), lv.ch(5, VAL)), addend = gl_add(gl_mul(lv.ch(2, VAL), P32), lv.ch(3, VAL));
        gl_t mul = gl_mul(lv.ch(0, VAL), lv.ch(1, VAL));
        k.constraint(gl_mul(gl_mul(f, auxm), gl_sub(auxm, P32)));
        k.constraint(gl_mul(f, gl_sub(gl_sub(gl_add(mul, addend), gl_mul(auxm, P32)), result)));
    }
}

template <int TABLE, int NA>
__device__ __forceinline__ void eval_table_constraints(const gl_t* __restrict__ lv, size_t cs, ptrdiff_t dnext, consumer_t<NA>& k) {
    if constexpr (TABLE == ZKM_TABLE_POSEIDON) eval_poseidon_constraints<NA>(lv, cs, k);
    else if constexpr (TABLE == ZKM_TABLE_LOGIC) eval_logic_constraints<NA>(lv, cs, k);
    else if constexpr (TABLE == ZKM_TABLE_KECCAK_SPONGE) eval_keccak_sponge_constraints<NA>(lv, cs, dnext, k);
    else if constexpr (TABLE == ZKM_TABLE_KECCAK) eval_keccak_constraints<NA>(lv, cs, dnext, k);
    else if constexpr (TABLE == ZKM_TABLE_MEMORY) eval_memory_constraints<NA>(lv, cs, dnext, k);
    else if constexpr (TABLE == ZKM_TABLE_POSEIDON_SPONGE) eval_poseidon_sponge_constraints<NA>(lv, cs, dnext, k);
    else if constexpr (TABLE == ZKM_TABLE_SHA_EXTEND) eval_sha_extend_constraints<NA>(lv, cs, k);
    else if constexpr (TABLE == ZKM_TABLE_SHA_EXTEND_SPONGE) eval_sha_extend_sponge_constraints<NA>(lv, cs, dnext, k);
    else if constexpr (TABLE == ZKM_TABLE_SHA_COMPRESS) eval_sha_compress_constraints<NA>(lv, cs, dnext, k);
    else if constexpr (TABLE == ZKM_TABLE_SHA_COMPRESS_SPONGE) eval_sha_compress_sponge_constraints<NA>(lv, cs, k);
    else if constexpr (TABLE == ZKM_TABLE_ARITHMETIC) eval_arithmetic_constraints<NA>(lv, cs, dnext, k);
    else eval_cpu_constraints<NA>(lv, cs, dnext, k);
}

