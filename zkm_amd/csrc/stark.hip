// stark.hip -- prove_single_table on the GPU: quotient (K7-K9), openings (K10), FRI (K11-K14), queries.
//
// Host orchestration restates /root/reference/prover/src/prover.rs:441-641 (stage and transcript order);
// the kernels replace the rayon loops of compute_quotient_polys (:700-782), StarkOpeningSet::new
// (proof.rs:299-334) and plonky2's PolynomialBatch::prove_openings / fri_proof.
// The Challenger stays on the host (a few hundred field elements per table); only caps, openings,
// the final polynomial and the query openings cross PCIe.
#include "poseidon_dev.h"
#include "ctl_dev.h"
#include "zkm_internal.h"

// ------------------------------------------------------------------ proof layout (include/zkm_hip.h)
struct proof_layout {
    unsigned log_n, lde_bits, L, cap;
    size_t W, A, Q, Z, F, C, nq;
    size_t o_init, o_caps, o_open, o_fri_caps, o_final, o_pow, o_queries, query_words, total;
};

// FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) (config.rs:25)
static unsigned fri_num_layers(const zkm_stark_config* c, unsigned degree_bits) {
    unsigned l = 0, d = degree_bits;
    while (d > c->final_poly_bits && d + c->rate_bits - c->arity_bits >= c->cap_height) { d -= c->arity_bits; l++; }
    return l;
}

static void make_layout(proof_layout& y, const zkm_stark_config* c, unsigned log_n, size_t W, size_t A, size_t Z) {
    y.log_n = log_n; y.lde_bits = log_n + c->rate_bits; y.cap = c->cap_height;
    y.W = W; y.A = A; y.Q = (size_t)c->num_challenges * 2; y.Z = Z;
    y.L = fri_num_layers(c, log_n);
    y.F = (size_t)1 << (log_n - y.L * c->arity_bits);
    y.C = (size_t)1 << c->cap_height;
    y.nq = c->num_queries;
    size_t o = 16;
    y.o_init = o; o += 12;
    y.o_caps = o; o += 3 * y.C * 4;
    y.o_open = o; o += 4 * W + 4 * A + Z + 2 * y.Q;
    y.o_fri_caps = o; o += y.L * y.C * 4;
    y.o_final = o; o += 2 * y.F;
    y.o_pow = o; o += 1;
    y.o_queries = o;
    size_t sib0 = (size_t)(y.lde_bits - y.cap) * 4;
    size_t q = (W + sib0) + (A + sib0) + (y.Q + sib0);
    for (unsigned i = 0; i < y.L; i++)
        q += 2 * ((size_t)1 << c->arity_bits) + (size_t)(y.lde_bits - c->arity_bits * (i + 1) - y.cap) * 4;
    y.query_words = q;
    y.total = o + q * y.nq;
}

// ------------------------------------------------------------------ K7: quotient evaluation, Poseidon table
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
        constexpr uint64_t RC[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
            0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
            0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
            0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
            0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
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
__device__ const uint32_t SHA256_K_DEV[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
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
    gl_t challenges[4];
    struct { uint32_t ncols, col_off, table_col, freq_col; } lk[2];
    uint32_t cols[24];
};
template <int NA>
__device__ void eval_lookup_constraints(const lookup_dev& d, const gl_t* __restrict__ lv, size_t N, const gl_t* __restrict__ aux,
                                        size_t j, size_t jn, consumer_t<NA>& k) {
    uint32_t start = 0;
    for (uint32_t l = 0; l < d.nlookups; l++) {
        const uint32_t ncols = d.lk[l].ncols, nh = (ncols + 1) / 2;
        const uint32_t* cols = d.cols + d.lk[l].col_off;
        for (uint32_t c = 0; c < d.nch; c++) {
            const gl_t ch = d.challenges[c];
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
    else eval_sha_compress_sponge_constraints<NA>(lv, cs, k);
}

// CTL checks driven by the column-set description (eval_helper_columns cross_table_lookup.rs:1006-1058,
// eval_cross_table_lookup_checks :1067-1150).  The benchmark's fake CTL data (helper columns, no column sets)
// is the ncolsets == 0 case: only the last-row / transition checks on Z are emitted.
template <int NA>
__device__ void eval_ctl_constraints(const ctl_dev& d, const gl_t* __restrict__ tl, size_t N, ptrdiff_t dnext,
                                     const gl_t* __restrict__ aux, size_t j, size_t jn, consumer_t<NA>& k) {
    uint32_t start = 0;
    for (uint32_t i = 0; i < d.nzs; i++) {
        const zkm_ctl_z z = d.zs[i];
        const uint32_t* ids = d.colset_ids + z.colset_off;
        gl_t local_z = aux[(size_t)(d.total_helpers + i) * N + j], next_z = aux[(size_t)(d.total_helpers + i) * N + jn];
        if (z.num_helpers) {
            for (uint32_t q = 0; 2 * q < z.ncolsets; q++) {
                gl_t h = aux[(size_t)(start + q) * N + j];
                const zkm_colset c0 = d.colsets[ids[2 * q]];
                gl_t combin0 = ctl_combine(d, c0, z.beta, z.gamma, tl, N, dnext, true), f0 = ctl_eval_filter(d, c0, tl, N, dnext, true);
                if (2 * q + 1 < z.ncolsets) {
                    const zkm_colset c1 = d.colsets[ids[2 * q + 1]];
                    gl_t combin1 = ctl_combine(d, c1, z.beta, z.gamma, tl, N, dnext, true), f1 = ctl_eval_filter(d, c1, tl, N, dnext, true);
                    k.constraint(gl_sub(gl_sub(gl_mul(gl_mul(combin1, combin0), h), gl_mul(f0, combin1)), gl_mul(f1, combin0)));
                } else {
                    k.constraint(gl_sub(gl_mul(combin0, h), f0));
                }
            }
            gl_t h_sum = 0;
            for (uint32_t q = 0; q < z.num_helpers; q++) h_sum = gl_add(h_sum, aux[(size_t)(start + q) * N + j]);
            k.last_row(gl_sub(local_z, h_sum));
            k.transition(gl_sub(gl_sub(local_z, next_z), h_sum));
        } else if (z.ncolsets > 1) {
            const zkm_colset c0 = d.colsets[ids[0]], c1 = d.colsets[ids[1]];
            gl_t combin0 = ctl_combine(d, c0, z.beta, z.gamma, tl, N, dnext, true), combin1 = ctl_combine(d, c1, z.beta, z.gamma, tl, N, dnext, true);
            gl_t f0 = ctl_eval_filter(d, c0, tl, N, dnext, true), f1 = ctl_eval_filter(d, c1, tl, N, dnext, true);
            gl_t cc = gl_mul(combin0, combin1), rhs = gl_add(gl_mul(f0, combin1), gl_mul(f1, combin0));
            k.last_row(gl_sub(gl_mul(cc, local_z), rhs));
            k.transition(gl_sub(gl_mul(cc, gl_sub(local_z, next_z)), rhs));
        } else {
            const zkm_colset c0 = d.colsets[ids[0]];
            gl_t combin0 = ctl_combine(d, c0, z.beta, z.gamma, tl, N, dnext, true), f0 = ctl_eval_filter(d, c0, tl, N, dnext, true);
            k.last_row(gl_sub(gl_mul(combin0, local_z), f0));
            k.transition(gl_sub(gl_mul(combin0, gl_sub(local_z, next_z)), f0));
        }
        start += z.num_helpers;
    }
}

// One thread per point of the quotient domain g<w_2n>, visited in LDE storage order: storage row j < 2n
// of the 4n-row LDE is natural quotient index i = bitrev_{L-1}(j) (every `step` = 2nd natural LDE row,
// prover.rs:668-675); "next" is natural +2 in the 2n domain (prover.rs:704) = +4 in the 4n domain.
template <int TABLE, int NA>
__global__ __launch_bounds__(256) void k_quotient(const gl_t* __restrict__ trace, const gl_t* __restrict__ aux,
                                                           unsigned log_n, unsigned lde_bits, ctl_dev ctl, lookup_dev lookups,
                                                           uint32_t num_lookup_cols, const gl_t* alphas,
                                                           const gl_t* __restrict__ wpow /* w_{4n}^t two-level table */,
                                                           gl_t gn, gl_t zh_inv0, gl_t zh_inv1, gl_t last, gl_t w_n, gl_t n_inv,
                                                           gl_t* __restrict__ out) {
    size_t N = (size_t)1 << lde_bits;
    size_t size = N >> 1;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= size) return;
    uint32_t t = bitrev32((uint32_t)j, lde_bits);  // natural index in the 4n domain (even)
    uint32_t i = t >> 1;                           // natural index in the 2n quotient domain
    uint32_t tn = (t + 4) & (uint32_t)(N - 1);
    size_t jn = bitrev32(tn, lde_bits);

    // x = g * w_{4n}^t
    unsigned h = (lde_bits + 1) / 2;
    gl_t x = gl_mul(GL_GENERATOR, gl_mul_loose(wpow[t & ((1u << h) - 1)], wpow[((size_t)1 << h) + (t >> h)]));
    consumer_t<NA> k;
#pragma unroll
    for (int a = 0; a < NA; a++) { k.alpha[a] = alphas[a]; k.acc[a] = 0; }
    k.z_last = gl_sub(x, last);
    // Z_H(x) = x^n - 1 = g^n (-1)^i - 1;  L_first = Z_H / (n (x - 1)),  L_last = Z_H / (n (w x - 1))
    gl_t zh = gl_sub((i & 1) ? gl_neg(gn) : gn, 1);
    gl_t d0 = gl_sub(x, 1), d1 = gl_sub(gl_mul(w_n, x), 1);
    gl_t dinv = gl_inv(gl_mul(d0, d1));
    gl_t zn = gl_mul(zh, n_inv);
    k.l_first = gl_mul(zn, gl_mul(dinv, d1));
    k.l_last = gl_mul(zn, gl_mul(dinv, d0));

    eval_table_constraints<TABLE, NA>(trace + j, N, (ptrdiff_t)jn - (ptrdiff_t)j, k);
    // auxiliary columns: the table's lookup helper columns first, then the CTL helper columns and Zs (prover.rs:495-508)
    if constexpr (TABLE == ZKM_TABLE_MEMORY) eval_lookup_constraints<NA>(lookups, trace + j, N, aux, j, jn, k);
    eval_ctl_constraints<NA>(ctl, trace + j, N, (ptrdiff_t)jn - (ptrdiff_t)j, aux + (size_t)num_lookup_cols * N, j, jn, k);
    gl_t zi = (i & 1) ? zh_inv1 : zh_inv0;
#pragma unroll
    for (int a = 0; a < NA; a++) out[(size_t)a * size + i] = gl_mul(k.acc[a], zi);
}

// quotient polys: d_out = nalphas x 2n natural-order coefficients (device)
static void quotient_device(zkm_ctx* c, int table_id, const zkm_batch* trace, const zkm_batch* aux, const ctl_dev_owner& own,
                            const uint64_t* lookup_challenges, const gl_t* alphas_host, size_t nalphas, gl_t* d_out) {
    lookup_dev lookups{};
    uint32_t NL = 0;
    {
        size_t nl = 0;
        const zkm_table_lookup* defs = zkm_table_lookups(table_id, &nl);
        if (nl && !lookup_challenges) throw std::runtime_error("zkm_quotient: this table has lookups; lookup challenges are required");
        if (nl > 2) throw std::runtime_error("zkm_quotient: too many lookups");
        lookups.nlookups = (uint32_t)nl;
        lookups.nch = (uint32_t)nalphas;
        uint32_t off = 0;
        for (size_t l = 0; l < nl; l++) {
            if (off + defs[l].ncols > 24) throw std::runtime_error("zkm_quotient: too many lookup columns");
            lookups.lk[l] = {defs[l].ncols, off, defs[l].table_col, defs[l].freq_col};
            for (uint32_t i = 0; i < defs[l].ncols; i++) lookups.cols[off + i] = defs[l].cols[i];
            off += defs[l].ncols;
            NL += ((defs[l].ncols + 1) / 2 + 1) * (uint32_t)nalphas;
        }
        for (size_t i = 0; nl && i < nalphas; i++) lookups.challenges[i] = lookup_challenges[i];
    }
    if (zkm_table_width(table_id) == 0 || trace->ncols != zkm_table_width(table_id))
        throw std::runtime_error("zkm_quotient: unknown table id, or the trace width does not match the table");
    if (trace->rate_bits != 2 || aux->rate_bits != 2 || trace->log_n != aux->log_n)
        throw std::runtime_error("zkm_quotient: rate_bits must be 2 and the batches must have equal degree");
    if (nalphas < 1 || nalphas > 2) throw std::runtime_error("zkm_quotient: 1 or 2 challenges supported");
    if (own.naux + NL != aux->ncols) throw std::runtime_error("zkm_quotient: aux column count does not match the CTL description");
    const ctl_dev& ctl = own.d;
    unsigned log_n = trace->log_n, lde_bits = log_n + 2, log_q = log_n + 1;
    size_t size = (size_t)1 << log_q;
    gl_t w4 = gl_root_of_unity(lde_bits);
    const gl_t* wpow = c->pow_table(w4, lde_bits);
    gl_t gn = gl_exp_pow2(GL_GENERATOR, log_n);
    gl_t zh0 = gl_inv(gl_sub(gn, 1)), zh1 = gl_inv(gl_sub(gl_neg(gn), 1));
    gl_t w_n = gl_root_of_unity(log_n), last = gl_inv(w_n);
    gl_t n_inv = gl_inv((gl_t)(((uint64_t)1 << log_n) % GL_P));
    gl_t* d_alphas = (gl_t*)c->alloc(4 * sizeof(gl_t));
    ZKM_HIP_CHECK(hipMemcpyAsync(d_alphas, alphas_host, nalphas * sizeof(gl_t), hipMemcpyHostToDevice, c->stream));
    gl_t* d_vals = (gl_t*)c->alloc(nalphas * size * sizeof(gl_t));
    {
        static const char* const names[] = {"quotient_poseidon", "quotient_logic", "quotient_keccak_sponge", "quotient_keccak", "quotient_memory", "quotient_poseidon_sponge", "quotient_sha_extend", "quotient_sha_extend_sponge", "quotient_sha_compress",
                                            "quotient_sha_compress_sponge"};
        zkm_prof_scope ps(c, names[table_id]);
        dim3 grid((size + 255) / 256), block(256);
#define ZKM_LAUNCH_QUOTIENT(T, NA)                                                                                              \
    hipLaunchKernelGGL((k_quotient<T, NA>), grid, block, 0, c->stream, trace->lde, aux->lde, log_n, lde_bits, ctl, lookups, NL, d_alphas, \
                       wpow, gn, zh0, zh1, last, w_n, n_inv, d_vals)
        switch (table_id * 2 + (int)nalphas - 1) {
            case 0: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON, 1); break;
            case 1: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON, 2); break;
            case 2: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_LOGIC, 1); break;
            case 3: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_LOGIC, 2); break;
            case 4: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK_SPONGE, 1); break;
            case 5: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK_SPONGE, 2); break;
            case 6: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK, 1); break;
            case 7: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_KECCAK, 2); break;
            case 8: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_MEMORY, 1); break;
            case 9: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_MEMORY, 2); break;
            case 10: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON_SPONGE, 1); break;
            case 11: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_POSEIDON_SPONGE, 2); break;
            case 12: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND, 1); break;
            case 13: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND, 2); break;
            case 14: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND_SPONGE, 1); break;
            case 15: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_EXTEND_SPONGE, 2); break;
            case 16: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS, 1); break;
            case 17: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS, 2); break;
            case 18: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS_SPONGE, 1); break;
            default: ZKM_LAUNCH_QUOTIENT(ZKM_TABLE_SHA_COMPRESS_SPONGE, 2); break;
        }
#undef ZKM_LAUNCH_QUOTIENT
        ZKM_HIP_CHECK(hipGetLastError());
    }
    // coset_ifft(g) of each challenge's evaluations (prover.rs:784-788)
    zkm_ntt_natural(c, d_vals, d_out, nalphas, size, size, log_q, true, GL_GENERATOR);
    c->sync();  // d_alphas / d_vals are recycled below
    c->release(d_vals);
    c->release(d_alphas);
}

// ------------------------------------------------------------------ K10: openings
// partial[col][chunk] = sum_{k in chunk} c_k z^(k - chunk_start) for z in {zeta, g*zeta}, plus the plain
// sum (evaluation at 1).  Horner in z^256 per thread over a 256-strided slice, so loads are coalesced.
#define OPEN_CHUNK_LOG 14
__global__ __launch_bounds__(256) void k_open_partials(const gl_t* __restrict__ coeffs, unsigned log_n, gl2_t z0, gl2_t z1,
                                                       gl_t* __restrict__ partial /* [col][chunk][5] */) {
    __shared__ gl_t red[256 * 5];
    unsigned chunk_log = log_n < OPEN_CHUNK_LOG ? log_n : OPEN_CHUNK_LOG;
    size_t chunk_len = (size_t)1 << chunk_log, nchunks = (size_t)1 << (log_n - chunk_log);
    size_t col = blockIdx.y, chunk = blockIdx.x;
    const gl_t* p = coeffs + (col << log_n) + chunk * chunk_len;
    unsigned t = threadIdx.x;
    gl2_t w0 = gl2_pow(z0, 256), w1 = gl2_pow(z1, 256);
    gl2_t a0{0, 0}, a1{0, 0};
    gl_t sum = 0;
    // elements t, t+256, ... : Horner from the top
    for (size_t m = chunk_len >> 8; m-- > 0;) {
        size_t idx = (m << 8) + t;
        gl_t cv = idx < chunk_len ? p[idx] : 0;
        a0 = gl2_mul(a0, w0); a0.c0 = gl_add(a0.c0, cv);
        a1 = gl2_mul(a1, w1); a1.c0 = gl_add(a1.c0, cv);
        sum = gl_add(sum, cv);
    }
    if (chunk_len < 256) {  // tiny polynomials: one element per thread at most
        gl_t cv = t < chunk_len ? p[t] : 0;
        a0 = gl2_t{cv, 0}; a1 = gl2_t{cv, 0}; sum = cv;
    }
    a0 = gl2_mul(a0, gl2_pow(z0, t));
    a1 = gl2_mul(a1, gl2_pow(z1, t));
    red[t * 5 + 0] = a0.c0; red[t * 5 + 1] = a0.c1; red[t * 5 + 2] = a1.c0; red[t * 5 + 3] = a1.c1; red[t * 5 + 4] = sum;
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (t < s)
            for (int q = 0; q < 5; q++) red[t * 5 + q] = gl_add(red[t * 5 + q], red[(t + s) * 5 + q]);
        __syncthreads();
    }
    if (t < 5) partial[(col * nchunks + chunk) * 5 + t] = red[t];
}

struct open_vals { gl2_t at_z0, at_z1; gl_t at_one; };

// evaluate every polynomial of the batch at z0, z1 (F2) and 1
static std::vector<open_vals> eval_batch(zkm_ctx* c, const zkm_batch* b, gl2_t z0, gl2_t z1) {
    unsigned log_n = b->log_n;
    unsigned chunk_log = log_n < OPEN_CHUNK_LOG ? log_n : OPEN_CHUNK_LOG;
    size_t nchunks = (size_t)1 << (log_n - chunk_log), words = b->ncols * nchunks * 5;
    gl_t* d_part = (gl_t*)c->alloc(words * sizeof(gl_t));
    {
        zkm_prof_scope ps(c, "open_partials");
        hipLaunchKernelGGL(k_open_partials, dim3(nchunks, b->ncols), dim3(256), 0, c->stream, b->coeffs, log_n, z0, z1, d_part);
        ZKM_HIP_CHECK(hipGetLastError());
    }
    std::vector<gl_t> part(words);
    ZKM_HIP_CHECK(hipMemcpyAsync(part.data(), d_part, words * sizeof(gl_t), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->release(d_part);
    gl2_t s0 = gl2_pow(z0, (uint64_t)1 << chunk_log), s1 = gl2_pow(z1, (uint64_t)1 << chunk_log);
    std::vector<open_vals> out(b->ncols);
    for (size_t col = 0; col < b->ncols; col++) {
        gl2_t a0{0, 0}, a1{0, 0};
        gl_t sum = 0;
        for (size_t ch = nchunks; ch-- > 0;) {
            const gl_t* q = &part[(col * nchunks + ch) * 5];
            a0 = gl2_add(gl2_mul(a0, s0), gl2_t{q[0], q[1]});
            a1 = gl2_add(gl2_mul(a1, s1), gl2_t{q[2], q[3]});
            sum = gl_add(sum, q[4]);
        }
        out[col] = open_vals{a0, a1, sum};
    }
    return out;
}

// ------------------------------------------------------------------ K11: FRI batch combination
// comp1 = sum_{j < W+A} alpha^j p_j (g*zeta batch), comp0 = comp1 + sum_q alpha^(W+A+q) quot_q (zeta batch),
// comp2 = sum_k alpha^k ctl_z_k (batch at 1)   -- polynomial order of stark.rs:127-148.  One pass over all
// coefficient polynomials; alpha powers are wave-uniform.
__global__ __launch_bounds__(256) void k_fri_combine(const gl_t* __restrict__ tc, size_t W, const gl_t* __restrict__ ac, size_t A,
                                                     const gl_t* __restrict__ qc, size_t Q, size_t ctl_start,
                                                     const gl_t* __restrict__ apow /* [(W+A+Q)][2] */, size_t n,
                                                     gl_t* __restrict__ comp /* [3][2][n] */) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl_t a0 = 0, a1 = 0, z0 = 0, z1 = 0;
    size_t j = 0;
    for (size_t cidx = 0; cidx < W; cidx++, j++) {
        gl_t v = tc[cidx * n + i];
        a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
        a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
    }
    for (size_t cidx = 0; cidx < A; cidx++, j++) {
        gl_t v = ac[cidx * n + i];
        a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
        a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
        if (cidx >= ctl_start) {
            size_t k = cidx - ctl_start;
            z0 = gl_add(z0, gl_mul(v, apow[2 * k]));
            z1 = gl_add(z1, gl_mul(v, apow[2 * k + 1]));
        }
    }
    comp[2 * n + i] = a0;
    comp[3 * n + i] = a1;
    for (size_t cidx = 0; cidx < Q; cidx++, j++) {
        gl_t v = qc[cidx * n + i];
        a0 = gl_add(a0, gl_mul(v, apow[2 * j]));
        a1 = gl_add(a1, gl_mul(v, apow[2 * j + 1]));
    }
    comp[i] = a0;
    comp[n + i] = a1;
    comp[4 * n + i] = z0;
    comp[5 * n + i] = z1;
}

// divide_by_linear as a hierarchical suffix scan with segments of 64:
// totals:  out[s] = sum_{k<64} a[64 s + k] z^k
__global__ void k_seg_totals(const gl_t* __restrict__ a0, const gl_t* __restrict__ a1, size_t m, gl2_t z, gl_t* __restrict__ o0,
                             gl_t* __restrict__ o1) {
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nseg = (m + 63) / 64;
    if (s >= nseg) return;
    gl2_t acc{0, 0};
    size_t end = (s + 1) * 64 < m ? (s + 1) * 64 : m;
    for (size_t k = end; k-- > s * 64;) acc = gl2_add(gl2_mul(acc, z), gl2_t{a0[k], a1[k]});
    o0[s] = acc.c0;
    o1[s] = acc.c1;
}
// scan:  S[k] = a[k] + z S[k+1] inside each segment, carry-in = upper[s+1] (0 past the end)
// mode 0: store S;  mode 1 (bottom level): fin[k-1] = fin[k-1] * shift + S[k]  (the dropped remainder is S[0]),
//         fin[m-1] = fin[m-1] * shift
__global__ void k_seg_scan(const gl_t* __restrict__ a0, const gl_t* __restrict__ a1, size_t m, gl2_t z, const gl_t* __restrict__ u0,
                           const gl_t* __restrict__ u1, size_t nupper, gl_t* __restrict__ s0, gl_t* __restrict__ s1, int mode,
                           gl2_t shift) {
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nseg = (m + 63) / 64;
    if (s >= nseg) return;
    gl2_t acc{0, 0};
    if (u0 && s + 1 < nupper) acc = gl2_t{u0[s + 1], u1[s + 1]};
    size_t end = (s + 1) * 64 < m ? (s + 1) * 64 : m;
    if (mode == 1 && end == m) {
        gl2_t f = gl2_mul(gl2_t{s0[m - 1], s1[m - 1]}, shift);
        s0[m - 1] = f.c0;
        s1[m - 1] = f.c1;
    }
    for (size_t k = end; k-- > s * 64;) {
        acc = gl2_add(gl2_mul(acc, z), gl2_t{a0[k], a1[k]});
        if (mode == 0) {
            s0[k] = acc.c0;
            s1[k] = acc.c1;
        } else if (k > 0) {
            gl2_t f = gl2_add(gl2_mul(gl2_t{s0[k - 1], s1[k - 1]}, shift), acc);
            s0[k - 1] = f.c0;
            s1[k - 1] = f.c1;
        }
    }
}

// fin = fin * shift + (comp(X) - comp(z)) / (X - z), re-padded to n coefficients
static void divide_accumulate(zkm_ctx* c, const gl_t* a0, const gl_t* a1, size_t n, gl2_t z, gl2_t shift, gl_t* f0, gl_t* f1) {
    // build the pyramid of totals
    struct level { gl_t *t0, *t1; size_t m; gl2_t z; };
    std::vector<level> lv;
    lv.push_back(level{const_cast<gl_t*>(a0), const_cast<gl_t*>(a1), n, z});
    zkm_prof_scope ps(c, "fri_divide_linear");
    while (lv.back().m > 64) {
        level& b = lv.back();
        size_t nseg = (b.m + 63) / 64;
        gl_t* t0 = (gl_t*)c->alloc(nseg * sizeof(gl_t));
        gl_t* t1 = (gl_t*)c->alloc(nseg * sizeof(gl_t));
        hipLaunchKernelGGL(k_seg_totals, dim3((nseg + 63) / 64), dim3(64), 0, c->stream, b.t0, b.t1, b.m, b.z, t0, t1);
        lv.push_back(level{t0, t1, nseg, gl2_pow(b.z, 64)});
    }
    // top-down: suffix values of each level
    std::vector<std::pair<gl_t*, gl_t*>> S(lv.size(), {nullptr, nullptr});
    for (size_t l = lv.size(); l-- > 0;) {
        level& b = lv[l];
        size_t nseg = (b.m + 63) / 64;
        const gl_t *u0 = nullptr, *u1 = nullptr;
        size_t nupper = 0;
        if (l + 1 < lv.size()) { u0 = S[l + 1].first; u1 = S[l + 1].second; nupper = lv[l + 1].m; }
        if (l == 0) {
            hipLaunchKernelGGL(k_seg_scan, dim3((nseg + 63) / 64), dim3(64), 0, c->stream, b.t0, b.t1, b.m, b.z, u0, u1, nupper, f0, f1, 1, shift);
        } else {
            S[l].first = (gl_t*)c->alloc(b.m * sizeof(gl_t));
            S[l].second = (gl_t*)c->alloc(b.m * sizeof(gl_t));
            hipLaunchKernelGGL(k_seg_scan, dim3((nseg + 63) / 64), dim3(64), 0, c->stream, b.t0, b.t1, b.m, b.z, u0, u1, nupper, S[l].first,
                               S[l].second, 0, shift);
        }
    }
    ZKM_HIP_CHECK(hipGetLastError());
    c->sync();
    for (size_t l = 1; l < lv.size(); l++) {
        c->release(lv[l].t0); c->release(lv[l].t1);
        c->release(S[l].first); c->release(S[l].second);
    }
}

// ------------------------------------------------------------------ K13: FRI fold
// c'_j = sum_{i < arity} beta^i c_{arity j + i}   (reduce_with_powers per chunk)
__global__ __launch_bounds__(256) void k_fri_fold(const gl_t* __restrict__ c0, const gl_t* __restrict__ c1, size_t nout, unsigned arity,
                                                  gl2_t beta, gl_t* __restrict__ o0, gl_t* __restrict__ o1) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nout) return;
    gl2_t acc{0, 0};
    for (unsigned i = arity; i-- > 0;) acc = gl2_add(gl2_mul(acc, beta), gl2_t{c0[j * arity + i], c1[j * arity + i]});
    o0[j] = acc.c0;
    o1[j] = acc.c1;
}

// ------------------------------------------------------------------ K14: proof of work
// Smallest candidate w such that Poseidon(state with w written at `pos`)[7] has >= pow_bits leading zeros.
struct pow_state { uint64_t s[12]; };
__global__ __launch_bounds__(256) void k_pow_search(pow_state st, unsigned pos, unsigned pow_bits, uint64_t base, unsigned long long* best) {
    uint64_t w = base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = st.s[i];
#pragma unroll
    for (int i = 0; i < 8; i++)
        if ((unsigned)i == pos) s[i] = w;
    poseidon_permute(s);
    if ((s[7] >> (64 - pow_bits)) == 0) atomicMin(best, (unsigned long long)w);
}

// ------------------------------------------------------------------ query gather
struct gather_oracle { const gl_t* lde; const gl_t* digests; uint32_t ncols, nsib; uint64_t N; uint64_t level_off[32]; };
struct gather_layer { const gl_t *c0, *c1, *digests; uint32_t nsib; uint64_t level_off[32]; };
struct gather_args {
    gather_oracle o[3];
    gather_layer l[8];
    uint32_t nlayers, arity_bits;
    uint64_t query_words;
};
__global__ void k_gather_queries(gather_args g, const uint64_t* __restrict__ xs, gl_t* __restrict__ out) {
    uint64_t x = xs[blockIdx.x];
    gl_t* o = out + (size_t)blockIdx.x * g.query_words;
    for (int k = 0; k < 3; k++) {
        const gather_oracle& r = g.o[k];
        for (uint32_t c = threadIdx.x; c < r.ncols; c += blockDim.x) o[c] = r.lde[(size_t)c * r.N + x];
        o += r.ncols;
        for (uint32_t e = threadIdx.x; e < r.nsib * 4; e += blockDim.x) {
            uint32_t lvl = e >> 2;
            o[e] = r.digests[r.level_off[lvl] + 4 * ((x >> lvl) ^ 1) + (e & 3)];
        }
        o += r.nsib * 4;
    }
    uint32_t arity = 1u << g.arity_bits;
    for (uint32_t l = 0; l < g.nlayers; l++) {
        const gather_layer& r = g.l[l];
        x >>= g.arity_bits;
        for (uint32_t e = threadIdx.x; e < 2 * arity; e += blockDim.x) o[e] = (e & 1) ? r.c1[x * arity + (e >> 1)] : r.c0[x * arity + (e >> 1)];
        o += 2 * arity;
        for (uint32_t e = threadIdx.x; e < r.nsib * 4; e += blockDim.x) {
            uint32_t lvl = e >> 2;
            o[e] = r.digests[r.level_off[lvl] + 4 * ((x >> lvl) ^ 1) + (e & 3)];
        }
        o += r.nsib * 4;
    }
}

// ------------------------------------------------------------------ host helpers
static gl2_t challenger_get_ext(zkm_challenger* ch) {
    gl_t a = zkm_challenger_get(ch), b = zkm_challenger_get(ch);
    return gl2_t{a, b};
}

struct fri_layer {
    gl_t* values = nullptr;   // [2][len] bit-reversed
    gl_t* digests = nullptr;
    std::vector<size_t> level_off;
    size_t len = 0;
    unsigned log_leaves = 0;
};

static void prove_single_table(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t W, unsigned log_n,
                               const zkm_batch* trace_batch, const uint64_t* aux, size_t A_ctl, const zkm_ctl_table* ctl_table,
                               const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t Z, const uint64_t* lookup_challenges,
                               zkm_challenger* ch, uint64_t* proof, const zkm_batch* aux_batch_in = nullptr,
                               const zkm_batch* quot_batch_in = nullptr) {
    // openings-only mode (zkm_prove_openings, BASELINE config 4): the three commitments exist already; the transcript
    // is compact -> zeta -> openings -> prove_openings
    const bool openings_only = aux_batch_in != nullptr;
    if (cfg->rate_bits != 2 || cfg->arity_bits < 2 || cfg->arity_bits > 6 || cfg->pow_bits == 0 || cfg->pow_bits > 32)
        throw std::runtime_error("zkm_prove_single_table: unsupported FRI configuration");
    // the table's own lookup helper columns come first among the auxiliary polynomials (prover.rs:467-508)
    const size_t NL = openings_only ? 0 : zkm_num_lookup_columns(table_id, cfg);
    if (NL && !lookup_challenges) throw std::runtime_error("this table has lookups: lookup challenges are required");
    if (NL && !trace) throw std::runtime_error("this table has lookups: the trace values are required to build their helper columns");
    if (!NL) lookup_challenges = nullptr;
    const size_t A = NL + A_ctl;
    proof_layout y;
    make_layout(y, cfg, log_n, W, A, Z);
    if (y.L > 8) throw std::runtime_error("too many FRI layers");
    size_t n = (size_t)1 << log_n, N = (size_t)1 << y.lde_bits;
    if (openings_only) {
        if (Z > A) throw std::runtime_error("zkm_prove_openings: more CTL Zs than auxiliary polynomials");
    } else {
        size_t ctl_helpers = 0;
        for (size_t i = 0; i < Z; i++) ctl_helpers += zs[i].num_helpers;
        if (ctl_helpers + Z != A_ctl) throw std::runtime_error("No CTL? aux column count does not match the CTL description");
    }
    if (A_ctl == 0) throw std::runtime_error("No CTL? aux column count does not match the CTL description");  // prover.rs:509
    const size_t total_helpers = A - Z;  // index of the first CTL Z among the auxiliary polynomials
    ctl_dev_owner own;
    if (!openings_only) own.upload(c, ctl_table, zs, colset_ids, Z);
    if (ctl_table && !openings_only)
        for (size_t i = 0; i < ctl_table->nterms; i++)
            if (ctl_table->term_col[i] >= W) throw std::runtime_error("CTL description: trace column index out of range");

    memset(proof, 0, y.total * sizeof(uint64_t));
    proof[0] = ZKM_PROOF_MAGIC; proof[1] = log_n; proof[2] = W; proof[3] = A; proof[4] = y.Q; proof[5] = Z; proof[6] = y.cap;
    proof[7] = y.L; proof[8] = y.F; proof[9] = y.nq; proof[10] = cfg->rate_bits; proof[11] = cfg->arity_bits;

    zkm_batch* own_trace = nullptr;
    zkm_batch *ab = nullptr, *qb = nullptr;
    std::vector<fri_layer> layers(y.L);
    std::vector<void*> scratch;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        zkm_batch_free(own_trace);
        zkm_batch_free(ab);
        zkm_batch_free(qb);
        for (auto& l : layers) { c->release(l.values); c->release(l.digests); }
        for (void* p : scratch) c->release(p);
    };
    try {
        const zkm_batch* tb = trace_batch;
        if (!tb) {
            own_trace = new zkm_batch();
            own_trace->ctx = c; own_trace->ncols = W; own_trace->log_n = log_n; own_trace->rate_bits = cfg->rate_bits;
            own_trace->cap_height = cfg->cap_height;
            zkm_batch_build(own_trace, trace, true);
            tb = own_trace;
        }
        if (tb->ncols != W || tb->log_n != log_n || tb->rate_bits != cfg->rate_bits || tb->cap_height != cfg->cap_height)
            throw std::runtime_error("trace commitment does not match the table shape / config");

        zkm_challenger_compact(ch, proof + y.o_init);  // :466
        uint64_t* caps = proof + y.o_caps;
        memcpy(caps, tb->cap.data(), y.C * 4 * 8);
        const zkm_batch *abp = aux_batch_in, *qbp = quot_batch_in;
        if (!openings_only) {
            // auxiliary commitment :511-522
            ab = new zkm_batch();
            ab->ctx = c; ab->ncols = A; ab->log_n = log_n; ab->rate_bits = cfg->rate_bits; ab->cap_height = cfg->cap_height;
            if (NL) {
                // "compute lookup helper columns" :475-493, then the CTL columns behind them
                gl_t* d_all = (gl_t*)c->alloc(A * n * sizeof(gl_t));
                scratch.push_back(d_all);
                const gl_t* d_trace = trace;
                if (!zkm_is_device_ptr(trace)) {
                    gl_t* d = (gl_t*)c->alloc(W * n * sizeof(gl_t));
                    scratch.push_back(d);
                    ZKM_HIP_CHECK(hipMemcpyAsync(d, trace, W * n * sizeof(gl_t), hipMemcpyHostToDevice, c->stream));
                    d_trace = d;
                }
                zkm_table_lookup_columns_device(c, table_id, lookup_challenges, cfg->num_challenges, d_trace, n, d_all);
                ZKM_HIP_CHECK(hipMemcpyAsync(d_all + NL * n, aux, A_ctl * n * sizeof(gl_t), hipMemcpyDefault, c->stream));
                zkm_batch_build(ab, d_all, true);
            } else {
                zkm_batch_build(ab, aux, true);
            }
            memcpy(caps + y.C * 4, ab->cap.data(), y.C * 4 * 8);
            zkm_challenger_observe(ch, caps + y.C * 4, y.C * 4);  // :525
            gl_t alphas[4];
            for (unsigned i = 0; i < cfg->num_challenges; i++) alphas[i] = zkm_challenger_get(ch);  // :527

            // quotient :543-587
            gl_t* d_quot = (gl_t*)c->alloc(cfg->num_challenges * 2 * n * sizeof(gl_t));
            scratch.push_back(d_quot);
            quotient_device(c, table_id, tb, ab, own, lookup_challenges, alphas, cfg->num_challenges, d_quot);
            qb = new zkm_batch();
            qb->ctx = c; qb->ncols = y.Q; qb->log_n = log_n; qb->rate_bits = cfg->rate_bits; qb->cap_height = cfg->cap_height;
            zkm_batch_build(qb, d_quot, false);  // chunks [q0_lo, q0_hi, q1_lo, q1_hi] == d_quot viewed as Q columns of n
            memcpy(caps + 2 * y.C * 4, qb->cap.data(), y.C * 4 * 8);
            zkm_challenger_observe(ch, caps + 2 * y.C * 4, y.C * 4);  // :589
            abp = ab;
            qbp = qb;
        } else {
            for (const zkm_batch* b : {abp, qbp})
                if (!b || b->log_n != log_n || b->rate_bits != cfg->rate_bits || b->cap_height != cfg->cap_height)
                    throw std::runtime_error("zkm_prove_openings: commitments do not match the table shape / config");
            if (abp->ncols != A || qbp->ncols != y.Q) throw std::runtime_error("zkm_prove_openings: unexpected number of polynomials");
            memcpy(caps + y.C * 4, abp->cap.data(), y.C * 4 * 8);
            memcpy(caps + 2 * y.C * 4, qbp->cap.data(), y.C * 4 * 8);
        }

        gl2_t zeta = challenger_get_ext(ch);  // :591
        gl_t g = gl_root_of_unity(log_n);
        if (gl2_eq(gl2_exp_pow2(zeta, log_n), gl2_t{1, 0})) throw std::runtime_error("Opening point is in the subgroup.");  // :596-599
        gl2_t zeta_next = gl2_scalar_mul(zeta, g);

        // openings proof.rs:299-334
        uint64_t* op = proof + y.o_open;
        uint64_t *o_local = op, *o_next = op + 2 * W, *o_aux = op + 4 * W, *o_auxn = o_aux + 2 * A, *o_ctl = o_auxn + 2 * A, *o_quot = o_ctl + Z;
        {
            auto tv = eval_batch(c, tb, zeta, zeta_next);
            for (size_t i = 0; i < W; i++) {
                o_local[2 * i] = tv[i].at_z0.c0; o_local[2 * i + 1] = tv[i].at_z0.c1;
                o_next[2 * i] = tv[i].at_z1.c0; o_next[2 * i + 1] = tv[i].at_z1.c1;
            }
            auto av = eval_batch(c, abp, zeta, zeta_next);
            for (size_t i = 0; i < A; i++) {
                o_aux[2 * i] = av[i].at_z0.c0; o_aux[2 * i + 1] = av[i].at_z0.c1;
                o_auxn[2 * i] = av[i].at_z1.c0; o_auxn[2 * i + 1] = av[i].at_z1.c1;
                if (i >= total_helpers) o_ctl[i - total_helpers] = av[i].at_one;
            }
            auto qv = eval_batch(c, qbp, zeta, zeta_next);
            for (size_t i = 0; i < y.Q; i++) { o_quot[2 * i] = qv[i].at_z0.c0; o_quot[2 * i + 1] = qv[i].at_z0.c1; }
        }
        // observe_openings(to_fri_openings) proof.rs:336-367
        zkm_challenger_observe(ch, o_local, 2 * W);
        zkm_challenger_observe(ch, o_aux, 2 * A);
        zkm_challenger_observe(ch, o_quot, 2 * y.Q);
        zkm_challenger_observe(ch, o_next, 2 * W);
        zkm_challenger_observe(ch, o_auxn, 2 * A);
        for (size_t i = 0; i < Z; i++) { uint64_t e[2] = {o_ctl[i], 0}; zkm_challenger_observe(ch, e, 2); }

        // ---- prove_openings (App. A.8)
        gl2_t alpha = challenger_get_ext(ch);
        size_t np0 = W + A + y.Q, np1 = W + A, np2 = Z;
        std::vector<gl_t> apow(2 * (np0 + 1));
        {
            gl2_t p{1, 0};
            for (size_t j = 0; j <= np0; j++) { apow[2 * j] = p.c0; apow[2 * j + 1] = p.c1; p = gl2_mul(p, alpha); }
        }
        gl_t* d_apow = (gl_t*)c->alloc(apow.size() * sizeof(gl_t));
        scratch.push_back(d_apow);
        ZKM_HIP_CHECK(hipMemcpyAsync(d_apow, apow.data(), apow.size() * sizeof(gl_t), hipMemcpyHostToDevice, c->stream));
        gl_t* d_comp = (gl_t*)c->alloc(6 * n * sizeof(gl_t));
        scratch.push_back(d_comp);
        {
            zkm_prof_scope ps(c, "fri_combine");
            hipLaunchKernelGGL(k_fri_combine, dim3((n + 255) / 256), dim3(256), 0, c->stream, tb->coeffs, W, abp->coeffs, A, qbp->coeffs, y.Q,
                               total_helpers, d_apow, n, d_comp);
            ZKM_HIP_CHECK(hipGetLastError());
        }
        gl_t* d_fin = (gl_t*)c->alloc(2 * n * sizeof(gl_t));  // final poly coefficients [2][n]
        scratch.push_back(d_fin);
        ZKM_HIP_CHECK(hipMemsetAsync(d_fin, 0, 2 * n * sizeof(gl_t), c->stream));
        auto apow_at = [&](size_t j) { return gl2_t{apow[2 * j], apow[2 * j + 1]}; };
        divide_accumulate(c, d_comp, d_comp + n, n, zeta, apow_at(np0), d_fin, d_fin + n);
        divide_accumulate(c, d_comp + 2 * n, d_comp + 3 * n, n, zeta_next, apow_at(np1), d_fin, d_fin + n);
        divide_accumulate(c, d_comp + 4 * n, d_comp + 5 * n, n, gl2_t{1, 0}, apow_at(np2), d_fin, d_fin + n);

        // commit phase: coefficients stay in d_fin (length clen, implicitly zero-padded x4)
        size_t clen = n;
        unsigned clog = log_n;
        gl_t shift = GL_GENERATOR;
        unsigned arity = 1u << cfg->arity_bits;
        gl_t* d_coef0 = d_fin;      // c0 array (clen)
        gl_t* d_coef1 = d_fin + n;  // c1 array
        for (unsigned l = 0; l < y.L; l++) {
            fri_layer& fl = layers[l];
            fl.len = clen << cfg->rate_bits;
            fl.values = (gl_t*)c->alloc(2 * fl.len * sizeof(gl_t));
            // values = coset_fft(shift) of the zero-padded coefficients, bit-reversed: two base-field columns
            // (the two coefficient arrays are contiguous: [c0 | c1], column stride clen)
            if (d_coef1 != d_coef0 + clen) throw std::runtime_error("internal: FRI coefficient arrays not contiguous");
            zkm_lde_bitrev(c, d_coef0, fl.values, 2, clog, cfg->rate_bits, shift);
            fl.log_leaves = clog + cfg->rate_bits - cfg->arity_bits;
            size_t dwords = zkm_merkle_layout(fl.log_leaves, cfg->cap_height, fl.level_off);
            fl.digests = (gl_t*)c->alloc(dwords * sizeof(gl_t));
            zkm_launch_merkle_leaves_ext(c, fl.values, fl.values + fl.len, (size_t)1 << fl.log_leaves, arity, fl.digests);
            zkm_merkle_build_inner(c, fl.digests, fl.level_off, fl.log_leaves, cfg->cap_height);
            uint64_t* capo = proof + y.o_fri_caps + l * y.C * 4;
            ZKM_HIP_CHECK(hipMemcpyAsync(capo, fl.digests + fl.level_off[fl.log_leaves - cfg->cap_height], y.C * 4 * 8, hipMemcpyDeviceToHost, c->stream));
            c->sync();
            zkm_challenger_observe(ch, capo, y.C * 4);
            gl2_t beta = challenger_get_ext(ch);
            size_t nout = clen >> cfg->arity_bits;
            gl_t* d_new = (gl_t*)c->alloc(2 * nout * sizeof(gl_t));
            scratch.push_back(d_new);
            {
                zkm_prof_scope ps(c, "fri_fold");
                hipLaunchKernelGGL(k_fri_fold, dim3((nout + 255) / 256), dim3(256), 0, c->stream, d_coef0, d_coef1, nout, arity, beta, d_new, d_new + nout);
                ZKM_HIP_CHECK(hipGetLastError());
            }
            d_coef0 = d_new;
            d_coef1 = d_new + nout;
            clen = nout;
            clog -= cfg->arity_bits;
            shift = gl_pow(shift, arity);
        }
        if (clen != y.F) throw std::runtime_error("internal: final polynomial length mismatch");
        {
            std::vector<gl_t> f(2 * clen);
            ZKM_HIP_CHECK(hipMemcpyAsync(f.data(), d_coef0, clen * 8, hipMemcpyDeviceToHost, c->stream));
            ZKM_HIP_CHECK(hipMemcpyAsync(f.data() + clen, d_coef1, clen * 8, hipMemcpyDeviceToHost, c->stream));
            c->sync();
            uint64_t* fp = proof + y.o_final;
            for (size_t i = 0; i < clen; i++) { fp[2 * i] = f[i]; fp[2 * i + 1] = f[clen + i]; }
            zkm_challenger_observe(ch, fp, 2 * clen);
        }

        // proof of work (App. A.9), smallest witness
        {
            pow_state st;
            memcpy(st.s, ch->state, sizeof st.s);
            for (uint32_t i = 0; i < ch->n_in; i++) st.s[i] = ch->in_buf[i];
            unsigned long long* d_best = (unsigned long long*)c->alloc(8);
            scratch.push_back(d_best);
            unsigned long long best = ~0ULL;
            const uint64_t span = (uint64_t)1 << 20;
            for (uint64_t base = 0; best == ~0ULL; base += span) {
                if (base > ((uint64_t)1 << 40)) throw std::runtime_error("Proof of work failed. This is highly unlikely!");
                ZKM_HIP_CHECK(hipMemsetAsync(d_best, 0xff, 8, c->stream));
                {
                    zkm_prof_scope ps(c, "fri_pow_search");
                    hipLaunchKernelGGL(k_pow_search, dim3(span / 256), dim3(256), 0, c->stream, st, ch->n_in, cfg->pow_bits, base, d_best);
                    ZKM_HIP_CHECK(hipGetLastError());
                }
                ZKM_HIP_CHECK(hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, c->stream));
                c->sync();
            }
            uint64_t w = best;
            proof[y.o_pow] = w;
            zkm_challenger_observe(ch, &w, 1);
            uint64_t resp = zkm_challenger_get(ch);
            if ((resp >> (64 - cfg->pow_bits)) != 0) throw std::runtime_error("internal: proof-of-work response check failed");
        }

        // query rounds
        {
            std::vector<uint64_t> xs(y.nq);
            for (size_t q = 0; q < y.nq; q++) xs[q] = zkm_challenger_get(ch) % N;
            uint64_t* d_xs = (uint64_t*)c->alloc(y.nq * 8);
            scratch.push_back(d_xs);
            ZKM_HIP_CHECK(hipMemcpyAsync(d_xs, xs.data(), y.nq * 8, hipMemcpyHostToDevice, c->stream));
            gather_args ga{};
            const zkm_batch* orc[3] = {tb, abp, qbp};
            for (int k = 0; k < 3; k++) {
                ga.o[k].lde = orc[k]->lde; ga.o[k].digests = orc[k]->digests; ga.o[k].ncols = (uint32_t)orc[k]->ncols;
                ga.o[k].nsib = y.lde_bits - y.cap; ga.o[k].N = N;
                for (size_t i = 0; i < orc[k]->level_off.size() && i < 32; i++) ga.o[k].level_off[i] = orc[k]->level_off[i];
            }
            for (unsigned l = 0; l < y.L; l++) {
                ga.l[l].c0 = layers[l].values; ga.l[l].c1 = layers[l].values + layers[l].len; ga.l[l].digests = layers[l].digests;
                ga.l[l].nsib = layers[l].log_leaves - y.cap;
                for (size_t i = 0; i < layers[l].level_off.size() && i < 32; i++) ga.l[l].level_off[i] = layers[l].level_off[i];
            }
            ga.nlayers = y.L; ga.arity_bits = cfg->arity_bits; ga.query_words = y.query_words;
            gl_t* d_q = (gl_t*)c->alloc(y.nq * y.query_words * 8);
            scratch.push_back(d_q);
            {
                zkm_prof_scope ps(c, "fri_gather_queries");
                hipLaunchKernelGGL(k_gather_queries, dim3(y.nq), dim3(256), 0, c->stream, ga, d_xs, d_q);
                ZKM_HIP_CHECK(hipGetLastError());
            }
            ZKM_HIP_CHECK(hipMemcpyAsync(proof + y.o_queries, d_q, y.nq * y.query_words * 8, hipMemcpyDeviceToHost, c->stream));
            c->sync();
        }
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

// ------------------------------------------------------------------ C ABI
static int fail(char** err, const std::string& msg) {
    if (err) {
        *err = (char*)malloc(msg.size() + 1);
        if (*err) memcpy(*err, msg.c_str(), msg.size() + 1);
    }
    return 1;
}

extern "C" {

size_t zkm_proof_words(const zkm_stark_config* cfg, unsigned log_n, size_t ncols, size_t naux, size_t nctl_zs) {
    proof_layout y;
    make_layout(y, cfg, log_n, ncols, naux, nctl_zs);
    return y.total;
}

// CtlZData of the benchmark's fake CTL shape: helper columns, no column sets (poseidon_stark.rs:786-799)
static std::vector<zkm_ctl_z> fake_zs(const uint32_t* num_helpers, size_t n) {
    std::vector<zkm_ctl_z> zs(n);
    for (size_t i = 0; i < n; i++) {
        if (num_helpers[i] == 0) throw std::runtime_error("CTLs without helper columns need column sets: use zkm_prove_single_table_ctl");
        zs[i] = zkm_ctl_z{0, 0, num_helpers[i], 0, 0, 0};
    }
    return zs;
}

int zkm_prove_single_table_ctl(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                               const zkm_batch* trace_batch, const uint64_t* aux, size_t naux, const zkm_ctl_table* table,
                               const zkm_ctl_z* zs, const uint32_t* colset_ids, size_t nzs, const uint64_t* lookup_challenges,
                               zkm_challenger* challenger, uint64_t* proof_out, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!trace && !trace_batch) throw std::runtime_error("zkm_prove_single_table: need trace values or a trace commitment");
        prove_single_table(c, table_id, cfg, trace, ncols, log_n, trace_batch, aux, naux, table, zs, colset_ids, nzs, lookup_challenges,
                           challenger, proof_out);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
    return 0;
}

int zkm_prove_openings(zkm_ctx* c, const zkm_stark_config* cfg, const zkm_batch* trace_batch, const zkm_batch* aux_batch,
                       const zkm_batch* quot_batch, size_t nctl_zs, zkm_challenger* challenger, uint64_t* proof_out, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        if (!trace_batch || !aux_batch || !quot_batch) throw std::runtime_error("zkm_prove_openings: three commitments are required");
        prove_single_table(c, -1, cfg, nullptr, trace_batch->ncols, trace_batch->log_n, trace_batch, nullptr, aux_batch->ncols, nullptr,
                           nullptr, nullptr, nctl_zs, nullptr, challenger, proof_out, aux_batch, quot_batch);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
    return 0;
}

int zkm_prove_single_table(zkm_ctx* c, int table_id, const zkm_stark_config* cfg, const uint64_t* trace, size_t ncols, unsigned log_n,
                           const zkm_batch* trace_batch, const uint64_t* aux, size_t naux, const uint32_t* num_helpers, size_t nctl_zs,
                           zkm_challenger* challenger, uint64_t* proof_out, char** err) {
    try {
        auto zs = fake_zs(num_helpers, nctl_zs);
        return zkm_prove_single_table_ctl(c, table_id, cfg, trace, ncols, log_n, trace_batch, aux, naux, nullptr, zs.data(), nullptr,
                                          nctl_zs, nullptr, challenger, proof_out, err);
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
}

int zkm_quotient(zkm_ctx* c, int table_id, const zkm_batch* trace, const zkm_batch* aux, const uint32_t* num_helpers, size_t nctl_zs,
                 const uint64_t* alphas, size_t nalphas, uint64_t* out_coeffs, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        auto zs = fake_zs(num_helpers, nctl_zs);
        ctl_dev_owner own;
        own.upload(c, nullptr, zs.data(), nullptr, nctl_zs);
        size_t words = nalphas * 2 * trace->n();
        bool dev = zkm_is_device_ptr(out_coeffs);
        gl_t* d = dev ? out_coeffs : (gl_t*)c->alloc(words * 8);
        quotient_device(c, table_id, trace, aux, own, nullptr, alphas, nalphas, d);
        if (!dev) {
            ZKM_HIP_CHECK(hipMemcpyAsync(out_coeffs, d, words * 8, hipMemcpyDeviceToHost, c->stream));
            c->sync();
            c->release(d);
        }
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
    return 0;
}

int zkm_eval_openings(zkm_ctx* c, const zkm_batch* b, const uint64_t zeta[2], uint64_t* out, char** err) {
    try {
        ZKM_HIP_CHECK(hipSetDevice(c->device));
        gl2_t z{zeta[0], zeta[1]};
        auto v = eval_batch(c, b, z, z);
        for (size_t i = 0; i < b->ncols; i++) { out[2 * i] = v[i].at_z0.c0; out[2 * i + 1] = v[i].at_z0.c1; }
    } catch (const std::exception& e) {
        return fail(err, e.what());
    }
    return 0;
}

}  // extern "C"
