/* oracle/constraints_tmpl.h -- constraint evaluation, instantiated for F (prover, quotient domain)
 * and for F2 (verifier, at zeta).  Include with these macros defined:
 *   T            element type            TNAME(x)    name mangler
 *   T_ADD/T_SUB/T_MUL(a,b)               T_MULB(a,s) multiply by base-field scalar s
 *   T_FROMB(s)   embed base scalar
 *
 * TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restates, in emission order (= alpha-power order, constraint_consumer.rs:57-62):
 *   ConstraintConsumer            prover/src/constraint_consumer.rs:10-75
 *   PoseidonStark constraints     prover/src/poseidon/poseidon_stark.rs:554-594 (eval_packed_generic) with
 *     constant_layer_field :171-176, sbox_field :191-199, sbox_layer_field :253-275, mds_layer_field :295-308,
 *     partial_first_constant_layer :371-375, mds_partial_layer_init_field :406-419,
 *     partial_sbox_layer :442-454, mds_partial_layer_fast_field :503-519
 *   CTL checks (helper-column shape) prover/src/cross_table_lookup.rs:1067-1118, eval_helper_columns :1006-1058
 *   order table -> lookups -> CTL   prover/src/vanishing_poly.rs:30-45
 */

typedef struct {
    size_t nalphas;
    gl_t alphas[4];
    T acc[4];
    T z_last, l_first, l_last;
    T* rec; /* debugging aid: when set, every emitted constraint value is also stored here (zko_debug_constraints) */
    size_t nrec, rec_cap;
} TNAME(consumer);

static inline void TNAME(cons)(TNAME(consumer) * k, T c) {
    if (k->rec) {
        if (k->nrec < k->rec_cap) k->rec[k->nrec] = c;
        k->nrec++;
    }
    for (size_t j = 0; j < k->nalphas; j++) k->acc[j] = T_ADD(T_MULB(k->acc[j], k->alphas[j]), c);
}
static inline void TNAME(cons_transition)(TNAME(consumer) * k, T c) { TNAME(cons)(k, T_MUL(c, k->z_last)); }
static inline void TNAME(cons_first)(TNAME(consumer) * k, T c) { TNAME(cons)(k, T_MUL(c, k->l_first)); }
static inline void TNAME(cons_last)(TNAME(consumer) * k, T c) { TNAME(cons)(k, T_MUL(c, k->l_last)); }

static inline void TNAME(sbox_constr)(TNAME(consumer) * k, T in, T inter, T out) {
    TNAME(cons)(k, T_SUB(T_MUL(T_MUL(in, in), in), inter));
    TNAME(cons)(k, T_SUB(T_MUL(T_MUL(in, inter), inter), out));
}

static void TNAME(mds_layer)(T s[12]) {
    T r[12];
    for (int i = 0; i < 12; i++) {
        T acc = T_FROMB(0);
        for (int j = 0; j < 12; j++) acc = T_ADD(acc, T_MULB(s[(j + i) % 12], ZKM_POSEIDON_MDS_CIRC[j]));
        acc = T_ADD(acc, T_MULB(s[i], ZKM_POSEIDON_MDS_DIAG[i]));
        r[i] = acc;
    }
    for (int i = 0; i < 12; i++) s[i] = r[i];
}

static void TNAME(eval_poseidon)(const T* lv, TNAME(consumer) * k) {
    T s[12];
    for (int i = 0; i < 12; i++) s[i] = lv[1 + i];
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = T_ADD(s[i], T_FROMB(gl_canon(ZKM_POSEIDON_RC[i + 12 * rc])));
        for (int i = 0; i < 12; i++) {
            T tmp = lv[26 + 24 * r + 2 * i], out = lv[26 + 24 * r + 2 * i + 1];
            TNAME(sbox_constr)(k, s[i], tmp, out);
            s[i] = out;
        }
        TNAME(mds_layer)(s);
    }
    for (int i = 0; i < 12; i++) s[i] = T_ADD(s[i], T_FROMB(ZKM_POSEIDON_FAST_FIRST_RC[i]));
    {
        T t[12];
        t[0] = s[0];
        for (int c = 1; c < 12; c++) t[c] = T_FROMB(0);
        for (int r = 1; r < 12; r++)
            for (int c = 1; c < 12; c++) t[c] = T_ADD(t[c], T_MULB(s[r], ZKM_POSEIDON_FAST_INIT[r - 1][c - 1]));
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
    for (int r = 0; r < 22; r++) {
        T inter = lv[122 + 2 * r], out = lv[122 + 2 * r + 1];
        TNAME(sbox_constr)(k, s[0], inter, out);
        s[0] = out;
        if (r < 21) s[0] = T_ADD(s[0], T_FROMB(ZKM_POSEIDON_FAST_RC[r]));
        T d = T_MULB(s[0], ZKM_POSEIDON_MDS_CIRC[0] + ZKM_POSEIDON_MDS_DIAG[0]);
        for (int i = 1; i < 12; i++) d = T_ADD(d, T_MULB(s[i], ZKM_POSEIDON_FAST_W_HATS[r][i - 1]));
        for (int i = 1; i < 12; i++) s[i] = T_ADD(T_MULB(s[0], ZKM_POSEIDON_FAST_VS[r][i - 1]), s[i]);
        s[0] = d;
    }
    rc += 22;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = T_ADD(s[i], T_FROMB(gl_canon(ZKM_POSEIDON_RC[i + 12 * rc])));
        for (int i = 0; i < 12; i++) {
            T tmp = lv[166 + 24 * r + 2 * i], out = lv[166 + 24 * r + 2 * i + 1];
            TNAME(sbox_constr)(k, s[i], tmp, out);
            s[i] = out;
        }
        TNAME(mds_layer)(s);
    }
    for (int i = 0; i < 12; i++) TNAME(cons)(k, T_SUB(s[i], lv[13 + i]));
}

/* ---- LogicStark constraints: logic.rs:199-248 (columns :25-50: IS_AND 0, IS_OR 1, IS_XOR 2, IS_NOR 3, INPUT0 bits 4..35,
 * INPUT1 bits 36..67, RESULT 68; VAL_BITS = PACKED_LIMB_BITS = 32, PACKED_LEN = 1) ---- */
static void TNAME(eval_logic)(const T* lv, TNAME(consumer) * k) {
    T is_and = lv[0], is_or = lv[1], is_xor = lv[2], is_nor = lv[3];
    T sum_coeff = T_SUB(T_ADD(is_or, is_xor), is_nor);
    T and_coeff = T_ADD(T_SUB(T_SUB(is_and, is_or), T_MULB(is_xor, 2)), is_nor);
    T not_coeff = is_nor;
    for (int i = 4; i < 68; i++) TNAME(cons)(k, T_MUL(lv[i], T_SUB(lv[i], T_FROMB(1))));
    T x = T_FROMB(0), y = T_FROMB(0), x_land_y = T_FROMB(0);
    for (int i = 0; i < 32; i++) {
        x = T_ADD(x, T_MULB(lv[4 + i], (gl_t)1 << i));
        y = T_ADD(y, T_MULB(lv[36 + i], (gl_t)1 << i));
        x_land_y = T_ADD(x_land_y, T_MULB(T_MUL(lv[4 + i], lv[36 + i]), (gl_t)1 << i));
    }
    T x_op_y = T_ADD(T_ADD(T_MUL(sum_coeff, T_ADD(x, y)), T_MUL(and_coeff, x_land_y)), T_MULB(not_coeff, 0xFFFFFFFFULL));
    TNAME(cons)(k, T_SUB(lv[68], x_op_y));
}

/* ---- KeccakSpongeStark constraints: keccak_sponge_stark.rs:456-567 (column map keccak_sponge/columns.rs:19-70:
 * full 0, context 1, segment 2, virt 3..36, timestamp 37, len 38, already_absorbed 39, is_final_input_len 40..175,
 * original_rate 176..209, original_capacity 210..225, block_bytes 226..361, xored_rate 362..395,
 * partial_updated_state 396..437, updated_digest_state_bytes 438..469) ---- */
static void TNAME(eval_keccak_sponge)(const T* lv, const T* nv, TNAME(consumer) * k) {
    T one = T_FROMB(1);
    T full = lv[0];
    TNAME(cons)(k, T_MUL(full, T_SUB(full, one)));
    T is_final = T_FROMB(0);
    for (int i = 0; i < 136; i++) is_final = T_ADD(is_final, lv[40 + i]);
    TNAME(cons)(k, T_MUL(is_final, T_SUB(is_final, one)));
    for (int i = 0; i < 136; i++) TNAME(cons)(k, T_MUL(lv[40 + i], T_SUB(lv[40 + i], one)));
    TNAME(cons)(k, T_MUL(is_final, full));
    T absorbed = lv[39];
    TNAME(cons_first)(k, absorbed);
    for (int i = 0; i < 34; i++) TNAME(cons_first)(k, lv[176 + i]);
    for (int i = 0; i < 16; i++) TNAME(cons_first)(k, lv[210 + i]);
    TNAME(cons_transition)(k, T_MUL(is_final, nv[39]));
    for (int i = 0; i < 34; i++) TNAME(cons_transition)(k, T_MUL(is_final, nv[176 + i]));
    for (int i = 0; i < 16; i++) TNAME(cons_transition)(k, T_MUL(is_final, nv[210 + i]));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(lv[1], nv[1])));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(lv[2], nv[2])));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(lv[37], nv[37])));
    for (int l = 0; l < 8; l++) {
        T cur = lv[438 + 4 * l];
        for (int i = 1; i < 4; i++) cur = T_ADD(cur, T_MULB(lv[438 + 4 * l + i], (gl_t)1 << (8 * i)));
        TNAME(cons_transition)(k, T_MUL(full, T_SUB(nv[176 + l], cur)));
    }
    for (int i = 0; i < 26; i++) TNAME(cons_transition)(k, T_MUL(full, T_SUB(nv[176 + 8 + i], lv[396 + i])));
    for (int i = 0; i < 16; i++) TNAME(cons_transition)(k, T_MUL(full, T_SUB(nv[210 + i], lv[396 + 26 + i])));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(T_ADD(absorbed, T_FROMB(136)), nv[39])));
    T is_dummy = T_SUB(T_SUB(one, full), is_final);
    T next_final = T_FROMB(0);
    for (int i = 0; i < 136; i++) next_final = T_ADD(next_final, nv[40 + i]);
    TNAME(cons_transition)(k, T_MUL(is_dummy, T_ADD(nv[0], next_final)));
    T offset = T_SUB(lv[38], absorbed);
    for (int i = 0; i < 136; i++) TNAME(cons)(k, T_MUL(lv[40 + i], T_SUB(offset, T_FROMB((gl_t)i))));
}

/* ---- KeccakStark constraints: keccak/keccak_stark.rs:256-413 (797 = 3 + 320 + 50 + 320 + 50 + 4 + 50), register map
 * keccak/columns.rs:7-134, xor_gen / xor3_gen / andn_gen keccak/logic.rs:16-54, round-constant bits keccak/constants.rs ---- */
#ifndef ZKO_KECCAK_REGS
#define ZKO_KECCAK_REGS
enum { KK_TIMESTAMP = 24, KK_A = 25, KK_C = 75, KK_CP = 395, KK_AP = 715, KK_APP = 2315, KK_APP00_BITS = 2365, KK_APPP00 = 2429, KK_COLS = 2431 };
static inline int kk_a(int x, int y) { return KK_A + (x * 5 + y) * 2; }
static inline int kk_c(int x, int z) { return KK_C + x * 64 + z; }
static inline int kk_cp(int x, int z) { return KK_CP + x * 64 + z; }
static inline int kk_ap(int x, int y, int z) { return KK_AP + x * 320 + y * 64 + z; }
static inline int kk_b(int x, int y, int z) { int a = (x + 3 * y) % 5, b = x; return kk_ap(a, b, (z + 64 - ZKO_KECCAK_R[a][b]) % 64); }
static inline int kk_app(int x, int y) { return KK_APP + x * 10 + y * 2; }
static inline int kk_appp(int x, int y) { return (x == 0 && y == 0) ? KK_APPP00 : kk_app(x, y); }
#endif
static inline T TNAME(xor_gen)(T x, T y) { return T_SUB(T_ADD(x, y), T_MUL(x, T_ADD(y, y))); }
static inline T TNAME(xor3_gen)(T x, T y, T z) { return TNAME(xor_gen)(x, TNAME(xor_gen)(y, z)); }
static void TNAME(eval_keccak)(const T* lv, const T* nv, TNAME(consumer) * k) {
    T one = T_FROMB(1);
    T filter = lv[23];
    TNAME(cons)(k, T_MUL(filter, T_SUB(filter, one)));
    T not_final = T_SUB(one, lv[23]);
    TNAME(cons)(k, T_MUL(not_final, filter));
    T sum_flags = T_FROMB(0);
    for (int i = 0; i < 24; i++) sum_flags = T_ADD(sum_flags, lv[i]);
    TNAME(cons)(k, T_MUL(T_MUL(sum_flags, not_final), T_SUB(nv[KK_TIMESTAMP], lv[KK_TIMESTAMP])));
    for (int x = 0; x < 5; x++)
        for (int z = 0; z < 64; z++) {
            T xr = TNAME(xor3_gen)(lv[kk_c(x, z)], lv[kk_c((x + 4) % 5, z)], lv[kk_c((x + 1) % 5, (z + 63) % 64)]);
            TNAME(cons)(k, T_SUB(lv[kk_cp(x, z)], xr));
        }
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++)
            for (int half = 0; half < 2; half++) {
                T acc = T_FROMB(0);
                for (int z = 32 * half + 31; z >= 32 * half; z--)
                    acc = T_ADD(T_ADD(acc, acc), TNAME(xor3_gen)(lv[kk_ap(x, y, z)], lv[kk_c(x, z)], lv[kk_cp(x, z)]));
                TNAME(cons)(k, T_SUB(acc, lv[kk_a(x, y) + half]));
            }
    for (int x = 0; x < 5; x++)
        for (int z = 0; z < 64; z++) {
            T sum = T_FROMB(0);
            for (int i = 0; i < 5; i++) sum = T_ADD(sum, lv[kk_ap(x, i, z)]);
            T diff = T_SUB(sum, lv[kk_cp(x, z)]);
            TNAME(cons)(k, T_MUL(T_MUL(diff, T_SUB(diff, T_FROMB(2))), T_SUB(diff, T_FROMB(4))));
        }
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++)
            for (int half = 0; half < 2; half++) {
                T acc = T_FROMB(0);
                for (int z = 32 * half + 31; z >= 32 * half; z--) {
                    T andn = T_MUL(T_SUB(one, lv[kk_b((x + 1) % 5, y, z)]), lv[kk_b((x + 2) % 5, y, z)]);
                    acc = T_ADD(T_ADD(acc, acc), TNAME(xor_gen)(lv[kk_b(x, y, z)], andn));
                }
                TNAME(cons)(k, T_SUB(acc, lv[kk_app(x, y) + half]));
            }
    for (int half = 0; half < 2; half++) {
        T acc = T_FROMB(0);
        for (int z = 32 * half + 31; z >= 32 * half; z--) acc = T_ADD(T_ADD(acc, acc), lv[KK_APP00_BITS + z]);
        TNAME(cons)(k, T_SUB(acc, lv[kk_app(0, 0) + half]));
    }
    for (int half = 0; half < 2; half++) {
        T acc = T_FROMB(0);
        for (int z = 32 * half + 31; z >= 32 * half; z--) {
            T rc_bit = T_FROMB(0);
            for (int r = 0; r < 24; r++) rc_bit = T_ADD(rc_bit, T_MULB(lv[r], (ZKO_KECCAK_RC[r] >> z) & 1));
            acc = T_ADD(T_ADD(acc, acc), TNAME(xor_gen)(lv[KK_APP00_BITS + z], rc_bit));
        }
        TNAME(cons)(k, T_SUB(acc, lv[KK_APPP00 + half]));
    }
    T not_last = T_SUB(one, lv[23]);
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++)
            for (int half = 0; half < 2; half++)
                TNAME(cons_transition)(k, T_MUL(not_last, T_SUB(lv[kk_appp(x, y) + half], nv[kk_a(x, y) + half])));
}

/* ---- PoseidonSpongeStark constraints: poseidon_sponge/poseidon_sponge_stark.rs:383-478 (column map poseidon_sponge/columns.rs:17-66:
 * full 0, context 1, segment 2, virt 3..10, timestamp 11, len 12, already_absorbed 13, is_final_input_len 14..45, original_rate 46..53,
 * original_capacity 54..57, block_bytes 58..89, new_rate 90..97, partial_updated_state 98..105, updated_digest_state 106..109) ---- */
static void TNAME(eval_poseidon_sponge)(const T* lv, const T* nv, TNAME(consumer) * k) {
    T one = T_FROMB(1);
    T full = lv[0];
    TNAME(cons)(k, T_MUL(full, T_SUB(full, one)));
    T is_final = T_FROMB(0), next_final = T_FROMB(0);
    for (int i = 0; i < 32; i++) { is_final = T_ADD(is_final, lv[14 + i]); next_final = T_ADD(next_final, nv[14 + i]); }
    TNAME(cons)(k, T_MUL(is_final, T_SUB(is_final, one)));
    for (int i = 0; i < 32; i++) TNAME(cons)(k, T_MUL(lv[14 + i], T_SUB(lv[14 + i], one)));
    TNAME(cons)(k, T_MUL(is_final, full));
    T absorbed = lv[13];
    TNAME(cons_first)(k, absorbed);
    for (int i = 0; i < 12; i++) TNAME(cons_first)(k, lv[46 + i]);     /* original_rate then original_capacity */
    TNAME(cons_transition)(k, T_MUL(is_final, nv[13]));
    for (int i = 0; i < 12; i++) TNAME(cons_transition)(k, T_MUL(is_final, nv[46 + i]));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(lv[1], nv[1])));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(lv[2], nv[2])));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(lv[11], nv[11])));
    for (int i = 0; i < 4; i++) TNAME(cons_transition)(k, T_MUL(full, T_SUB(nv[46 + i], lv[106 + i])));
    for (int i = 0; i < 4; i++) TNAME(cons_transition)(k, T_MUL(full, T_SUB(nv[50 + i], lv[98 + i])));
    for (int i = 0; i < 4; i++) TNAME(cons_transition)(k, T_MUL(full, T_SUB(nv[54 + i], lv[102 + i])));
    TNAME(cons_transition)(k, T_MUL(full, T_SUB(T_ADD(absorbed, T_FROMB(32)), nv[13])));
    T is_dummy = T_SUB(T_SUB(one, full), is_final);
    TNAME(cons_transition)(k, T_MUL(is_dummy, T_ADD(nv[0], next_final)));
    T offset = T_SUB(lv[12], absorbed);
    for (int i = 0; i < 32; i++) TNAME(cons)(k, T_MUL(lv[14 + i], T_SUB(offset, T_FROMB((gl_t)i))));
}

/* ---- ShaExtendStark constraints: sha_extend/sha_extend_stark.rs:238-317 with rotate_right.rs:29-62, shift_right.rs:29-60,
 * wrapping_add_4.rs:35-78.  Columns (sha_extend/columns.rs:8-36): w_i value 0..3 carry 4..7, w_i_minus_15 8..11, w_i_minus_2 12..15,
 * w_i_minus_16 16..19, w_i_minus_7 20..23, s_0_inter 24..27, s_0 28..31, s_1_inter 32..35, s_1 36..39, rotate/shift ops
 * (value[4], shift, carry): rr_7 40, rr_18 46, rr_17 52, rr_19 58, rs_10 64, rs_3 70; timestamp 76, is_real_round 77 ---- */
static inline T TNAME(le4)(const T* b) {
    return T_ADD(T_ADD(b[0], T_MULB(b[1], 1u << 8)), T_ADD(T_MULB(b[2], 1u << 16), T_MULB(b[3], 1u << 24)));
}
static void TNAME(sha_rot)(const T* in, const T* op, unsigned r, int is_shift, TNAME(consumer) * k) {
    T out = TNAME(le4)(op), inv = TNAME(le4)(in), shift = op[4], carry = op[5];
    if (is_shift) TNAME(cons)(k, T_SUB(out, shift));
    else TNAME(cons)(k, T_SUB(T_SUB(out, T_MULB(carry, (gl_t)1 << (32 - r))), shift));
    TNAME(cons)(k, T_SUB(T_SUB(inv, T_MULB(shift, (gl_t)1 << r)), carry));
}
static void TNAME(eval_sha_extend)(const T* lv, TNAME(consumer) * k) {
    TNAME(sha_rot)(lv + 8, lv + 40, 7, 0, k);
    TNAME(sha_rot)(lv + 8, lv + 46, 18, 0, k);
    TNAME(sha_rot)(lv + 12, lv + 52, 17, 0, k);
    TNAME(sha_rot)(lv + 12, lv + 58, 19, 0, k);
    TNAME(sha_rot)(lv + 8, lv + 70, 3, 1, k);
    TNAME(sha_rot)(lv + 12, lv + 64, 10, 1, k);
    /* w_i = s_1 + w_i_minus_7 + s_0 + w_i_minus_16, every constraint times is_real_round */
    T real = lv[77], one = T_FROMB(1);
    const T *a = lv + 36, *b = lv + 20, *c = lv + 28, *d = lv + 16, *val = lv, *cy = lv + 4;
    for (int i = 0; i < 4; i++) TNAME(cons)(k, T_MUL(T_MUL(cy[i], T_SUB(one, cy[i])), real));
    TNAME(cons)(k, T_MUL(T_SUB(T_ADD(T_ADD(cy[0], cy[1]), T_ADD(cy[2], cy[3])), one), real));
    T carry = T_ADD(T_ADD(cy[1], T_MULB(cy[2], 2)), T_MULB(cy[3], 3));
    T sum = T_FROMB(0);
    for (int i = 3; i >= 0; i--) sum = T_ADD(T_MULB(sum, 1u << 8), T_ADD(T_ADD(a[i], b[i]), T_ADD(c[i], d[i])));
    TNAME(cons)(k, T_MUL(T_SUB(T_SUB(sum, T_MULB(carry, (gl_t)1 << 32)), TNAME(le4)(val)), real));
}

/* ---- ShaExtendSpongeStark constraints: sha_extend_sponge/sha_extend_sponge_stark.rs:220-330.  Columns
 * (sha_extend_sponge/columns.rs:7-33): round 0..47, w_i_minus_15 48..51, w_i_minus_2 52..55, w_i_minus_16 56..59, w_i_minus_7 60..63,
 * w_i 64..67, input_virt 68..71, output_virt 72, context 73, segment 74, timestamp 75.  NUM_CHANNELS = 10 (cpu/membus.rs:10-32) ---- */
static void TNAME(eval_sha_extend_sponge)(const T* lv, const T* nv, TNAME(consumer) * k) {
    T one = T_FROMB(1);
    for (int i = 0; i < 48; i++) TNAME(cons)(k, T_MUL(lv[i], T_SUB(lv[i], one)));
    T is_final = lv[47];
    TNAME(cons)(k, T_MUL(is_final, T_SUB(is_final, one)));
    T not_final = T_SUB(one, is_final);
    T sum = T_FROMB(0), lidx = T_FROMB(0), nidx = T_FROMB(0);
    for (int i = 0; i < 48; i++) {
        sum = T_ADD(sum, lv[i]);
        lidx = T_ADD(lidx, T_MULB(lv[i], (gl_t)i));
        nidx = T_ADD(nidx, T_MULB(nv[i], (gl_t)i));
    }
    T g = T_MUL(sum, not_final);
    TNAME(cons)(k, T_MUL(g, T_SUB(T_SUB(nv[75], lv[75]), T_FROMB(20))));
    TNAME(cons)(k, T_MUL(g, T_SUB(T_SUB(nidx, lidx), one)));
    for (int i = 0; i < 4; i++) TNAME(cons)(k, T_MUL(g, T_SUB(T_SUB(nv[68 + i], lv[68 + i]), T_FROMB(4))));
    TNAME(cons)(k, T_MUL(g, T_SUB(T_SUB(nv[72], lv[72]), T_FROMB(4))));
    TNAME(cons)(k, T_MUL(sum, T_SUB(T_SUB(lv[68], lv[70]), T_FROMB(4))));
    TNAME(cons)(k, T_MUL(sum, T_SUB(T_SUB(lv[69], lv[70]), T_FROMB(56))));
    TNAME(cons)(k, T_MUL(sum, T_SUB(T_SUB(lv[71], lv[70]), T_FROMB(36))));
    TNAME(cons)(k, T_MUL(sum, T_SUB(T_SUB(lv[72], lv[70]), T_FROMB(64))));
}

/* ---- ShaCompressStark constraints: sha_compress/sha_compress_stark.rs:402-606 (rotate_right.rs:29-62, not_operation.rs:23-34,
 * wrapping_add_5.rs:37-83, wrapping_add_2.rs:34-67, logic.rs:7-16).  Columns (sha_compress/columns.rs:9-55): state 0..31 (a..h, 4
 * little-endian bytes each), e_not 32, w_i 36, k_i 40, s_1_inter 44, s_1 48, e_and_f 52, e_not_and_g 56, ch 60, s_0_inter 64, s_0 68,
 * a_and_b 72, a_and_c 76, b_and_c 80, maj_inter 84, maj 88, rotations (value[4], shift, carry) e_rr_6 92, e_rr_11 98, e_rr_25 104,
 * a_rr_2 110, a_rr_13 116, a_rr_22 122, temp2 (value[4], carry[2]) 128, d_add_temp1 134, temp1_add_temp2 140, timestamp 146,
 * segment 147, context 148, w_i_virt 149, temp1 (value[4], carry[5]) 150, round 159..223 (NUM_COMPRESS_ROWS = 65) ---- */
/* wrapping add of `nin` byte quadruples: op = value[4] then carry[ncarry]; every constraint is multiplied by `gate` */
static void TNAME(sha_wadd)(const T* const* in, int nin, const T* op, int ncarry, T gate, TNAME(consumer) * k) {
    T one = T_FROMB(1), csum = T_FROMB(0), carry = T_FROMB(0);
    for (int i = 0; i < ncarry; i++) {
        TNAME(cons)(k, T_MUL(gate, T_MUL(op[4 + i], T_SUB(one, op[4 + i]))));
        csum = T_ADD(csum, op[4 + i]);
        if (i) carry = T_ADD(carry, T_MULB(op[4 + i], (gl_t)i));
    }
    TNAME(cons)(k, T_MUL(gate, T_SUB(csum, one)));
    T sum = T_FROMB(0);
    for (int b = 3; b >= 0; b--) {
        T s = in[0][b];
        for (int q = 1; q < nin; q++) s = T_ADD(s, in[q][b]);
        sum = T_ADD(T_MULB(sum, 1u << 8), s);
    }
    TNAME(cons)(k, T_MUL(gate, T_SUB(T_SUB(sum, T_MULB(carry, (gl_t)1 << 32)), TNAME(le4)(op))));
}
static void TNAME(eval_sha_compress)(const T* lv, const T* nv, TNAME(consumer) * k) {
    T one = T_FROMB(1);
    T is_final = lv[159 + 64];
    TNAME(cons)(k, T_MUL(is_final, T_SUB(is_final, one)));
    T not_final = T_SUB(one, is_final);
    T sum = T_FROMB(0);
    for (int i = 0; i < 65; i++) sum = T_ADD(sum, lv[159 + i]);
    TNAME(cons)(k, T_MUL(sum, T_SUB(sum, one)));
    T g = T_MUL(sum, not_final);
    for (int i = 0; i < 4; i++) {
        T kb = T_FROMB(0);
        for (int j = 0; j < 64; j++) kb = T_ADD(kb, T_MULB(lv[159 + j], (ZKO_SHA256_K[j] >> (8 * i)) & 0xFF));
        TNAME(cons)(k, T_MUL(g, T_SUB(lv[40 + i], kb)));
    }
    TNAME(sha_rot)(lv + 16, lv + 92, 6, 0, k);
    TNAME(sha_rot)(lv + 16, lv + 98, 11, 0, k);
    TNAME(sha_rot)(lv + 16, lv + 104, 25, 0, k);
    TNAME(sha_rot)(lv + 0, lv + 110, 2, 0, k);
    TNAME(sha_rot)(lv + 0, lv + 116, 13, 0, k);
    TNAME(sha_rot)(lv + 0, lv + 122, 22, 0, k);
    for (int i = 0; i < 4; i++) TNAME(cons)(k, T_MUL(sum, T_SUB(T_ADD(lv[16 + i], lv[32 + i]), T_FROMB(255))));
    { const T* in[5] = {lv + 28, lv + 48, lv + 60, lv + 40, lv + 36}; TNAME(sha_wadd)(in, 5, lv + 150, 5, sum, k); }   /* temp1 = h + s_1 + ch + k_i + w_i */
    { const T* in[2] = {lv + 68, lv + 88}; TNAME(sha_wadd)(in, 2, lv + 128, 2, sum, k); }                               /* temp2 = s_0 + maj */
    { const T* in[2] = {lv + 12, lv + 150}; TNAME(sha_wadd)(in, 2, lv + 134, 2, sum, k); }                              /* d + temp1 */
    { const T* in[2] = {lv + 150, lv + 128}; TNAME(sha_wadd)(in, 2, lv + 140, 2, sum, k); }                             /* temp1 + temp2 */
    TNAME(cons)(k, T_MUL(g, T_SUB(nv[146], lv[146])));
    TNAME(cons)(k, T_MUL(g, T_SUB(T_SUB(nv[149], lv[149]), T_FROMB(4))));
    for (int i = 0; i < 4; i++) TNAME(cons)(k, T_MUL(g, T_SUB(lv[140 + i], nv[0 + i])));       /* temp1 + temp2 = next a */
    for (int w = 0; w < 3; w++)                                                                  /* a, b, c -> next b, c, d */
        for (int i = 0; i < 4; i++) TNAME(cons)(k, T_MUL(g, T_SUB(lv[4 * w + i], nv[4 * (w + 1) + i])));
    for (int i = 0; i < 4; i++) TNAME(cons)(k, T_MUL(g, T_SUB(lv[134 + i], nv[16 + i])));      /* d + temp1 = next e */
    for (int w = 4; w < 7; w++)                                                                  /* e, f, g -> next f, g, h */
        for (int i = 0; i < 4; i++) TNAME(cons)(k, T_MUL(g, T_SUB(lv[4 * w + i], nv[4 * (w + 1) + i])));
}

/* ---- ShaCompressSpongeStark constraints: sha_compress_sponge/sha_compress_sponge_stark.rs:233-268.  Columns
 * (sha_compress_sponge/columns.rs:7-27): hx 0..31, output_state 32..63, output_hx[8] x (value[4], carry[2]) 64..111, hx_virt 112..119,
 * w_start_virt 120, timestamp 121, context 122, segment 123, w_start_segment 124, w_start_context 125, is_real_round 126 ---- */
static void TNAME(eval_sha_compress_sponge)(const T* lv, TNAME(consumer) * k) {
    T one = T_FROMB(1), real = lv[126];
    TNAME(cons)(k, T_MUL(real, T_SUB(real, one)));
    for (int i = 0; i < 7; i++) TNAME(cons)(k, T_MUL(real, T_SUB(T_SUB(lv[113 + i], lv[112 + i]), T_FROMB(4))));
    for (int i = 0; i < 8; i++) {
        const T* in[2] = {lv + 4 * i, lv + 32 + 4 * i};
        TNAME(sha_wadd)(in, 2, lv + 64 + 6 * i, 2, real, k);
    }
}

/* ---- ArithmeticStark constraints: arithmetic/arithmetic_stark.rs:214-240 -> mul.rs:109-190, mult.rs:115-262, addcy.rs:41-160,
 * slt.rs:50-120, lui.rs:31-49, div.rs:264-560, shift.rs:52-96, sra.rs:66-133, lo_hi.rs:23-36.  N_LIMBS = 2 (16-bit limbs of 32-bit
 * registers).  Columns (arithmetic/columns.rs): 26 operation flags 0..25; shared columns 26..43 (INPUT_REGISTER_0 26, _1 28, _2 30,
 * OUTPUT_REGISTER 32 (= OUTPUT_REGISTER_LO), AUX_INPUT_REGISTER_0 34 (= OUTPUT_REGISTER_HI), _1 36, _2 38, QUOT_ABS 40, REM_ABS 42;
 * MULT_AUX_LO 36..39, MULT_AUX_HI 40..43); RANGE_COUNTER 44, RC_FREQUENCIES 45, AUX_EXTRA 46..53.  Second-row (nv) registers of the
 * two-row operations: MODULAR_OUT_AUX_RED 26, MODULAR_MOD_IS_ZERO 28, MODULAR_AUX_INPUT_LO 29..31, _HI 32..34,
 * MODULAR_DIV_DENOM_IS_ZERO 35. ---- */
#ifndef ZKO_ARITH_COLS
#define ZKO_ARITH_COLS
enum { AR_ADD = 0, AR_ADDU, AR_ADDI, AR_ADDIU, AR_SUB, AR_SUBU, AR_MULT, AR_MULTU, AR_MUL, AR_DIV, AR_DIVU, AR_SLLV, AR_SRLV, AR_SRAV, AR_SLL, AR_SRL,
       AR_SRA, AR_SLT, AR_SLTU, AR_SLTI, AR_SLTIU, AR_LUI, AR_MFHI, AR_MTHI, AR_MFLO, AR_MTLO,
       AR_IN0 = 26, AR_IN1 = 28, AR_IN2 = 30, AR_OUT = 32, AR_AUX0 = 34, AR_AUX1 = 36, AR_AUX2 = 38, AR_QUOT_ABS = 40, AR_REM_ABS = 42,
       AR_MULT_AUX_LO = 36, AR_MULT_AUX_HI = 40, AR_RANGE_COUNTER = 44, AR_RC_FREQ = 45, AR_AUX_EXTRA = 46,
       AR_NV_RED = 26, AR_NV_MOD_IS_ZERO = 28, AR_NV_AUX_LO = 29, AR_NV_AUX_HI = 32, AR_NV_DENOM_IS_ZERO = 35 };
#define ZKO_INV_65536 18446462594437939201ULL /* addcy.rs:41 GOLDILOCKS_INVERSE_65536 */
#endif
/* eval_packed_generic_addcy addcy.rs:43-93: x + y == z + cy 2^32 limb by limb */
static void TNAME(ar_addcy)(TNAME(consumer) * k, T filter, const T* x, const T* y, const T* z, const T* given_cy, int two_row) {
    T cy = T_FROMB(0), ovf = T_FROMB(65536);
    for (int i = 0; i < 2; i++) {
        T t = T_SUB(T_ADD(T_ADD(cy, x[i]), y[i]), z[i]);
        T c = T_MUL(T_MUL(filter, t), T_SUB(ovf, t));
        if (two_row) TNAME(cons_transition)(k, c); else TNAME(cons)(k, c);
        cy = T_MULB(t, ZKO_INV_65536);
    }
    if (two_row) {
        TNAME(cons_transition)(k, T_MUL(filter, T_SUB(cy, given_cy[0])));
        TNAME(cons_transition)(k, T_MUL(filter, given_cy[1]));
    } else {
        TNAME(cons)(k, T_MUL(T_MUL(filter, given_cy[0]), T_SUB(given_cy[0], T_FROMB(1))));
        TNAME(cons)(k, T_MUL(filter, T_SUB(cy, given_cy[0])));
        TNAME(cons)(k, T_MUL(filter, given_cy[1]));
    }
}
/* eval_packed_generic_mul mul.rs:109-135: left * right == out + (x - 2^16) aux(x) at x = 2^16, low two limbs */
static void TNAME(ar_mul)(const T* lv, T filter, const T* l, const T* r, TNAME(consumer) * k) {
    T aux[2], cp[2];
    for (int i = 0; i < 2; i++) aux[i] = T_SUB(T_ADD(lv[AR_AUX0 + i], T_MULB(lv[AR_AUX1 + i], 65536)), T_FROMB(1 << 20));
    cp[0] = T_SUB(T_MUL(l[0], r[0]), lv[AR_OUT]);
    cp[1] = T_SUB(T_ADD(T_MUL(l[0], r[1]), T_MUL(l[1], r[0])), lv[AR_OUT + 1]);
    cp[0] = T_ADD(cp[0], T_MULB(aux[0], 65536));                       /* minus pol_adjoin_root: res[0] = -base aux[0] */
    cp[1] = T_SUB(cp[1], T_SUB(aux[0], T_MULB(aux[1], 65536)));        /* res[1] = aux[0] - base aux[1] */
    for (int i = 0; i < 2; i++) TNAME(cons)(k, T_MUL(filter, cp[i]));
}
/* eval_packed_generic_mult_helper mult.rs:236-262 on 4-limb operands */
static void TNAME(ar_mult_helper)(const T* lv, T filter, const T* l, const T* r, TNAME(consumer) * k) {
    T aux[4], cp[4];
    for (int i = 0; i < 4; i++) aux[i] = T_SUB(T_ADD(lv[AR_MULT_AUX_LO + i], T_MULB(lv[AR_MULT_AUX_HI + i], 65536)), T_FROMB(1 << 20));
    for (int d = 0; d < 4; d++) {
        T s = T_FROMB(0);
        for (int i = 0; i <= d; i++) s = T_ADD(s, T_MUL(l[i], r[d - i]));
        cp[d] = T_SUB(s, lv[AR_OUT + d]);
    }
    cp[0] = T_ADD(cp[0], T_MULB(aux[0], 65536));
    for (int d = 1; d < 4; d++) cp[d] = T_SUB(cp[d], T_SUB(aux[d - 1], T_MULB(aux[d], 65536)));
    for (int d = 0; d < 4; d++) TNAME(cons)(k, T_MUL(filter, cp[d]));
}
/* eval_packed_div_helper div.rs:509-541 with modular_constr_poly :325-380 and check_reduced :300-323 */
static void TNAME(ar_div_helper)(const T* lv, const T* nv, TNAME(consumer) * k, T filter, int num, int den, int quo, int rem) {
    T one = T_FROMB(1);
    TNAME(cons_last)(k, filter);
    T miz = nv[AR_NV_MOD_IS_ZERO];
    TNAME(cons_transition)(k, T_MUL(filter, T_SUB(T_MUL(miz, miz), miz)));
    T modulus[2] = {lv[den], lv[den + 1]}, output[2] = {lv[rem], lv[rem + 1]};
    TNAME(cons_transition)(k, T_MUL(T_MUL(filter, T_ADD(modulus[0], modulus[1])), miz));
    modulus[0] = T_ADD(modulus[0], miz);
    T ddz = nv[AR_NV_DENOM_IS_ZERO];
    T shr_div = T_ADD(T_ADD(T_ADD(lv[AR_DIV], lv[AR_DIVU]), T_ADD(lv[AR_SRL], lv[AR_SRLV])), T_ADD(lv[AR_SRA], lv[AR_SRAV]));
    TNAME(cons_transition)(k, T_MUL(filter, T_SUB(T_MUL(miz, shr_div), ddz)));
    output[0] = T_ADD(output[0], ddz);
    {
        T less[2] = {T_SUB(one, T_MUL(miz, shr_div)), T_FROMB(0)};
        TNAME(ar_addcy)(k, filter, modulus, nv + AR_NV_RED, output, less, 1);
    }
    output[0] = T_SUB(output[0], ddz);
    T q[4] = {lv[quo], lv[quo + 1], T_FROMB(0), T_FROMB(0)}, prod[5];
    for (int d = 0; d < 5; d++) prod[d] = T_FROMB(0);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++) prod[i + j] = T_ADD(prod[i + j], T_MUL(q[i], modulus[j]));
    TNAME(cons_transition)(k, T_MUL(filter, prod[4]));
    T cp[4] = {T_ADD(prod[0], output[0]), T_ADD(prod[1], output[1]), prod[2], prod[3]}, aux[4];
    for (int i = 0; i < 3; i++) aux[i] = T_ADD(T_SUB(nv[AR_NV_AUX_LO + i], T_FROMB(1 << 20)), T_MULB(nv[AR_NV_AUX_HI + i], 65536));
    aux[3] = T_FROMB(0);
    cp[0] = T_SUB(cp[0], T_MULB(aux[0], 65536));                       /* plus pol_adjoin_root(aux, base) */
    for (int d = 1; d < 4; d++) cp[d] = T_ADD(cp[d], T_SUB(aux[d - 1], T_MULB(aux[d], 65536)));
    cp[0] = T_SUB(cp[0], lv[num]);
    cp[1] = T_SUB(cp[1], lv[num + 1]);
    for (int d = 0; d < 4; d++) TNAME(cons_transition)(k, T_MUL(filter, cp[d]));
}
/* check_abs of eval_packed_div div.rs:400-430; returns is_neg */
static T TNAME(ar_check_abs)(const T* lv, const T* nv, TNAME(consumer) * k, T filter, int input, int abs_col, int sum_col, int neg_col, int borrow_col) {
    T one = T_FROMB(1), ovf = T_FROMB(65536);
    T is_neg = nv[neg_col];
    TNAME(cons_transition)(k, T_MUL(T_MUL(filter, is_neg), T_SUB(one, is_neg)));
    TNAME(cons_transition)(k, T_MUL(filter, T_SUB(T_SUB(T_ADD(lv[input + 1], T_FROMB(32768)), nv[sum_col]), T_MUL(is_neg, ovf))));
    T b = nv[borrow_col];
    TNAME(cons_transition)(k, T_MUL(T_MUL(filter, b), T_SUB(one, b)));
    T neg_in[2] = {T_SUB(T_MUL(b, ovf), lv[input]), T_SUB(T_SUB(ovf, lv[input + 1]), b)};
    for (int i = 0; i < 2; i++)
        TNAME(cons_transition)(k, T_MUL(filter, T_SUB(T_ADD(T_MUL(is_neg, neg_in[i]), T_MUL(T_SUB(one, is_neg), lv[input + i])), lv[abs_col + i])));
    return is_neg;
}
static void TNAME(eval_arithmetic)(const T* lv, const T* nv, TNAME(consumer) * k) {
    T one = T_FROMB(1), ovf = T_FROMB(65536);
    /* range counter arithmetic_stark.rs:224-230 */
    T rc1 = lv[AR_RANGE_COUNTER], incr = T_SUB(nv[AR_RANGE_COUNTER], rc1);
    TNAME(cons_first)(k, rc1);
    TNAME(cons_transition)(k, T_SUB(T_MUL(incr, incr), incr));
    TNAME(cons_last)(k, T_SUB(rc1, T_FROMB(65535)));
    /* mul */
    TNAME(ar_mul)(lv, lv[AR_MUL], lv + AR_IN0, lv + AR_IN1, k);
    /* mult (signed, then unsigned) mult.rs:115-234 */
    {
        T f = lv[AR_MULT], l[4], r[4];
        for (int s = 0; s < 2; s++) {
            T is_neg = lv[AR_AUX_EXTRA + s], in_hi = lv[(s ? AR_IN1 : AR_IN0) + 1], sum = lv[AR_IN2 + s];
            TNAME(cons)(k, T_MUL(T_MUL(f, is_neg), T_SUB(one, is_neg)));
            TNAME(cons)(k, T_MUL(f, T_SUB(T_SUB(T_ADD(in_hi, T_FROMB(32768)), sum), T_MUL(is_neg, ovf))));
            T* dst = s ? r : l;
            dst[0] = lv[(s ? AR_IN1 : AR_IN0)]; dst[1] = in_hi; dst[2] = dst[3] = T_MULB(is_neg, 65535);
        }
        TNAME(ar_mult_helper)(lv, f, l, r, k);
        T lu[4] = {lv[AR_IN0], lv[AR_IN0 + 1], T_FROMB(0), T_FROMB(0)}, ru[4] = {lv[AR_IN1], lv[AR_IN1 + 1], T_FROMB(0), T_FROMB(0)};
        TNAME(ar_mult_helper)(lv, lv[AR_MULTU], lu, ru, k);
    }
    /* addcy addcy.rs:142-160 (ADDU and SUBU rows are not constrained there) */
    TNAME(ar_addcy)(k, lv[AR_ADD], lv + AR_IN0, lv + AR_IN1, lv + AR_OUT, lv + AR_AUX0, 0);
    TNAME(ar_addcy)(k, lv[AR_SUB], lv + AR_IN1, lv + AR_OUT, lv + AR_IN0, lv + AR_AUX0, 0);
    TNAME(ar_addcy)(k, lv[AR_ADDI], lv + AR_IN0, lv + AR_IN1, lv + AR_OUT, lv + AR_AUX0, 0);
    TNAME(ar_addcy)(k, lv[AR_ADDIU], lv + AR_IN0, lv + AR_IN1, lv + AR_OUT, lv + AR_AUX0, 0);
    /* slt slt.rs:50-120: x = in1, y = aux (diff), z = in0, given_cy = AUX_INPUT_REGISTER_1, rd = OUTPUT_REGISTER */
    {
        T f = T_ADD(T_ADD(lv[AR_SLT], lv[AR_SLTU]), T_ADD(lv[AR_SLTI], lv[AR_SLTIU])), sign = T_ADD(lv[AR_SLT], lv[AR_SLTI]);
        const T *x = lv + AR_IN1, *y = lv + AR_AUX0, *z = lv + AR_IN0, *gc = lv + AR_AUX1, *rd = lv + AR_OUT;
        T cy = T_FROMB(0);
        for (int i = 0; i < 2; i++) {
            T t = T_SUB(T_ADD(T_ADD(cy, x[i]), y[i]), z[i]);
            TNAME(cons)(k, T_MUL(T_MUL(f, t), T_SUB(ovf, t)));
            cy = T_MULB(t, ZKO_INV_65536);
        }
        TNAME(cons)(k, T_MUL(T_MUL(f, gc[0]), T_SUB(gc[0], one)));
        TNAME(cons)(k, T_MUL(T_MUL(f, T_SUB(cy, gc[0])), T_SUB(one, sign)));
        TNAME(cons)(k, T_MUL(T_MUL(f, gc[1]), T_SUB(T_SUB(one, cy), gc[0])));
        TNAME(cons_transition)(k, T_MUL(f, T_SUB(rd[0], gc[0])));
        TNAME(cons)(k, T_MUL(T_MUL(f, gc[1]), T_SUB(one, sign)));
        TNAME(cons_transition)(k, T_MUL(f, rd[1]));
    }
    /* lui */
    TNAME(ar_mul)(lv, lv[AR_LUI], lv + AR_IN0, lv + AR_IN1, k);
    /* div: unsigned, then signed div.rs:382-507 */
    TNAME(ar_div_helper)(lv, nv, k, lv[AR_DIVU], AR_IN0, AR_IN1, AR_OUT, AR_AUX0);
    {
        T f = lv[AR_DIV];
        T n0 = TNAME(ar_check_abs)(lv, nv, k, f, AR_IN0, AR_IN2, AR_NV_DENOM_IS_ZERO + 1, AR_NV_DENOM_IS_ZERO + 5, AR_NV_DENOM_IS_ZERO + 6);
        T n1 = TNAME(ar_check_abs)(lv, nv, k, f, AR_IN1, AR_AUX2, AR_NV_DENOM_IS_ZERO + 2, AR_NV_DENOM_IS_ZERO + 7, AR_NV_DENOM_IS_ZERO + 8);
        T nq = TNAME(ar_check_abs)(lv, nv, k, f, AR_OUT, AR_QUOT_ABS, AR_NV_DENOM_IS_ZERO + 3, AR_RC_FREQ + 1, AR_RC_FREQ + 2);
        T nr = TNAME(ar_check_abs)(lv, nv, k, f, AR_AUX0, AR_REM_ABS, AR_NV_DENOM_IS_ZERO + 4, AR_RC_FREQ + 3, AR_RC_FREQ + 4);
        T same = nv[AR_RC_FREQ + 5];
        TNAME(cons_transition)(k, T_MUL(f, T_SUB(T_SUB(T_ADD(n0, n1), T_MULB(T_MUL(n0, n1), 2)), same)));
        TNAME(cons_transition)(k, T_MUL(T_MUL(f, T_SUB(nq, same)), T_ADD(lv[AR_OUT], lv[AR_OUT + 1])));
        TNAME(cons_transition)(k, T_MUL(T_MUL(f, T_SUB(nr, n0)), T_ADD(lv[AR_AUX0], lv[AR_AUX0 + 1])));
        TNAME(ar_div_helper)(lv, nv, k, f, AR_IN2, AR_AUX2, AR_QUOT_ABS, AR_REM_ABS);
    }
    /* shift: sll = input * (1 << shift), srl = input / (1 << shift) shift.rs:52-96 */
    TNAME(ar_mul)(lv, T_ADD(lv[AR_SLL], lv[AR_SLLV]), lv + AR_IN1, lv + AR_IN2, k);
    TNAME(ar_div_helper)(lv, nv, k, T_ADD(lv[AR_SRL], lv[AR_SRLV]), AR_IN1, AR_IN2, AR_OUT, AR_AUX0);
    /* sra sra.rs:66-133 */
    {
        T f = T_ADD(lv[AR_SRA], lv[AR_SRAV]), shift = lv[AR_IN0];
        TNAME(cons_transition)(k, T_MUL(f, lv[AR_IN0 + 1]));
        T is_neg = lv[AR_AUX2 + 3];
        TNAME(cons_transition)(k, T_MUL(T_MUL(f, is_neg), T_SUB(one, is_neg)));
        TNAME(cons_transition)(k, T_MUL(f, T_SUB(T_SUB(T_ADD(lv[AR_IN1 + 1], T_FROMB(32768)), lv[AR_AUX2 + 2]), T_MUL(is_neg, ovf))));
        T shift_sq = nv[AR_AUX2 + 2];
        TNAME(cons_transition)(k, T_MUL(f, T_SUB(shift_sq, T_MUL(shift, shift))));
        T acc = T_FROMB(0);
        for (int i = 0; i < 16; i++) {                                  /* pairs of coefficients from the top: (c_{31-2i}, c_{30-2i}) */
            T w = i < 8 ? lv[AR_AUX_EXTRA + i] : nv[AR_AUX_EXTRA + i - 8];
            T v = T_ADD(T_ADD(T_MUL(acc, shift_sq), T_MULB(shift, ZKM_ARITH_SIGN_EXTEND_POLY[31 - 2 * i])), T_FROMB(ZKM_ARITH_SIGN_EXTEND_POLY[30 - 2 * i]));
            TNAME(cons_transition)(k, T_MUL(f, T_SUB(v, w)));
            acc = w;
        }
        T acc_lo = nv[AR_AUX2], acc_hi = nv[AR_AUX2 + 1];
        TNAME(cons_transition)(k, T_MUL(f, T_SUB(T_ADD(T_MUL(acc_hi, ovf), acc_lo), acc)));
        TNAME(ar_div_helper)(lv, nv, k, f, AR_IN1, AR_IN2, AR_AUX2, AR_AUX0);
        TNAME(cons_transition)(k, T_MUL(f, T_SUB(T_ADD(lv[AR_AUX2], T_MUL(acc_lo, is_neg)), lv[AR_OUT])));
        TNAME(cons_transition)(k, T_MUL(f, T_SUB(T_ADD(lv[AR_AUX2 + 1], T_MUL(acc_hi, is_neg)), lv[AR_OUT + 1])));
    }
    /* lo_hi */
    {
        T f = T_ADD(T_ADD(lv[AR_MFHI], lv[AR_MTHI]), T_ADD(lv[AR_MFLO], lv[AR_MTLO]));
        for (int i = 0; i < 2; i++) TNAME(cons)(k, T_MUL(f, T_SUB(lv[AR_IN0 + i], lv[AR_OUT + i])));
    }
}

/* ---- MemoryStark constraints: memory/memory_stark.rs:253-341 (columns memory/columns.rs: FILTER 0, TIMESTAMP 1, IS_READ 2,
 * ADDR_CONTEXT 3, ADDR_SEGMENT 4, ADDR_VIRTUAL 5, VALUE 6 (VALUE_LIMBS = 1), CONTEXT/SEGMENT/VIRTUAL_FIRST_CHANGE 7..9,
 * RANGE_CHECK 10, COUNTER 11, FREQUENCIES 12) ---- */
static void TNAME(eval_memory)(const T* lv, const T* nv, TNAME(consumer) * k) {
    T one = T_FROMB(1);
    T filter = lv[0];
    TNAME(cons)(k, T_MUL(filter, T_SUB(filter, one)));
    T cfc = lv[7], sfc = lv[8], vfc = lv[9];
    T unchanged = T_SUB(T_SUB(T_SUB(one, cfc), sfc), vfc);
    TNAME(cons)(k, T_MUL(cfc, T_SUB(one, cfc)));
    TNAME(cons)(k, T_MUL(sfc, T_SUB(one, sfc)));
    TNAME(cons)(k, T_MUL(vfc, T_SUB(one, vfc)));
    TNAME(cons)(k, T_MUL(unchanged, T_SUB(one, unchanged)));
    T dctx = T_SUB(nv[3], lv[3]), dseg = T_SUB(nv[4], lv[4]), dvirt = T_SUB(nv[5], lv[5]);
    TNAME(cons_transition)(k, T_MUL(sfc, dctx));
    TNAME(cons_transition)(k, T_MUL(vfc, dctx));
    TNAME(cons_transition)(k, T_MUL(vfc, dseg));
    TNAME(cons_transition)(k, T_MUL(unchanged, dctx));
    TNAME(cons_transition)(k, T_MUL(unchanged, dseg));
    TNAME(cons_transition)(k, T_MUL(unchanged, dvirt));
    T computed = T_ADD(T_ADD(T_MUL(cfc, T_SUB(dctx, one)), T_MUL(sfc, T_SUB(dseg, one))),
                       T_ADD(T_MUL(vfc, T_SUB(dvirt, one)), T_MUL(unchanged, T_SUB(nv[1], lv[1]))));
    TNAME(cons_transition)(k, T_SUB(lv[10], computed));
    TNAME(cons_transition)(k, T_MUL(T_MUL(nv[2], unchanged), T_SUB(nv[6], lv[6])));
}

/* ---- logUp range-check lookups of a table: Stark::lookups() (memory_stark.rs:476-483: RANGE_CHECK in COUNTER with
 * FREQUENCIES; every reference use is Column::single columns without filters) and eval_packed_lookups_generic
 * (lookup.rs:138-198).  lk_aux / lk_aux_next = the first num_lookup_columns auxiliary values. ---- */
#ifndef ZKO_LOOKUP_DEFS
#define ZKO_LOOKUP_DEFS
typedef struct { uint32_t ncols; const uint32_t* cols; uint32_t table_col, freq_col; } zko_lookup_def;
static const uint32_t MEMORY_LOOKUP_COLS[1] = {10};
static const zko_lookup_def MEMORY_LOOKUPS[1] = {{1, MEMORY_LOOKUP_COLS, 11, 12}};
/* ArithmeticStark::lookups() arithmetic_stark.rs:269-276: the 18 shared columns in RANGE_COUNTER with RC_FREQUENCIES */
static const uint32_t ARITH_LOOKUP_COLS[18] = {26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43};
static const zko_lookup_def ARITH_LOOKUPS[1] = {{18, ARITH_LOOKUP_COLS, 44, 45}};
static const zko_lookup_def* zko_table_lookups(int table_id, size_t* n) {
    if (table_id == 4) { *n = 1; return MEMORY_LOOKUPS; }
    if (table_id == 10) { *n = 1; return ARITH_LOOKUPS; }
    *n = 0;
    return NULL;
}
static size_t zko_table_num_lookup_columns(int table_id, size_t nchallenges) {
    size_t nl, total = 0;
    const zko_lookup_def* d = zko_table_lookups(table_id, &nl);
    for (size_t i = 0; i < nl; i++) total += ((d[i].ncols + 1) / 2 + 1) * nchallenges;
    return total;
}
#endif
static void TNAME(eval_lookups)(int table_id, const gl_t* challenges, size_t nch, const T* lv, const T* lk_aux, const T* lk_aux_next,
                                TNAME(consumer) * k) {
    size_t nl, start = 0;
    const zko_lookup_def* defs = zko_table_lookups(table_id, &nl);
    for (size_t l = 0; l < nl; l++) {
        const zko_lookup_def* d = &defs[l];
        size_t nh = (d->ncols + 1) / 2;
        for (size_t c = 0; c < nch; c++) {
            T ch = T_FROMB(challenges[c]);
            T hsum = T_FROMB(0);
            for (size_t q = 0; q < nh; q++) {
                T h = lk_aux[start + q];
                T combin0 = T_ADD(lv[d->cols[2 * q]], ch);
                if (2 * q + 1 < d->ncols) {
                    T combin1 = T_ADD(lv[d->cols[2 * q + 1]], ch);
                    TNAME(cons)(k, T_SUB(T_SUB(T_MUL(T_MUL(combin1, combin0), h), combin1), combin0));
                } else {
                    TNAME(cons)(k, T_SUB(T_MUL(combin0, h), T_FROMB(1)));
                }
                hsum = T_ADD(hsum, h);
            }
            T z = lk_aux[start + nh], next_z = lk_aux_next[start + nh];
            T table_ch = T_ADD(lv[d->table_col], ch);
            T y = T_SUB(T_MUL(hsum, table_ch), lv[d->freq_col]);
            TNAME(cons_first)(k, z);
            TNAME(cons)(k, T_SUB(T_MUL(T_SUB(next_z, z), table_ch), y));
            start += nh + 1;
        }
    }
}

/* table dispatch (Table ids of include/zkm_hip.h) */
/* ---- CpuStark constraints: cpu/cpu_stark.rs:260-285, in emission order: bootstrap_kernel.rs:308-353, decode.rs:66-100,
 * jumps.rs (jump/jumpi/jumpdirect, branch), membus.rs:35-48, memio.rs (load :175-437, store :738-961), shift.rs:18-124,
 * count.rs:10-75, syscall.rs:12-232, bits.rs:9-64, misc.rs (rdhwr, condmov, teq, ext, ror, ins, maddu).
 * Column map (cpu/columns/mod.rs:62-96, ops.rs:10-45, general.rs): 0 is_bootstrap_kernel, 1 is_exit_kernel, 2 context,
 * 3 code_context, 4 program_counter, 5 next_program_counter, 6 is_kernel_mode, 7..39 op flags, 40..49 branch view,
 * 50 opcode_bits[6], 56 rs_bits[5], 61 rt_bits[5], 66 rd_bits[5], 71 shamt_bits[5], 76 func_bits[6], 82..85 sponge flags,
 * 86..187 general (union), 188..203 memio, 204 clock, 205 + 6 i memory channel i (used, is_read, context, segment, virtual,
 * value), 9 channels. ---- */
#ifndef ZKO_CPU_ENUMS
#define ZKO_CPU_ENUMS
enum { CPU_OP = 7, CPU_BR = 40, CPU_OPC = 50, CPU_RS = 56, CPU_RT = 61, CPU_RD = 66, CPU_SHAMT = 71, CPU_FUNC = 76, CPU_GEN = 86,
       CPU_MEMIO = 188, CPU_CLOCK = 204, CPU_CH0 = 205 };
enum { OPF_BINARY = 7, OPF_BINARY_IMM, OPF_EQ_ISZERO, OPF_LOGIC, OPF_LOGIC_IMM, OPF_MOVZ, OPF_MOVN, OPF_CLZ, OPF_CLO, OPF_SHIFT,
       OPF_SHIFT_IMM, OPF_KECCAK_GENERAL, OPF_JUMPS, OPF_JUMPI, OPF_JUMPDIRECT, OPF_BRANCH, OPF_PC, OPF_GET_CONTEXT, OPF_SET_CONTEXT,
       OPF_EXIT_KERNEL, OPF_M_OP_LOAD, OPF_M_OP_STORE, OPF_NOP, OPF_EXT, OPF_INS, OPF_MADDU, OPF_RDHWR, OPF_SIGNEXT8, OPF_SIGNEXT16,
       OPF_SWAPHALF, OPF_TEQ, OPF_ROR, OPF_SYSCALL };
#endif
#define CPU_USED(i) lv[CPU_CH0 + 6 * (i)]
#define CPU_ISREAD(i) lv[CPU_CH0 + 6 * (i) + 1]
#define CPU_CTX(i) lv[CPU_CH0 + 6 * (i) + 2]
#define CPU_SEG(i) lv[CPU_CH0 + 6 * (i) + 3]
#define CPU_VIRT(i) lv[CPU_CH0 + 6 * (i) + 4]
#define CPU_VAL(i) lv[CPU_CH0 + 6 * (i) + 5]

/* util.rs:15-21 limb_from_bits_le */
static T TNAME(le_sum)(const T* bits, int n) {
    T acc = T_FROMB(0);
    for (int i = 0; i < n; i++) acc = T_ADD(acc, T_MULB(bits[i], (gl_t)1 << i));
    return acc;
}
static void TNAME(cpy)(T* dst, const T* src, int n) {
    for (int i = 0; i < n; i++) dst[i] = src[i];
}
static void TNAME(zero32)(T* a) {
    for (int i = 0; i < 32; i++) a[i] = T_FROMB(0);
}
/* memio.rs:40-48 */
static void TNAME(sext)(T* a, int n) {
    for (int i = n; i < 32; i++) a[i] = a[n - 1];
}
/* offset field insn[15:0] as 32 little-endian bits, sign-extended when `se` (memio.rs:17-25; jumps.rs:77-83, 296-302 use it
 * shifted left by two) */
static void TNAME(cpu_imm_bits)(const T* lv, T* out, int shift) {
    TNAME(zero32)(out);
    TNAME(cpy)(out + shift, lv + CPU_FUNC, 6);
    TNAME(cpy)(out + shift + 6, lv + CPU_SHAMT, 5);
    TNAME(cpy)(out + shift + 11, lv + CPU_RD, 5);
    for (int i = shift + 16; i < 32; i++) out[i] = lv[CPU_RD + 4];
}
/* memio.rs:65-77 */
static void TNAME(half_word)(TNAME(consumer) * k, T op, const T* rs, T mem, T v1, T v0) {
    T a = T_MUL(T_SUB(rs[1], T_FROMB(1)), T_SUB(mem, v0));
    T b = T_MUL(rs[1], T_SUB(mem, v1));
    TNAME(cons)(k, T_MUL(op, T_ADD(a, b)));
}
/* memio.rs:106-129; v00 .. v11 indexed by (rs[0], rs[1]) as in the reference's mem_val_{rs0}_{rs1} */
static void TNAME(byte_sel)(TNAME(consumer) * k, const T* lv, T op, const T* rs, T mem, T v00, T v10, T v01, T v11) {
    T prod = T_MUL(rs[0], rs[1]), aux = lv[CPU_GEN + 96];
    TNAME(cons)(k, T_MUL(op, T_SUB(prod, aux)));
    T one = T_FROMB(1);
    T sum = T_MUL(T_SUB(mem, v00), T_ADD(T_SUB(T_SUB(aux, rs[1]), rs[0]), one));
    sum = T_ADD(sum, T_MUL(T_SUB(mem, v10), T_SUB(aux, rs[0])));
    sum = T_ADD(sum, T_MUL(T_SUB(mem, v01), T_SUB(aux, rs[1])));
    sum = T_ADD(sum, T_MUL(T_SUB(mem, v11), aux));
    TNAME(cons)(k, T_MUL(sum, op));
}

static void TNAME(cpu_memio)(const T* lv, TNAME(consumer) * k, int store) {
    const T one = T_FROMB(1);
    T filter = T_MUL(lv[store ? OPF_M_OP_STORE : OPF_M_OP_LOAD], lv[CPU_OPC + 5]);
    T aux_filter = lv[CPU_MEMIO + 15];
    TNAME(cons)(k, T_MUL(filter, T_SUB(one, aux_filter)));
    TNAME(cons)(k, T_MUL(filter, T_SUB(CPU_SEG(0), T_FROMB(4))));
    TNAME(cons)(k, T_MUL(filter, T_SUB(CPU_SEG(1), T_FROMB(4))));
    T rs = CPU_VAL(0), rt = CPU_VAL(1), mem = CPU_VAL(3);
    const T *rsl = lv + CPU_GEN, *rtl = lv + CPU_GEN + 32, *ml = lv + CPU_GEN + 64;
    T off[32];
    TNAME(cpu_imm_bits)(lv, off, 0);
    T virt_raw = T_ADD(rs, TNAME(le_sum)(off, 32));
    T rs_from_bits = TNAME(le_sum)(rsl, 32);
    TNAME(cons)(k, T_MUL(T_MUL(aux_filter, T_SUB(rs_from_bits, virt_raw)), T_SUB(T_ADD(rs_from_bits, T_FROMB(1ULL << 32)), virt_raw)));
    TNAME(cons)(k, T_MUL(filter, T_SUB(TNAME(le_sum)(rtl, 32), rt)));
    T tmp[32];
    TNAME(cpy)(tmp, rsl, 32);
    tmp[0] = tmp[1] = T_FROMB(0);
    TNAME(cons)(k, T_MUL(filter, T_SUB(TNAME(le_sum)(tmp, 32), CPU_VIRT(2))));
    T a[32], b[32], c[32], d[32];
#define CPU_SUM4() TNAME(le_sum)(a, 32), TNAME(le_sum)(b, 32), TNAME(le_sum)(c, 32), TNAME(le_sum)(d, 32)
#define CPU_Z4() TNAME(zero32)(a), TNAME(zero32)(b), TNAME(zero32)(c), TNAME(zero32)(d)
    if (!store) {
        /* LH */
        TNAME(zero32)(a); TNAME(cpy)(a, ml, 16); TNAME(sext)(a, 16);          /* rs[1] == 1 */
        TNAME(zero32)(b); TNAME(cpy)(b, ml + 16, 16); TNAME(sext)(b, 16);     /* rs[1] == 0 */
        TNAME(half_word)(k, lv[CPU_MEMIO + 0], rsl, mem, TNAME(le_sum)(a, 32), TNAME(le_sum)(b, 32));
        /* LWL: a = 0_0, b = 1_0, c = 0_1, d = 1_1 */
        CPU_Z4();
        TNAME(cpy)(a, ml, 32);
        TNAME(cpy)(b, rtl, 8); TNAME(cpy)(b + 8, ml, 24);
        TNAME(cpy)(c, rtl, 16); TNAME(cpy)(c + 16, ml, 16);
        TNAME(cpy)(d, rtl, 24); TNAME(cpy)(d + 24, ml, 8);
        TNAME(byte_sel)(k, lv, lv[CPU_MEMIO + 1], rsl, mem, CPU_SUM4());
        /* LW */
        TNAME(cons)(k, T_MUL(lv[CPU_MEMIO + 2], T_SUB(mem, TNAME(le_sum)(ml, 32))));
        /* LBU */
        CPU_Z4();
        TNAME(cpy)(a, ml + 24, 8); TNAME(cpy)(b, ml + 16, 8); TNAME(cpy)(c, ml + 8, 8); TNAME(cpy)(d, ml, 8);
        TNAME(byte_sel)(k, lv, lv[CPU_MEMIO + 3], rsl, mem, CPU_SUM4());
        /* LHU */
        TNAME(zero32)(a); TNAME(cpy)(a, ml, 16);
        TNAME(zero32)(b); TNAME(cpy)(b, ml + 16, 16);
        TNAME(half_word)(k, lv[CPU_MEMIO + 4], rsl, mem, TNAME(le_sum)(a, 32), TNAME(le_sum)(b, 32));
        /* LWR */
        CPU_Z4();
        TNAME(cpy)(a + 8, rtl + 8, 24); TNAME(cpy)(a, ml + 24, 8);
        TNAME(cpy)(b + 16, rtl + 16, 16); TNAME(cpy)(b, ml + 16, 16);
        TNAME(cpy)(c + 24, rtl + 24, 8); TNAME(cpy)(c, ml + 8, 24);
        TNAME(cpy)(d, ml, 32);
        TNAME(byte_sel)(k, lv, lv[CPU_MEMIO + 5], rsl, mem, CPU_SUM4());
        /* LL */
        TNAME(cons)(k, T_MUL(lv[CPU_MEMIO + 11], T_SUB(mem, TNAME(le_sum)(ml, 32))));
        /* LB */
        CPU_Z4();
        TNAME(cpy)(a, ml + 24, 8); TNAME(cpy)(b, ml + 16, 8); TNAME(cpy)(c, ml + 8, 8); TNAME(cpy)(d, ml, 8);
        TNAME(sext)(a, 8); TNAME(sext)(b, 8); TNAME(sext)(c, 8); TNAME(sext)(d, 8);
        TNAME(byte_sel)(k, lv, lv[CPU_MEMIO + 14], rsl, mem, CPU_SUM4());
    } else {
        /* SB */
        CPU_Z4();
        TNAME(cpy)(a + 24, rtl, 8); TNAME(cpy)(a, ml, 24);
        TNAME(cpy)(b + 24, ml + 24, 8); TNAME(cpy)(b + 16, rtl, 8); TNAME(cpy)(b, ml, 16);
        TNAME(cpy)(c + 16, ml + 16, 16); TNAME(cpy)(c + 8, rtl, 8); TNAME(cpy)(c, ml, 8);
        TNAME(cpy)(d, rtl, 8); TNAME(cpy)(d + 8, ml + 8, 24);
        TNAME(byte_sel)(k, lv, lv[CPU_MEMIO + 6], rsl, mem, CPU_SUM4());
        /* SH: a = rs[1] == 1, b = rs[1] == 0 */
        TNAME(zero32)(a); TNAME(cpy)(a, rtl, 16); TNAME(cpy)(a + 16, ml + 16, 16);
        TNAME(zero32)(b); TNAME(cpy)(b + 16, rtl, 16); TNAME(cpy)(b, ml, 16);
        TNAME(half_word)(k, lv[CPU_MEMIO + 7], rsl, mem, TNAME(le_sum)(a, 32), TNAME(le_sum)(b, 32));
        /* SWL */
        CPU_Z4();
        TNAME(cpy)(a, rtl, 32);
        TNAME(cpy)(b, rtl + 8, 24); TNAME(cpy)(b + 24, ml + 24, 8);
        TNAME(cpy)(c, rtl + 16, 16); TNAME(cpy)(c + 16, ml + 16, 16);
        TNAME(cpy)(d, rtl + 24, 8); TNAME(cpy)(d + 8, ml + 8, 24);
        TNAME(byte_sel)(k, lv, lv[CPU_MEMIO + 8], rsl, mem, CPU_SUM4());
        /* SW */
        TNAME(cons)(k, T_MUL(lv[CPU_MEMIO + 9], T_SUB(mem, TNAME(le_sum)(rtl, 32))));
        /* SWR */
        CPU_Z4();
        TNAME(cpy)(a + 24, rtl, 8); TNAME(cpy)(a, ml, 24);
        TNAME(cpy)(b + 16, rtl, 16); TNAME(cpy)(b, ml, 16);
        TNAME(cpy)(c + 8, rtl, 24); TNAME(cpy)(c, ml, 8);
        TNAME(cpy)(d, rtl, 32);
        TNAME(byte_sel)(k, lv, lv[CPU_MEMIO + 10], rsl, mem, CPU_SUM4());
        /* SC, SDC1 */
        TNAME(cons)(k, T_MUL(lv[CPU_MEMIO + 12], T_SUB(mem, TNAME(le_sum)(rtl, 32))));
        TNAME(cons)(k, T_MUL(lv[CPU_MEMIO + 13], mem));
    }
    for (int ch = 6; ch < 8; ch++) TNAME(cons)(k, T_MUL(filter, CPU_USED(ch)));
#undef CPU_SUM4
#undef CPU_Z4
}

static void TNAME(eval_cpu)(const T* lv, const T* nv, TNAME(consumer) * k) {
    const T one = T_FROMB(1);
    const gl_t P32 = 1ULL << 32;
    /* -- bootstrap_kernel.rs:308-353 */
    T boot = lv[0], dboot = T_SUB(nv[0], lv[0]);
    TNAME(cons_first)(k, T_SUB(boot, one));
    TNAME(cons_last)(k, boot);
    TNAME(cons_transition)(k, T_MUL(dboot, T_ADD(dboot, one)));
    for (int i = 0; i < 9; i++) {
        T f = T_MUL(boot, CPU_USED(i));
        TNAME(cons)(k, T_MUL(f, CPU_CTX(i)));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_SEG(i), T_FROMB(0))));
    }
    for (int i = 0; i < 9; i++) TNAME(cons_transition)(k, T_MUL(dboot, CPU_USED(i)));
    /* -- decode.rs:66-100 */
    TNAME(cons)(k, T_MUL(lv[6], T_SUB(lv[6], one)));
    for (int i = 0; i < 6; i++) TNAME(cons)(k, T_MUL(lv[CPU_OPC + i], T_SUB(lv[CPU_OPC + i], one)));
    static const int flags[15] = {OPF_EQ_ISZERO, OPF_KECCAK_GENERAL, OPF_JUMPS, OPF_BRANCH, OPF_PC, OPF_GET_CONTEXT, OPF_SET_CONTEXT,
                                  OPF_EXIT_KERNEL, OPF_LOGIC, OPF_BINARY, OPF_BINARY_IMM, OPF_SHIFT, OPF_SHIFT_IMM, OPF_M_OP_LOAD,
                                  OPF_M_OP_STORE};
    T flag_sum = T_FROMB(0);
    for (int i = 0; i < 15; i++) {
        TNAME(cons)(k, T_MUL(lv[flags[i]], T_SUB(lv[flags[i]], one)));
        flag_sum = T_ADD(flag_sum, lv[flags[i]]);
    }
    TNAME(cons)(k, T_MUL(flag_sum, T_SUB(flag_sum, one)));
    /* -- jumps.rs eval_packed_jump_jumpi */
    {
        T is_jump = lv[OPF_JUMPS], is_jumpi = lv[OPF_JUMPI], is_jd = lv[OPF_JUMPDIRECT];
        T is_link = T_MUL(is_jump, lv[CPU_FUNC]), is_linki = T_MUL(is_jumpi, lv[CPU_OPC]);
        TNAME(cons)(k, T_MUL(is_jump, T_SUB(nv[5], CPU_VAL(0))));
        TNAME(cons)(k, T_MUL(is_jump, T_SUB(TNAME(le_sum)(lv + CPU_RS, 5), CPU_VIRT(0))));
        T imm[32];
        TNAME(zero32)(imm);
        TNAME(cpy)(imm + 2, lv + CPU_FUNC, 6);
        TNAME(cpy)(imm + 8, lv + CPU_SHAMT, 5);
        TNAME(cpy)(imm + 13, lv + CPU_RD, 5);
        TNAME(cpy)(imm + 18, lv + CPU_RT, 5);
        TNAME(cpy)(imm + 23, lv + CPU_RS, 5);
        T jump_dest = T_ADD(CPU_VAL(2), TNAME(le_sum)(imm, 28));
        TNAME(cons)(k, T_MUL(is_jumpi, T_SUB(nv[5], jump_dest)));
        T aux = CPU_VAL(2);
        TNAME(cpu_imm_bits)(lv, imm, 2);
        TNAME(cons)(k, T_MUL(is_jd, T_SUB(aux, TNAME(le_sum)(imm, 32))));
        T dst = T_ADD(T_ADD(lv[4], T_FROMB(4)), aux);
        TNAME(cons)(k, T_MUL(T_MUL(is_jd, T_SUB(nv[5], dst)), T_SUB(T_ADD(nv[5], T_FROMB(P32)), dst)));
        T links = T_ADD(T_ADD(is_link, is_linki), is_jd);
        TNAME(cons)(k, T_MUL(links, T_SUB(T_ADD(lv[4], T_FROMB(8)), CPU_VAL(1))));
        TNAME(cons)(k, T_MUL(is_link, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RD, 5))));
        TNAME(cons)(k, T_MUL(T_ADD(is_linki, is_jd), T_SUB(CPU_VIRT(1), T_FROMB(31))));
    }
    /* -- jumps.rs eval_packed_branch */
    {
        const T* br = lv + CPU_BR; /* should_jump, gt, lt, eq, is_gt, is_lt, is_eq, is_ge, is_le, is_ne */
        T filter = lv[OPF_BRANCH], sj = br[0];
        T is_gt = br[4], is_lt = br[5], is_eq = br[6], is_ge = br[7], is_le = br[8], is_ne = br[9];
        T norm = T_ADD(T_ADD(T_ADD(is_eq, is_ne), is_le), is_gt), special = T_ADD(is_ge, is_lt);
        T src1 = CPU_VAL(0), src2 = CPU_VAL(1), aux1 = CPU_VAL(2), aux2 = CPU_VAL(3), aux3 = CPU_VAL(4), aux4 = CPU_VAL(5);
        const gl_t inv32 = 18446744065119617026ULL; /* 2^-32, jumps.rs:15 */
        TNAME(cons)(k, T_MUL(sj, T_SUB(one, sj)));
        TNAME(cons)(k, T_MUL(sj, T_SUB(one, filter)));
        TNAME(cons)(k, T_MUL(filter, T_SUB(one, T_ADD(norm, special))));
        TNAME(cons)(k, T_MUL(filter, T_SUB(one, T_ADD(T_ADD(br[2], br[1]), br[3]))));
        T off[32];
        TNAME(cpu_imm_bits)(lv, off, 2);
        TNAME(cons)(k, T_MUL(filter, T_SUB(aux4, TNAME(le_sum)(off, 32))));
        T dst = T_ADD(T_ADD(lv[4], T_FROMB(4)), aux4);
        TNAME(cons)(k, T_MUL(T_MUL(sj, T_SUB(nv[5], dst)), T_SUB(T_ADD(nv[5], T_FROMB(P32)), dst)));
        TNAME(cons)(k, T_MUL(T_MUL(filter, T_SUB(one, sj)), T_SUB(nv[5], T_ADD(lv[4], T_FROMB(8)))));
        T d1 = T_SUB(T_ADD(aux1, src2), src1), d2 = T_SUB(T_ADD(aux2, src1), src2);
        TNAME(cons)(k, T_MUL(T_MUL(filter, d1), T_SUB(d1, T_FROMB(P32))));
        TNAME(cons)(k, T_MUL(T_MUL(filter, d2), T_SUB(d2, T_FROMB(P32))));
        TNAME(cons)(k, T_MUL(T_MUL(filter, aux1), T_SUB(T_ADD(aux1, aux2), T_FROMB(P32))));
        TNAME(cons)(k, T_MUL(T_MUL(filter, aux3), T_SUB(one, aux3)));
        TNAME(cons)(k, T_MUL(filter, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RS, 5))));
        TNAME(cons)(k, T_MUL(norm, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RT, 5))));
        TNAME(cons)(k, T_MUL(T_MUL(special, CPU_VIRT(1)), T_SUB(one, CPU_VIRT(1))));
        T ca = T_SUB(T_ADD(src2, aux1), src1);
        TNAME(cons)(k, T_MUL(T_MUL(filter, ca), T_SUB(T_FROMB(P32), ca)));
        T lt = T_MULB(ca, inv32);
        TNAME(cons)(k, T_MUL(br[2], T_SUB(one, lt)));
        T cb = T_SUB(T_ADD(src1, aux2), src2);
        TNAME(cons)(k, T_MUL(T_MUL(filter, cb), T_SUB(T_FROMB(P32), cb)));
        T gt = T_MULB(cb, inv32);
        TNAME(cons)(k, T_MUL(br[1], T_SUB(one, gt)));
        T ne = T_ADD(lt, gt);
        TNAME(cons)(k, T_MUL(br[3], ne));
        T na3 = T_SUB(one, aux3);
        T lt2 = T_ADD(T_MUL(br[2], na3), T_MUL(T_SUB(one, br[2]), aux3));
        T gt2 = T_ADD(T_MUL(br[1], na3), T_MUL(T_SUB(one, br[1]), aux3));
        T nf = T_SUB(one, filter);
        TNAME(cons)(k, T_MUL(is_eq, nf));
        TNAME(cons)(k, T_MUL(is_eq, T_SUB(sj, T_SUB(one, ne))));
        TNAME(cons)(k, T_MUL(is_ne, nf));
        TNAME(cons)(k, T_MUL(is_ne, T_SUB(sj, ne)));
        TNAME(cons)(k, T_MUL(is_le, nf));
        TNAME(cons)(k, T_MUL(is_le, T_SUB(sj, T_SUB(one, gt2))));
        TNAME(cons)(k, T_MUL(is_ge, nf));
        TNAME(cons)(k, T_MUL(is_ge, T_SUB(sj, T_SUB(one, lt2))));
        TNAME(cons)(k, T_MUL(is_gt, nf));
        TNAME(cons)(k, T_MUL(is_gt, T_SUB(sj, gt2)));
        TNAME(cons)(k, T_MUL(is_lt, nf));
        TNAME(cons)(k, T_MUL(is_lt, T_SUB(sj, lt2)));
    }
    /* -- membus.rs:35-48 */
    TNAME(cons)(k, T_SUB(lv[3], T_MUL(T_SUB(one, lv[6]), lv[2])));
    for (int i = 0; i < 9; i++) TNAME(cons)(k, T_MUL(CPU_USED(i), T_SUB(CPU_USED(i), one)));
    /* -- memio.rs */
    TNAME(cpu_memio)(lv, k, 0);
    TNAME(cpu_memio)(lv, k, 1);
    /* -- shift.rs:18-124 (variable, then immediate); two_exp = channel 3 */
    for (int imm = 0; imm < 2; imm++) {
        T is_shift = lv[imm ? OPF_SHIFT_IMM : OPF_SHIFT];
        T disp = imm ? TNAME(le_sum)(lv + CPU_SHAMT, 5) : CPU_VAL(0);
        TNAME(cons)(k, T_MUL(T_MUL(is_shift, CPU_USED(3)), T_SUB(CPU_ISREAD(3), one)));
        TNAME(cons)(k, T_MUL(is_shift, CPU_CTX(3)));
        TNAME(cons)(k, T_MUL(is_shift, T_SUB(CPU_SEG(3), T_FROMB(3))));
        TNAME(cons)(k, T_MUL(is_shift, T_SUB(CPU_VIRT(3), disp)));
    }
    /* -- count.rs:10-75 (CLZ / CLO) */
    {
        T fz = lv[OPF_CLZ], fo = lv[OPF_CLO], f = T_ADD(fo, fz);
        TNAME(cons)(k, T_MUL(f, T_SUB(TNAME(le_sum)(lv + CPU_OPC, 6), T_FROMB(0x1c))));
        T func = TNAME(le_sum)(lv + CPU_FUNC, 6);
        TNAME(cons)(k, T_MUL(fz, T_SUB(func, T_FROMB(0x20))));
        TNAME(cons)(k, T_MUL(fo, T_SUB(func, T_FROMB(0x21))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RS, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RD, 5))));
        const T *bits = lv + CPU_GEN, *eqs = lv + CPU_GEN + 32, *invs = lv + CPU_GEN + 64;
        for (int i = 0; i < 32; i++) TNAME(cons)(k, T_MUL(T_MUL(f, bits[i]), T_SUB(one, bits[i])));
        T sum = TNAME(le_sum)(bits, 32), rs = CPU_VAL(0), rd = CPU_VAL(1);
        TNAME(cons)(k, T_MUL(fz, T_SUB(rs, sum)));
        TNAME(cons)(k, T_MUL(fo, T_SUB(T_SUB(T_FROMB(0xffffffffULL), rs), sum)));
        TNAME(cons)(k, T_MUL(T_MUL(f, bits[31]), rd));
        int j = 0;
        for (int i = 30; i >= 0; i--) {
            T partial = TNAME(le_sum)(bits + i, 32 - i);
            T diff = T_SUB(partial, one);
            TNAME(cons)(k, T_MUL(T_MUL(f, diff), eqs[j]));
            TNAME(cons)(k, T_MUL(f, T_SUB(T_ADD(T_MUL(diff, invs[j]), eqs[j]), one)));
            TNAME(cons)(k, T_MUL(T_MUL(f, eqs[j]), T_SUB(rd, T_FROMB(31 - i))));
            j++;
            if (i == 0) {
                TNAME(cons)(k, T_MUL(T_MUL(f, partial), eqs[j]));
                TNAME(cons)(k, T_MUL(f, T_SUB(T_ADD(T_MUL(partial, invs[j]), eqs[j]), one)));
                TNAME(cons)(k, T_MUL(T_MUL(f, eqs[j]), T_SUB(rd, T_FROMB(32))));
            }
        }
    }
    /* -- syscall.rs:12-232 */
    {
        T f = lv[OPF_SYSCALL];
        T a0 = CPU_VAL(1), a1 = CPU_VAL(2), a2 = CPU_VAL(3), v0 = T_FROMB(0), v1 = T_FROMB(0);
        const T *cond = lv + CPU_GEN, *sysnum = lv + CPU_GEN + 12, *a0f = lv + CPU_GEN + 24;
        T sc_a1 = lv[CPU_GEN + 27];
        T result_v0 = CPU_VAL(4), result_v1 = CPU_VAL(5);
        T is_sysmap = sysnum[1], sz_nz = sc_a1, sz_zero = sysnum[10], sz = a1, sz_mid = sysnum[9];
        T a0_zero = a0f[0], a0_nz = a0f[2], heap0 = CPU_VAL(6), result_heap = CPU_VAL(7);
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[0], T_MUL(is_sysmap, a0_zero))));
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[1], T_MUL(cond[0], sz_nz))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[1]), T_SUB(T_ADD(heap0, sz_mid), result_heap)));
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[2], T_MUL(cond[0], sz_zero))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[2]), T_SUB(T_ADD(heap0, sz), result_heap)));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[0]), T_SUB(heap0, result_v0)));
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[3], T_MUL(is_sysmap, a0_nz))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[3]), T_SUB(a0, result_v0)));
        T is_brk = sysnum[2], brk_gt = cond[10], brk_le = cond[11], initial_brk = CPU_VAL(6);
        TNAME(cons)(k, T_MUL(T_MUL(f, is_brk), T_SUB(one, T_ADD(brk_gt, brk_le))));
        TNAME(cons)(k, T_MUL(T_MUL(f, brk_gt), T_SUB(a0, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, brk_le), T_SUB(initial_brk, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, is_brk), T_SUB(v1, result_v1)));
        T is_clone = sysnum[3];
        TNAME(cons)(k, T_MUL(T_MUL(f, is_clone), T_SUB(one, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, is_clone), T_SUB(v1, result_v1)));
        T is_read = sysnum[5], bad_v0 = T_FROMB(0xFFFFFFFFULL), bad_v1 = T_FROMB(9);
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[4], T_MUL(is_read, a0f[2]))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[4]), T_SUB(bad_v0, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[4]), T_SUB(bad_v1, result_v1)));
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[5], T_MUL(is_read, a0f[0]))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[5]), T_SUB(v0, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[5]), T_SUB(v1, result_v1)));
        T is_write = sysnum[6];
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[6], T_MUL(is_write, a0f[2]))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[6]), T_SUB(bad_v0, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[6]), T_SUB(bad_v1, result_v1)));
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[7], T_MUL(is_write, a0f[1]))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[7]), T_SUB(a2, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[7]), T_SUB(v1, result_v1)));
        T is_fcntl = sysnum[7];
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[8], T_MUL(is_fcntl, a0f[0]))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[8]), T_SUB(T_FROMB(0), result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[8]), T_SUB(v1, result_v1)));
        TNAME(cons)(k, T_MUL(f, T_SUB(cond[9], T_MUL(is_fcntl, a0f[1]))));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[9]), T_SUB(one, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, cond[9]), T_SUB(v1, result_v1)));
        T rest = T_SUB(T_SUB(is_fcntl, cond[8]), cond[9]);
        TNAME(cons)(k, T_MUL(f, T_SUB(rest, T_MUL(is_fcntl, a0f[2]))));
        TNAME(cons)(k, T_MUL(T_MUL(f, rest), T_SUB(bad_v0, result_v0)));
        TNAME(cons)(k, T_MUL(T_MUL(f, rest), T_SUB(bad_v1, result_v1)));
        TNAME(cons)(k, T_MUL(T_MUL(f, sysnum[8]), T_SUB(a0, CPU_VAL(6))));
    }
    /* -- bits.rs:9-64 (SEB, SEH, WSBH) */
    {
        T seh = lv[OPF_SIGNEXT16], seb = lv[OPF_SIGNEXT8], wsbh = lv[OPF_SWAPHALF];
        T f = T_ADD(T_ADD(seh, seb), wsbh);
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RT, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RD, 5))));
        const T* bits = lv + CPU_GEN + 32; /* io().rt_le */
        for (int i = 0; i < 32; i++) TNAME(cons)(k, T_MUL(T_MUL(f, bits[i]), T_SUB(one, bits[i])));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VAL(0), TNAME(le_sum)(bits, 32))));
        T rd = CPU_VAL(1), r[32];
        for (int i = 0; i < 32; i++) r[i] = bits[7];
        TNAME(cpy)(r, bits, 7);
        TNAME(cons)(k, T_MUL(seb, T_SUB(rd, TNAME(le_sum)(r, 32))));
        for (int i = 0; i < 32; i++) r[i] = bits[15];
        TNAME(cpy)(r, bits, 15);
        TNAME(cons)(k, T_MUL(seh, T_SUB(rd, TNAME(le_sum)(r, 32))));
        TNAME(cpy)(r, bits + 8, 8); TNAME(cpy)(r + 8, bits, 8); TNAME(cpy)(r + 16, bits + 24, 8); TNAME(cpy)(r + 24, bits + 16, 8);
        TNAME(cons)(k, T_MUL(wsbh, T_SUB(rd, TNAME(le_sum)(r, 32))));
    }
    /* -- misc.rs */
    const T *m_rs = lv + CPU_GEN, *m_msb = lv + CPU_GEN + 32, *m_lsb = lv + CPU_GEN + 64;
    T auxm = lv[CPU_GEN + 96], auxl = lv[CPU_GEN + 97], auxs = lv[CPU_GEN + 98];
    { /* rdhwr :10-45 */
        T f = lv[OPF_RDHWR], rd_index = lv[CPU_GEN + 99], eq0 = lv[CPU_GEN + 100], eq29 = lv[CPU_GEN + 101];
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RT, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(rd_index, TNAME(le_sum)(lv + CPU_RD, 5))));
        T rt_val = CPU_VAL(0), local_user = CPU_VAL(1);
        TNAME(cons)(k, T_MUL(T_MUL(f, eq0), rd_index));
        TNAME(cons)(k, T_MUL(T_MUL(f, eq0), T_SUB(rt_val, one)));
        TNAME(cons)(k, T_MUL(T_MUL(f, eq29), T_SUB(rd_index, T_FROMB(29))));
        TNAME(cons)(k, T_MUL(T_MUL(f, eq29), T_SUB(rt_val, local_user)));
        TNAME(cons)(k, T_MUL(T_MUL(f, T_SUB(T_SUB(one, eq29), eq0)), rt_val));
    }
    { /* condmov :108-146 */
        T rs = CPU_VAL(0), rt = CPU_VAL(1), rd = CPU_VAL(2), out = CPU_VAL(3), mov = CPU_VAL(4);
        T movn = lv[OPF_MOVN], movz = lv[OPF_MOVZ], f = T_ADD(movn, movz);
        T is_ne = T_MUL(lv[CPU_GEN], rt), is_eq = T_SUB(one, is_ne), no_mov = T_SUB(one, mov);
        TNAME(cons)(k, T_MUL(movn, T_SUB(mov, is_ne)));
        TNAME(cons)(k, T_MUL(movz, T_SUB(mov, is_eq)));
        TNAME(cons)(k, T_MUL(T_MUL(f, mov), no_mov));
        TNAME(cons)(k, T_MUL(f, T_SUB(out, T_ADD(T_MUL(mov, rs), T_MUL(no_mov, rd)))));
    }
    { /* teq :197-226 */
        T f = lv[OPF_TEQ];
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RT, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RS, 5))));
        T is_ne = T_MUL(T_SUB(CPU_VAL(0), CPU_VAL(1)), lv[CPU_GEN]);
        TNAME(cons)(k, T_MUL(f, T_SUB(one, is_ne)));
    }
    { /* ext :266-317 */
        T f = lv[OPF_EXT];
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RT, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RS, 5))));
        T msbd = TNAME(le_sum)(lv + CPU_RD, 5), lsb = TNAME(le_sum)(lv + CPU_SHAMT, 5), msb = T_ADD(lsb, msbd);
        TNAME(cons)(k, T_MUL(f, T_SUB(T_ADD(T_MUL(CPU_VAL(1), auxs), auxl), auxm)));
        for (int i = 0; i < 32; i++) {
            T mpartial = TNAME(le_sum)(m_rs, i + 1), lpartial = TNAME(le_sum)(m_rs, i);
            T fm = T_MUL(f, m_msb[i]), fl = T_MUL(f, m_lsb[i]);
            TNAME(cons)(k, T_MUL(fm, T_SUB(msb, T_FROMB(i))));
            TNAME(cons)(k, T_MUL(fm, T_SUB(auxm, mpartial)));
            TNAME(cons)(k, T_MUL(fl, T_SUB(lsb, T_FROMB(i))));
            TNAME(cons)(k, T_MUL(fl, T_SUB(auxl, lpartial)));
            TNAME(cons)(k, T_MUL(fl, T_SUB(auxs, T_FROMB((gl_t)1 << i))));
        }
    }
    { /* ror :563-604 */
        T f = lv[OPF_ROR];
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RD, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RT, 5))));
        T sa = TNAME(le_sum)(lv + CPU_SHAMT, 5), rd_result = CPU_VAL(1), r[32];
        for (int i = 0; i < 32; i++) {
            TNAME(cpy)(r, m_rs + i, 32 - i);
            TNAME(cpy)(r + 32 - i, m_rs, i);
            T fs = T_MUL(f, m_lsb[i]);
            TNAME(cons)(k, T_MUL(fs, T_SUB(sa, T_FROMB(i))));
            TNAME(cons)(k, T_MUL(fs, T_SUB(rd_result, TNAME(le_sum)(r, 32))));
        }
    }
    { /* ins :397-467 */
        T f = lv[OPF_INS];
        T rt_src = TNAME(le_sum)(lv + CPU_RT, 5);
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(1), rt_src)));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(2), rt_src)));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RS, 5))));
        T msb = TNAME(le_sum)(lv + CPU_RD, 5), lsb = TNAME(le_sum)(lv + CPU_SHAMT, 5);
        TNAME(cons)(k, T_MUL(f, T_SUB(T_SUB(CPU_VAL(2), auxm), T_MUL(auxl, auxs))));
        for (int i = 0; i < 32; i++) {
            T fm = T_MUL(f, m_msb[i]), fl = T_MUL(f, m_lsb[i]);
            TNAME(cons)(k, T_MUL(fl, T_SUB(lsb, T_FROMB(i))));
            TNAME(cons)(k, T_MUL(fl, T_SUB(auxs, T_FROMB((gl_t)1 << i))));
            TNAME(cons)(k, T_MUL(fm, T_SUB(T_SUB(msb, lsb), T_FROMB(i))));
            TNAME(cons)(k, T_MUL(fm, T_SUB(auxl, TNAME(le_sum)(m_rs, i + 1))));
        }
    }
    { /* maddu :659-726 */
        T f = lv[OPF_MADDU];
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(0), TNAME(le_sum)(lv + CPU_RS, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(1), TNAME(le_sum)(lv + CPU_RT, 5))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(2), T_FROMB(33))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(4), T_FROMB(33))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(3), T_FROMB(32))));
        TNAME(cons)(k, T_MUL(f, T_SUB(CPU_VIRT(5), T_FROMB(32))));
        T result = T_ADD(T_MULB(CPU_VAL(4), P32), CPU_VAL(5));
        T mul = T_MUL(CPU_VAL(0), CPU_VAL(1));
        T addend = T_ADD(T_MULB(CPU_VAL(2), P32), CPU_VAL(3));
        T carry = auxm, overflow = T_MULB(carry, P32);
        TNAME(cons)(k, T_MUL(T_MUL(f, carry), T_SUB(carry, T_FROMB(P32))));
        TNAME(cons)(k, T_MUL(f, T_SUB(T_SUB(T_ADD(mul, addend), overflow), result)));
    }
}
#undef CPU_USED
#undef CPU_ISREAD
#undef CPU_CTX
#undef CPU_SEG
#undef CPU_VIRT
#undef CPU_VAL

static int TNAME(table_width)(int table_id) { return table_id == 0 ? 262 : table_id == 1 ? 69 : table_id == 2 ? 470 : table_id == 3 ? 2431 : table_id == 4 ? 13 : table_id == 5 ? 110 : table_id == 6 ? 78 : table_id == 7 ? 76 : table_id == 8 ? 224 : table_id == 9 ? 127 : table_id == 10 ? 54 : table_id == 11 ? 259 : 0; }
static void TNAME(eval_table)(int table_id, const T* lv, const T* nv, TNAME(consumer) * k) {
    if (table_id == 0) TNAME(eval_poseidon)(lv, k);
    else if (table_id == 1) TNAME(eval_logic)(lv, k);
    else if (table_id == 2) TNAME(eval_keccak_sponge)(lv, nv, k);
    else if (table_id == 3) TNAME(eval_keccak)(lv, nv, k);
    else if (table_id == 4) TNAME(eval_memory)(lv, nv, k);
    else if (table_id == 5) TNAME(eval_poseidon_sponge)(lv, nv, k);
    else if (table_id == 6) TNAME(eval_sha_extend)(lv, k);
    else if (table_id == 7) TNAME(eval_sha_extend_sponge)(lv, nv, k);
    else if (table_id == 8) TNAME(eval_sha_compress)(lv, nv, k);
    else if (table_id == 9) TNAME(eval_sha_compress_sponge)(lv, k);
    else if (table_id == 10) TNAME(eval_arithmetic)(lv, nv, k);
    else TNAME(eval_cpu)(lv, nv, k);
}

/* ---- general CTL checks driven by the column-set description ----
 * Column::eval_with_next cross_table_lookup.rs:292-311, Filter::eval_filter :64-79,
 * GrandProductChallenge::combine :494-504 (reduce_with_powers(terms, beta) + gamma),
 * eval_helper_columns :1006-1058, eval_cross_table_lookup_checks :1067-1150.
 * The benchmark's fake CTL data (poseidon_stark.rs:786-799: helper columns, no column sets) is the
 * ncolsets == 0 case: eval_helper_columns emits nothing, then the last-row / transition checks on Z. */
static T TNAME(eval_column)(const zko_ctl_table* t, uint32_t ci, const T* lv, const T* nv) {
    const zko_column* c = &t->columns[ci];
    T acc = T_FROMB(0);
    for (uint32_t k = 0; k < c->n_local; k++) acc = T_ADD(acc, T_MULB(lv[t->term_col[c->term_off + k]], t->term_coeff[c->term_off + k]));
    for (uint32_t k = 0; k < c->n_next; k++) {
        uint32_t o = c->term_off + c->n_local + k;
        acc = T_ADD(acc, T_MULB(nv[t->term_col[o]], t->term_coeff[o]));
    }
    return T_ADD(acc, T_FROMB(c->constant));
}
static T TNAME(eval_filter)(const zko_ctl_table* t, const zko_colset* cs, const T* lv, const T* nv) {
    if (!cs->has_filter) return T_FROMB(1);
    T acc = T_FROMB(0);
    for (uint32_t k = 0; k < cs->nprod; k++)
        acc = T_ADD(acc, T_MUL(TNAME(eval_column)(t, t->filter_idx[cs->prod_off + 2 * k], lv, nv),
                               TNAME(eval_column)(t, t->filter_idx[cs->prod_off + 2 * k + 1], lv, nv)));
    for (uint32_t k = 0; k < cs->nconst; k++) acc = T_ADD(acc, TNAME(eval_column)(t, t->filter_idx[cs->const_off + k], lv, nv));
    return acc;
}
static T TNAME(combine)(const zko_ctl_table* t, const zko_colset* cs, const T* lv, const T* nv, gl_t beta, gl_t gamma) {
    T acc = T_FROMB(0);
    for (uint32_t k = cs->ncols; k-- > 0;) acc = T_ADD(T_MULB(acc, beta), TNAME(eval_column)(t, cs->col_off + k, lv, nv));
    return T_ADD(acc, T_FROMB(gamma));
}
static void TNAME(eval_ctl_general)(const zko_ctl_table* t, const zko_ctl_z* zs, const uint32_t* colset_ids, size_t nzs,
                                    const T* lv, const T* nv, const T* aux_local, const T* aux_next, TNAME(consumer) * k) {
    size_t total_helpers = 0, start = 0;
    for (size_t i = 0; i < nzs; i++) total_helpers += zs[i].num_helpers;
    for (size_t i = 0; i < nzs; i++) {
        const zko_ctl_z* z = &zs[i];
        const uint32_t* ids = colset_ids + z->colset_off;
        T local_z = aux_local[total_helpers + i], next_z = aux_next[total_helpers + i];
        if (z->num_helpers) {
            /* eval_helper_columns: chunks of constraint_degree - 1 = 2 column sets per helper */
            for (uint32_t j = 0; 2 * j < z->ncolsets; j++) {
                T h = aux_local[start + j];
                const zko_colset* c0 = &t->colsets[ids[2 * j]];
                T combin0 = TNAME(combine)(t, c0, lv, nv, z->beta, z->gamma), f0 = TNAME(eval_filter)(t, c0, lv, nv);
                if (2 * j + 1 < z->ncolsets) {
                    const zko_colset* c1 = &t->colsets[ids[2 * j + 1]];
                    T combin1 = TNAME(combine)(t, c1, lv, nv, z->beta, z->gamma), f1 = TNAME(eval_filter)(t, c1, lv, nv);
                    TNAME(cons)(k, T_SUB(T_SUB(T_MUL(T_MUL(combin1, combin0), h), T_MUL(f0, combin1)), T_MUL(f1, combin0)));
                } else {
                    TNAME(cons)(k, T_SUB(T_MUL(combin0, h), f0));
                }
            }
            T h_sum = T_FROMB(0);
            for (uint32_t h = 0; h < z->num_helpers; h++) h_sum = T_ADD(h_sum, aux_local[start + h]);
            TNAME(cons_last)(k, T_SUB(local_z, h_sum));
            TNAME(cons_transition)(k, T_SUB(T_SUB(local_z, next_z), h_sum));
        } else if (z->ncolsets > 1) {
            const zko_colset *c0 = &t->colsets[ids[0]], *c1 = &t->colsets[ids[1]];
            T combin0 = TNAME(combine)(t, c0, lv, nv, z->beta, z->gamma), combin1 = TNAME(combine)(t, c1, lv, nv, z->beta, z->gamma);
            T f0 = TNAME(eval_filter)(t, c0, lv, nv), f1 = TNAME(eval_filter)(t, c1, lv, nv);
            T cc = T_MUL(combin0, combin1), rhs = T_ADD(T_MUL(f0, combin1), T_MUL(f1, combin0));
            TNAME(cons_last)(k, T_SUB(T_MUL(cc, local_z), rhs));
            TNAME(cons_transition)(k, T_SUB(T_MUL(cc, T_SUB(local_z, next_z)), rhs));
        } else {
            const zko_colset* c0 = &t->colsets[ids[0]];
            T combin0 = TNAME(combine)(t, c0, lv, nv, z->beta, z->gamma), f0 = TNAME(eval_filter)(t, c0, lv, nv);
            TNAME(cons_last)(k, T_SUB(T_MUL(combin0, local_z), f0));
            TNAME(cons_transition)(k, T_SUB(T_MUL(combin0, T_SUB(local_z, next_z)), f0));
        }
        start += z->num_helpers;
    }
}
