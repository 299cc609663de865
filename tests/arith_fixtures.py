"""ArithmeticStark witness rows (test fixture): a Python restatement of the reference's row generators
(arithmetic/{mod,addcy,mul,mult,slt,lui,div,shift,sra,lo_hi}.rs `generate*`, arithmetic_stark.rs:138-199).  In the reference these
rows come from the CPU-side witness generator, so the product has no kernel for them; the GPU proves the table it is given."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from gen_arith_constants import interpolate, sign_extend_points  # noqa: E402

P = 0xFFFFFFFF00000001
NCOLS = 54
(IS_ADD, IS_ADDU, IS_ADDI, IS_ADDIU, IS_SUB, IS_SUBU, IS_MULT, IS_MULTU, IS_MUL, IS_DIV, IS_DIVU, IS_SLLV, IS_SRLV, IS_SRAV, IS_SLL, IS_SRL,
 IS_SRA, IS_SLT, IS_SLTU, IS_SLTI, IS_SLTIU, IS_LUI, IS_MFHI, IS_MTHI, IS_MFLO, IS_MTLO) = range(26)
IN0, IN1, IN2, OUT, AUX0, AUX1, AUX2 = 26, 28, 30, 32, 34, 36, 38
OUT_LO, OUT_HI, MULT_AUX_LO, MULT_AUX_HI, QUOT_ABS, REM_ABS = 32, 34, 36, 40, 40, 42
RANGE_COUNTER, RC_FREQ, AUX_EXTRA = 44, 45, 46
NV_OUT_AUX_RED, NV_MOD_IS_ZERO, NV_AUX_LO, NV_AUX_HI, NV_DENOM_IS_ZERO = 26, 28, 29, 32, 35
ABS_MAX = 1 << 20
SIGN_POLY = interpolate(sign_extend_points())


def limbs(x):
    return [x & 0xFFFF, (x >> 16) & 0xFFFF]


def s32(x):
    return x - (1 << 32) if x & 0x80000000 else x


def put(row, col, x):
    row[col], row[col + 1] = limbs(x)


def pol_mul_lo(a, b):
    n = len(a)
    return [sum(a[i] * b[d - i] for i in range(d + 1)) for d in range(n)]


def remove_root_2exp16(a):
    n = len(a)
    q = [0] * n
    q[0] = -(a[0] >> 16)
    for d in range(1, n - 1):
        q[d] = (q[d - 1] - a[d]) >> 16
    return q


def generate_mul(lv, left, right):                 # mul.rs:62-96
    n = len(left)
    un = pol_mul_lo(left, right)
    out, cy = [0] * n, 0
    for c in range(n):
        t = un[c] + cy
        cy = t >> 16
        out[c] = t & 0xFFFF
    lv[OUT:OUT + n] = out
    un = [u - o for u, o in zip(un, out)]
    aux = remove_root_2exp16(un)
    aux[n - 1] = -cy
    aux = [c + ABS_MAX for c in aux]
    for i, c in enumerate(aux):
        lv[AUX0 + i] = c & 0xFFFF
        lv[AUX1 + i] = (c >> 16) & 0xFFFF


def generate_mult_helper(lv, left, right):          # mult.rs:70-113
    un = pol_mul_lo(left, right)
    out, cy = [0] * 4, 0
    for c in range(4):
        t = un[c] + cy
        cy = t >> 16
        out[c] = t & 0xFFFF
    lv[OUT_LO:OUT_LO + 4] = out
    un = [u - o for u, o in zip(un, out)]
    aux = remove_root_2exp16(un)
    aux[3] = -cy
    for i, c in enumerate(aux):
        c += ABS_MAX
        lv[MULT_AUX_LO + i] = c & 0xFFFF
        lv[MULT_AUX_HI + i] = (c >> 16) & 0xFFFF


def generate_modular_op(lv, nv, filt, pol_input, mod_col):   # div.rs:182-262
    ml = [lv[mod_col], lv[mod_col + 1]]
    modulus = ml[0] + (ml[1] << 16)
    cp = list(pol_input) + [0]
    miz = 0
    if modulus == 0:
        if filt in (IS_DIV, IS_DIVU, IS_SRL, IS_SRLV):
            modulus = 1 << 32
        else:
            modulus, ml[0] = 1, 1
        miz = 1
    inp = sum(c << (16 * i) for i, c in enumerate(cp))
    out = inp % modulus
    out_l = limbs(out)
    quot = (inp - out) // modulus
    quot_l = [(quot >> (16 * i)) & 0xFFFF for i in range(4)]
    red = (1 << 32) - modulus + out
    cp[0] -= out_l[0]
    cp[1] -= out_l[1]
    prod = [0] * 5
    for i in range(4):
        for j in range(2):
            prod[i + j] += quot_l[i] * ml[j]
    assert prod[4] == 0
    cp = [c - p for c, p in zip(cp, prod[:4])]
    aux = [c + ABS_MAX for c in remove_root_2exp16(cp)]
    for i in range(3):
        nv[NV_AUX_LO + i] = aux[i] & 0xFFFF
        nv[NV_AUX_HI + i] = (aux[i] >> 16) & 0xFFFF
    nv[NV_MOD_IS_ZERO] = miz
    put(nv, NV_OUT_AUX_RED, red)
    nv[NV_DENOM_IS_ZERO] = miz * (lv[IS_DIV] + lv[IS_DIVU] + lv[IS_SRL] + lv[IS_SRLV])
    return out_l, quot_l


def generate_divu_helper(lv, nv, filt, in_col, mod_col, out_col, rem_col):   # div.rs:139-180
    out, quo = generate_modular_op(lv, nv, filt, [lv[in_col], lv[in_col + 1], 0], mod_col)
    assert quo[2] == quo[3] == 0 and [lv[out_col], lv[out_col + 1]] == quo[:2], "quotient mismatch"
    if rem_col is not None:
        assert [lv[rem_col], lv[rem_col + 1]] == out, "remainder mismatch"
    else:
        lv[AUX0:AUX0 + 2] = out


def rows_for(op, a, b):
    """(row, second row or None) of Operation::binary(op, a, b).to_rows() (mod.rs:150-313)."""
    lv, nv = [0] * NCOLS, [0] * NCOLS
    lv[op] = 1
    M = 0xFFFFFFFF
    if op in (IS_ADD, IS_ADDU, IS_ADDI, IS_ADDIU, IS_SUB, IS_SUBU):        # addcy.rs:12-39
        put(lv, IN0, a); put(lv, IN1, b)
        res = (a - b) if op in (IS_SUB, IS_SUBU) else (a + b)
        put(lv, AUX0, 1 if (res < 0 or res > M) else 0)
        put(lv, OUT, res & M)
        return lv, None
    if op == IS_MUL:                                                       # mul.rs:98-107
        put(lv, IN0, a); put(lv, IN1, b)
        generate_mul(lv, limbs(a), limbs(b))
        return lv, None
    if op in (IS_SLT, IS_SLTU, IS_SLTI, IS_SLTIU):                          # slt.rs:13-46
        put(lv, IN0, a); put(lv, IN1, b)
        cy = 1 if a < b else 0
        signed = op in (IS_SLT, IS_SLTI)
        rd = (1 if s32(a) < s32(b) else 0) if signed else cy
        cy_val = cy
        if signed and (a & 0x80000000) != (b & 0x80000000):
            cy_val = (1 << 16) | (1 - cy)
        put(lv, AUX0, (a - b) & M); put(lv, AUX1, cy_val); put(lv, OUT, rd)
        return lv, None
    if op in (IS_MULT, IS_MULTU):                                           # mult.rs:12-68
        put(lv, IN0, a); put(lv, IN1, b)
        if op == IS_MULT:
            n0, n1 = a >> 31, b >> 31
            lv[AUX_EXTRA], lv[AUX_EXTRA + 1] = n0, n1
            lv[IN2], lv[IN2 + 1] = (a >> 16) ^ 0x8000, (b >> 16) ^ 0x8000
            generate_mult_helper(lv, limbs(a) + [0xFFFF * n0] * 2, limbs(b) + [0xFFFF * n1] * 2)
        else:
            generate_mult_helper(lv, limbs(a) + [0, 0], limbs(b) + [0, 0])
        return lv, None
    if op in (IS_DIV, IS_DIVU):                                             # div.rs:21-137
        if op == IS_DIV:
            q = int(abs(s32(a)) // abs(s32(b))) * (1 if (s32(a) < 0) == (s32(b) < 0) else -1)
            quot, rem = q & M, (s32(a) - q * s32(b)) & M
        else:
            quot, rem = a // b, a % b
        put(lv, IN0, a); put(lv, IN1, b); put(lv, OUT_LO, quot); put(lv, OUT_HI, rem)
        if op == IS_DIVU:
            generate_divu_helper(lv, nv, op, IN0, IN1, OUT_LO, OUT_HI)
        else:
            def fill(x, abs_col, sum_col, neg_col, borrow_col):
                neg = x >> 31
                nv[neg_col], nv[sum_col], nv[borrow_col] = neg, (x >> 16) ^ 0x8000, 1 if x & 0xFFFF else 0
                put(lv, abs_col, abs(s32(x)))
                return neg
            n0 = fill(a, IN2, NV_DENOM_IS_ZERO + 1, NV_DENOM_IS_ZERO + 5, NV_DENOM_IS_ZERO + 6)
            n1 = fill(b, AUX2, NV_DENOM_IS_ZERO + 2, NV_DENOM_IS_ZERO + 7, NV_DENOM_IS_ZERO + 8)
            nv[RC_FREQ + 5] = n0 ^ n1
            fill(quot, QUOT_ABS, NV_DENOM_IS_ZERO + 3, RC_FREQ + 1, RC_FREQ + 2)
            fill(rem, REM_ABS, NV_DENOM_IS_ZERO + 4, RC_FREQ + 3, RC_FREQ + 4)
            generate_divu_helper(lv, nv, op, IN2, AUX2, QUOT_ABS, REM_ABS)
        return lv, nv
    if op == IS_LUI:                                                        # lui.rs:14-29  (a = imm)
        put(lv, IN0, a); put(lv, IN1, 1 << 16)
        generate_mul(lv, limbs(a), limbs(1 << 16))
        return lv, None
    if op in (IS_SLL, IS_SLLV, IS_SRL, IS_SRLV):                            # shift.rs:14-50  (a = value, b = shift amount)
        sh = b & 31
        put(lv, IN0, b); put(lv, IN1, a); put(lv, IN2, 1 << sh)
        if op in (IS_SLL, IS_SLLV):
            put(lv, OUT, (a << sh) & M)
            generate_mul(lv, limbs(a), limbs(1 << sh))
            return lv, None
        put(lv, OUT, a >> sh)
        generate_divu_helper(lv, nv, op, IN1, IN2, OUT, None)
        return lv, nv
    if op in (IS_SRA, IS_SRAV):                                             # sra.rs:18-64  (b = shift amount < 32)
        sh = b
        put(lv, IN0, sh); put(lv, IN1, a); put(lv, OUT, (s32(a) >> sh) & M); put(lv, IN2, 1 << (sh & 31))
        put(lv, AUX2, a >> sh)
        lv[AUX2 + 2], lv[AUX2 + 3] = (a >> 16) ^ 0x8000, a >> 31
        acc, w = 0, []
        for i in range(15, -1, -1):                                          # eval_poly sra.rs:284-300: pairs from the top
            acc = (SIGN_POLY[2 * i] + SIGN_POLY[2 * i + 1] * sh + acc * sh * sh) % P
            w.append(acc)
        lv[AUX_EXTRA:AUX_EXTRA + 8] = w[:8]
        nv[AUX_EXTRA:AUX_EXTRA + 8] = w[8:]
        put(nv, AUX2, (((1 << sh) - 1) << ((32 - sh) % 32)) & M)
        nv[AUX2 + 2] = sh * sh
        generate_divu_helper(lv, nv, op, IN1, IN2, AUX2, None)
        return lv, nv
    if op in (IS_MFHI, IS_MTHI, IS_MFLO, IS_MTLO):                          # lo_hi.rs:13-22
        put(lv, IN0, a); put(lv, OUT, a)
        return lv, None
    raise ValueError(op)


def generate_trace(ops, log_n=16):
    """ArithmeticStark::generate_trace (arithmetic_stark.rs:162-199) + generate_range_checks (:138-160).  ops: (op, a, b)."""
    n = 1 << log_n
    assert n >= 1 << 16, "the range-check table needs 2^16 rows"
    rows = []
    for op, a, b in ops:
        r1, r2 = rows_for(op, a, b)
        rows.append(r1)
        if r2 is not None:
            rows.append(r2)
    assert len(rows) <= n
    tr = np.zeros((NCOLS, n), dtype=np.uint64)
    if rows:
        tr[:, :len(rows)] = np.array(rows, dtype=np.uint64).T
    tr[RANGE_COUNTER] = np.minimum(np.arange(n), 65535)
    shared = tr[26:44].reshape(-1)
    assert int(shared.max()) < 1 << 16
    tr[RC_FREQ] += np.bincount(shared.astype(np.int64), minlength=n)[:n].astype(np.uint64)
    return np.ascontiguousarray(tr).reshape(-1)


def random_ops(seed, count, which=None):
    rng = np.random.default_rng(seed)
    all_ops = list(range(26)) if which is None else list(which)
    ops = []
    for i in range(count):
        op = all_ops[i % len(all_ops)]
        a, b = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
        if i % 5 == 0:
            a = [0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF][(i // 5) % 5]
        if i % 7 == 0:
            b = [1, 0xFFFFFFFF, 0x80000000, 2, 0x10000][(i // 7) % 5]
        if op in (IS_DIV, IS_DIVU) and b == 0:
            b = 3
        if op == IS_DIV and a == 0x80000000 and b == 0xFFFFFFFF:
            b = 7                                   # i32::MIN / -1 overflows in the reference too
        if op in (IS_SLL, IS_SLLV, IS_SRL, IS_SRLV, IS_SRA, IS_SRAV):
            b = int(rng.integers(0, 32))
        if op == IS_LUI:
            a &= 0xFFFF
        ops.append((op, a, b))
    return ops
