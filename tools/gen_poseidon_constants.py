#!/usr/bin/env python3
"""Generate the Goldilocks-Poseidon parameter tables used by the oracle and by the HIP kernels.

Inputs (data, not code): the 360 round constants and the circulant MDS first row / diagonal of the
width-12 Goldilocks Poseidon instance.  They are the public parameter set of the hash; the reference
carries them at prover/src/poseidon/constants.rs:11-105.  This script reads them from the reference
checkout when it is present (authoring container) and otherwise from the already generated .inc file,
so it can be re-run anywhere.

Everything else -- the "fast partial round" tables (first-round constant vector, per-round scalar
constants, the dense pre-matrix and the 22 sparse matrices) -- is DERIVED here from those parameters
by the equivalent-matrix factorisation of the Poseidon paper (App. B), and then cross-checked against
the reference's own tables (constants.rs:107-870) when the reference is readable.

Output: one C include file with plain `static const uint64_t` arrays, written to
  zkm_amd/csrc/poseidon_constants.inc   (product)
  oracle/poseidon_constants.inc         (oracle; identical bytes)
"""
import os
import re
import sys

P = 0xFFFFFFFF00000001
WIDTH = 12
HALF_FULL = 4
N_PARTIAL = 22
N_ROUNDS = 2 * HALF_FULL + N_PARTIAL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/prover/src/poseidon/constants.rs"
OUTS = [os.path.join(ROOT, "zkm_amd", "csrc", "poseidon_constants.inc"),
        os.path.join(ROOT, "oracle", "poseidon_constants.inc")]


def parse_rust_arrays(text):
    """Return {NAME: flat list of ints} for every `pub const NAME: [...] = [ ... ];`."""
    out = {}
    for m in re.finditer(r"pub const (\w+): \[[^=]*= \[(.*?)\];", text, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        out[m.group(1)] = [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", body)]
    return out


def parse_inc_arrays(text):
    out = {}
    for m in re.finditer(r"ZKM_CONST uint64_t (\w+)\[[^\]]*\](?:\[[^\]]*\])? = \{(.*?)\};", text, re.S):
        out[m.group(1)] = [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+", m.group(2))]
    return out


def inv(x):
    return pow(x % P, P - 2, P)


def mat_mul(a, b):
    n, m, k = len(a), len(b[0]), len(b)
    return [[sum(a[i][t] * b[t][j] for t in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(a, v):
    return [sum(a[i][j] * v[j] for j in range(len(v))) % P for i in range(len(a))]


def mat_inv(a):
    n = len(a)
    m = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(a)]
    for c in range(n):
        piv = next(r for r in range(c, n) if m[r][c] % P)
        m[c], m[piv] = m[piv], m[c]
        s = inv(m[c][c])
        m[c] = [x * s % P for x in m[c]]
        for r in range(n):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(x - f * y) % P for x, y in zip(m[r], m[c])]
    return [row[n:] for row in m]


def derive_fast(rc, circ, diag):
    """Derive the fast-partial-round tables.

    Dense MDS: M[r][c] = circ[(c - r) mod 12] + (r == c) * diag[r]  (a row of the state is multiplied
    as new[r] = sum_c M[r][c] * old[c]).
    """
    M = [[(circ[(c - r) % WIDTH] + (diag[r] if r == c else 0)) % P for c in range(WIDTH)] for r in range(WIDTH)]
    Minv = mat_inv(M)

    # --- constants: pull every partial round's constant vector backwards through the previous
    # round's MDS; what lands on lane 0 becomes a post-sbox scalar, the rest commutes with the
    # lane-0 sbox and merges into the previous round's vector.
    consts = [rc[(HALF_FULL + r) * WIDTH:(HALF_FULL + r + 1) * WIDTH] for r in range(N_PARTIAL)]
    scalars = [0] * N_PARTIAL
    acc = consts[N_PARTIAL - 1][:]
    for r in range(N_PARTIAL - 2, -1, -1):
        d = mat_vec(Minv, acc)
        scalars[r] = d[0]
        acc = [(consts[r][i] + (d[i] if i else 0)) % P for i in range(WIDTH)]
    first_round = acc

    # --- matrices: M = Sparse . diag(1, Mhat); diag(1, Mhat) commutes with the lane-0 sbox and is
    # merged into the previous round's dense matrix.
    vs = [None] * N_PARTIAL
    w_hats = [None] * N_PARTIAL
    cur = M
    for r in range(N_PARTIAL - 1, -1, -1):
        mhat = [row[1:] for row in cur[1:]]
        mhat_inv = mat_inv(mhat)
        v = cur[0][1:]
        w = [cur[i][0] for i in range(1, WIDTH)]
        # first row of the sparse factor: v'^T = v^T . Mhat^{-1}
        w_hats[r] = [sum(v[k] * mhat_inv[k][j] for k in range(WIDTH - 1)) % P for j in range(WIDTH - 1)]
        vs[r] = w
        dmat = [[1 if i == j else 0 for j in range(WIDTH)] for i in range(WIDTH)]
        for i in range(1, WIDTH):
            for j in range(1, WIDTH):
                dmat[i][j] = mhat[i - 1][j - 1]
        cur = mat_mul(dmat, M)
        last_mhat = mhat
    # the dense pre-matrix is applied as result[c] = sum_r state[r] * INIT[r-1][c-1]  => INIT = Mhat^T
    init = [[last_mhat[c][r] for c in range(WIDTH - 1)] for r in range(WIDTH - 1)]
    return first_round, scalars, vs, w_hats, init


def derive_fused(rc, circ, diag):
    """Tables for the fused partial rounds of the HIP permutation (poseidon_dev.h): the linear layers of three
    consecutive rounds are applied as ONE matrix M^3 whose entries are still small integers (< 2^25, NOT reduced
    mod p), so a row is 24 multiply-adds of 32-bit halves into two 64-bit accumulators.  Groups: 7 x 3 MDS layers
    (rounds 3..23) and one group of 2 (rounds 24, 25).  Per group: c1 / c2 = the constants added to lane 0 after the
    first / second layer, c3 = the constant vector after the last layer (canonical mod p)."""
    M = [[circ[(c - r) % WIDTH] + (diag[r] if r == c else 0) for c in range(WIDTH)] for r in range(WIDTH)]
    imul = lambda a, b: [[sum(a[i][t] * b[t][j] for t in range(WIDTH)) for j in range(WIDTH)] for i in range(WIDTH)]
    M2 = imul(M, M)
    M3 = imul(M2, M)
    for i in range(WIDTH):  # 64-bit accumulators must not overflow: (row sum + delta coefficients) * (2^32 - 1) < 2^64
        assert sum(M3[i]) + M2[i][0] + M[i][0] < 1 << 32 and sum(M2[i]) + M[i][0] < 1 << 32
    K = lambda r: rc[r * WIDTH:(r + 1) * WIDTH]
    add = lambda a, b: [(x + y) % P for x, y in zip(a, b)]
    c1, c2, c3 = [], [], []
    for g in range(7):
        rho = 3 + 3 * g
        k1, k2, k3 = K(rho + 1), K(rho + 2), K(rho + 3)
        c1.append(k1[0])
        c2.append(add(mat_vec(M, k1), k2)[0])
        c3.append(add(add(mat_vec(M2, k1), mat_vec(M, k2)), k3))
    k1, k2 = K(25), K(26)
    c1.append(k1[0])
    c2.append(0)
    c3.append(add(mat_vec(M, k1), k2))
    return M2, M3, c1, c2, c3


def fmt32(name, rows):
    lines = ["ZKM_CONSTEXPR uint32_t %s[%d][%d] = {" % (name, len(rows), len(rows[0]))]
    for r in rows:
        lines.append("    {" + ", ".join("%du" % x for x in r) + "},")
    lines.append("};")
    return "\n".join(lines)


def fmt(name, dims, flat, per_line=4):
    lines = ["ZKM_CONST uint64_t %s%s = {" % (name, "".join("[%d]" % d for d in dims))]
    if len(dims) == 2:
        for r in range(dims[0]):
            row = flat[r * dims[1]:(r + 1) * dims[1]]
            lines.append("    {")
            for i in range(0, len(row), per_line):
                lines.append("        " + " ".join("0x%016xULL," % x for x in row[i:i + per_line]))
            lines.append("    },")
    else:
        for i in range(0, len(flat), per_line):
            lines.append("    " + " ".join("0x%016xULL," % x for x in flat[i:i + per_line]))
    lines.append("};")
    return "\n".join(lines)


def main():
    checked = False
    if os.path.exists(REF):
        ref = parse_rust_arrays(open(REF).read())
        rc, circ, diag = ref["ALL_ROUND_CONSTANTS"], ref["MDS_MATRIX_CIRC"], ref["MDS_MATRIX_DIAG"]
    else:
        cur = parse_inc_arrays(open(OUTS[0]).read())
        rc, circ, diag = cur["ZKM_POSEIDON_RC"][:WIDTH * N_ROUNDS], cur["ZKM_POSEIDON_MDS_CIRC"], cur["ZKM_POSEIDON_MDS_DIAG"]
        ref = None
    assert len(rc) == WIDTH * N_ROUNDS and len(circ) == WIDTH and len(diag) == WIDTH

    first_round, scalars, vs, w_hats, init = derive_fast(rc, circ, diag)
    if ref is not None:
        flat = lambda m: [x for row in m for x in row]
        assert first_round == ref["FAST_PARTIAL_FIRST_ROUND_CONSTANT"], "first-round constant mismatch"
        assert scalars == ref["FAST_PARTIAL_ROUND_CONSTANTS"], "per-round scalar mismatch"
        assert flat(vs) == ref["FAST_PARTIAL_ROUND_VS"], "VS mismatch"
        assert flat(w_hats) == ref["FAST_PARTIAL_ROUND_W_HATS"], "W_HATS mismatch"
        assert flat(init) == ref["FAST_PARTIAL_ROUND_INITIAL_MATRIX"], "initial matrix mismatch"
        checked = True

    M2, M3, fc1, fc2, fc3 = derive_fused(rc, circ, diag)
    body = [
        "/* Goldilocks Poseidon (width 12, 4+22+4 rounds, x^7) parameter tables.",
        " * GENERATED by tools/gen_poseidon_constants.py -- do not edit.",
        " * Round constants + MDS row/diag: the public parameter set (reference carries it at",
        " * prover/src/poseidon/constants.rs:11-105).  FAST_* tables: derived by the generator",
        " * (equivalent-matrix factorisation) and cross-checked against constants.rs:107-870.",
        " * ZKM_CONST is the storage qualifier (default `static const`; the HIP side re-includes this",
        " * file with `static __device__ __constant__ const`). */",
        "#ifndef ZKM_CONST",
        "#define ZKM_CONST static const",
        "#endif",
        "/* (one more row of zeros after the 30 x 12 round constants: \"the constants added after the last round\", so that code which",
        " * folds the NEXT round's constants into a linear layer reads row r + 1 unconditionally) */",
        fmt("ZKM_POSEIDON_RC", [WIDTH * (N_ROUNDS + 1)], rc + [0] * WIDTH),
        fmt("ZKM_POSEIDON_MDS_CIRC", [WIDTH], circ),
        fmt("ZKM_POSEIDON_MDS_DIAG", [WIDTH], diag),
        fmt("ZKM_POSEIDON_FAST_FIRST_RC", [WIDTH], first_round),
        fmt("ZKM_POSEIDON_FAST_RC", [N_PARTIAL], scalars),
        fmt("ZKM_POSEIDON_FAST_VS", [N_PARTIAL, WIDTH - 1], [x for r in vs for x in r], 4),
        fmt("ZKM_POSEIDON_FAST_W_HATS", [N_PARTIAL, WIDTH - 1], [x for r in w_hats for x in r], 4),
        fmt("ZKM_POSEIDON_FAST_INIT", [WIDTH - 1, WIDTH - 1], [x for r in init for x in r], 4),
        "/* fused partial rounds (tools/gen_poseidon_constants.py derive_fused): integer powers of the MDS matrix and the",
        " * per-group constants; ZKM_CONSTEXPR = compile-time table (`static constexpr` in C++, `static const` in C) */",
        "#ifndef ZKM_CONSTEXPR",
        "#define ZKM_CONSTEXPR static const",
        "#endif",
        fmt32("ZKM_POSEIDON_M2", M2),
        fmt32("ZKM_POSEIDON_M3", M3),
        fmt("ZKM_POSEIDON_FUSED_C1", [8], fc1),
        fmt("ZKM_POSEIDON_FUSED_C2", [8], fc2),
        fmt("ZKM_POSEIDON_FUSED_C3", [8, WIDTH], [x for r in fc3 for x in r], 4),
        "",
    ]
    text = "\n".join(body)
    for path in OUTS:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    print("wrote", ", ".join(OUTS), "(cross-checked against reference)" if checked else "(reference absent: derivation not cross-checked)")


if __name__ == "__main__":
    sys.exit(main())
