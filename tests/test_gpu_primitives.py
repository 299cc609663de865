"""GPU parity: HIP kernels behind the C ABI vs the CPU oracle, bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
G = 14293326489335486720
GOLD = os.path.join(os.path.dirname(__file__), "golden")

EDGE = [0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001, 0x7FFFFFFF80000000]


def rand_field(rng, size):
    a = rng.integers(0, P, size, dtype=np.uint64)
    return a


def test_poseidon_permute_batch(ctx, oracle):
    kat = json.load(open(os.path.join(GOLD, "poseidon_kat.json")))
    states = np.array([v["in"] for v in kat["vectors"]], dtype=np.uint64).reshape(-1)
    out = ctx.poseidon_permute_batch(states.copy()).reshape(-1, 12)
    for row, v in zip(out, kat["vectors"]):
        assert [int(x) for x in row] == v["out"]
    rng = np.random.default_rng(0)
    k = 5000
    st = rand_field(rng, 12 * k)
    # edge values in every lane position
    for i, e in enumerate(EDGE):
        st[12 * i:12 * i + 12] = e
        st[12 * (20 + i) + (i % 12)] = e
    got = ctx.poseidon_permute_batch(st.copy())
    assert (got == oracle.poseidon_permute_batch(st)).all()


def test_poseidon_permute_device_resident(ctx, oracle):
    rng = np.random.default_rng(1)
    st = rand_field(rng, 12 * 1000)
    buf = ctx.alloc(st.size).upload(st)
    ctx.poseidon_permute_batch(buf)
    assert (buf.download() == oracle.poseidon_permute_batch(st)).all()
    buf.free()


def test_keccakf_batch(ctx, oracle):
    kat = json.load(open(os.path.join(GOLD, "keccakf_kat.json")))
    v = kat["keccakf"][0]
    out = ctx.keccakf_batch(np.array(v["in"], dtype=np.uint64))
    assert [int(x) for x in out] == v["out"]
    rng = np.random.default_rng(2)
    st = rng.integers(0, 2**64, 25 * 3000, dtype=np.uint64)
    assert (ctx.keccakf_batch(st.copy()) == oracle.keccakf_batch(st)).all()


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 10, 11, 13, 16])
def test_ntt_matches_oracle(ctx, oracle, log_n):
    rng = np.random.default_rng(log_n)
    ncols = 3 if log_n < 14 else 2
    x = rand_field(rng, ncols << log_n)
    k = min(len(EDGE), x.size)
    x[:k] = EDGE[:k]
    for inverse in (False, True):
        for shift in (0, G, 7):
            got = ctx.ntt(x.copy(), ncols, log_n, inverse=inverse, coset_shift=shift)
            assert (got == oracle.ntt(x, log_n, inverse=inverse, coset_shift=shift)).all(), (log_n, inverse, shift)


def test_ntt_roundtrip_large(ctx):
    # size-independent property at 2^20: iNTT(NTT(x)) == x, also on the coset
    rng = np.random.default_rng(20)
    log_n, ncols = 20, 2
    x = rand_field(rng, ncols << log_n)
    y = ctx.ntt(x.copy(), ncols, log_n)
    assert not (y == x).all()
    assert (ctx.ntt(y, ncols, log_n, inverse=True) == x).all()
    y = ctx.ntt(x.copy(), ncols, log_n, coset_shift=G)
    assert (ctx.ntt(y, ncols, log_n, inverse=True, coset_shift=G) == x).all()


@pytest.mark.parametrize("log_n,ncols", [(3, 1), (5, 4), (5, 5), (6, 13), (8, 262), (10, 9), (12, 20), (14, 5)])
def test_commit_matches_oracle(ctx, zkm, oracle, log_n, ncols):
    rng = np.random.default_rng(100 + log_n + ncols)
    vals = rand_field(rng, ncols << log_n)
    b = zkm.PolynomialBatch.from_values(ctx, vals, ncols, log_n)
    ob = oracle.batch_from_values(vals, ncols, log_n)
    assert (b.coeffs() == ob.coeffs()).all()
    assert (b.cap() == ob.cap()).all()
    N = 4 << log_n
    for i in (0, 1, N // 2 + 3, N - 1):
        assert (b.lde_row(i) == ob.lde_row(i)).all()
        assert (b.leaf(i) == ob.leaf(i)).all()
        assert (b.merkle_path(i) == ob.merkle_path(i)).all()
    for level in (0, 1, b.lde_bits - b.cap_height):
        assert (b.digest_layer(level) == ob.digest_layer(level)).all()
    # from_coeffs on the recovered coefficients gives the same commitment
    b2 = zkm.PolynomialBatch.from_coeffs(ctx, b.coeffs(), ncols, log_n)
    assert (b2.cap() == ob.cap()).all()
    b.free()
    b2.free()


@pytest.mark.parametrize("ncols", [5, 8, 9, 15, 16, 17, 24, 31])
def test_leaf_sponge_modes_at_chunk_boundaries(ctx, zkm, oracle, ncols):
    """The one-leaf-per-lane kernel (more than 2^14 LDE rows) only computes what the sponge carries on: capacity rows after a whole
    chunk that is followed by another, all twelve before a ragged chunk, digest rows at the end.  Column counts on either side of
    the multiples of 8 take every combination (8 and 16: the last whole chunk is the end; 9, 17: a ragged chunk of one follows)."""
    log_n = 13
    rng = np.random.default_rng(4000 + ncols)
    vals = rand_field(rng, ncols << log_n)
    b = zkm.PolynomialBatch.from_values(ctx, vals, ncols, log_n)
    ob = oracle.batch_from_values(vals, ncols, log_n)
    assert (b.digest_layer(0) == ob.digest_layer(0)).all()
    assert (b.cap() == ob.cap()).all()
    b.free()


@pytest.mark.parametrize("log_n,ncols", [(5, 13), (6, 262), (9, 9), (11, 16), (13, 135)])
def test_leaf_hashing_on_the_matrix_core_equals_the_vector_form(zkm, oracle, log_n, ncols):
    """One-lane-per-leaf hashing with the MDS layers of the full rounds on the matrix core (tuning key leaf_mfma, poseidon_mfma_dev.h)
    against the multiply-add form and the oracle: every digest.  The latency forms are switched off so that short matrices take the
    one-lane kernel too -- 2^5 rows = 128 leaves leave half of a 256-lane workgroup past the end (MFMA ignores EXEC: those lanes stay
    alive on the last row); column counts cover the capacity-only, all-rows and digest forms of the last layer."""
    rng = np.random.default_rng(5100 + 7 * log_n + ncols)
    vals = rand_field(rng, ncols << log_n)
    ob = oracle.batch_from_values(vals, ncols, log_n)
    layers = {}
    for mfma in (0, 1):
        c = zkm.Context(0)
        for k in ("wide_max_hashes", "quad_max_hashes"):
            c.set_tuning(k, 0)
        c.set_tuning("leaf_mfma", mfma)
        b = zkm.PolynomialBatch.from_values(c, vals, ncols, log_n)
        layers[mfma] = b.digest_layer(0).copy()
        assert (b.cap() == ob.cap()).all(), mfma
        b.free()
        c.close()
    assert (layers[0] == layers[1]).all()
    assert (layers[1] == ob.digest_layer(0)).all()


def test_commit_other_rate_and_cap(ctx, zkm, oracle):
    rng = np.random.default_rng(77)
    log_n, ncols = 6, 11
    vals = rand_field(rng, ncols << log_n)
    for rate_bits, cap_height in ((1, 0), (3, 2), (2, 8)):
        b = zkm.PolynomialBatch.from_values(ctx, vals, ncols, log_n, rate_bits, cap_height)
        ob = oracle.batch_from_values(vals, ncols, log_n, rate_bits, cap_height)
        assert (b.cap() == ob.cap()).all()
        b.free()


def test_commit_device_resident_input(ctx, zkm, oracle):
    rng = np.random.default_rng(78)
    log_n, ncols = 9, 17
    vals = rand_field(rng, ncols << log_n)
    buf = ctx.alloc(vals.size).upload(vals)
    b = zkm.PolynomialBatch.from_values(ctx, buf, ncols, log_n)
    assert (buf.download() == vals).all()  # input is borrowed, not consumed
    assert (b.cap() == oracle.batch_from_values(vals, ncols, log_n).cap()).all()
    b.free()
    buf.free()


def test_poseidon_trace_matches_oracle(ctx, oracle):
    for log_n, perms in ((5, 29), (8, 256), (10, 1000)):
        got = ctx.poseidon_trace(3, perms, log_n).download()
        assert (got == oracle.poseidon_trace(3, perms, log_n)).all()


def test_errors_are_reported(ctx, zkm):
    with pytest.raises(zkm.ZkmError):
        zkm.PolynomialBatch.from_values(ctx, np.zeros(8, dtype=np.uint64), 1, 3, rate_bits=2, cap_height=9)
    with pytest.raises(zkm.ZkmError):
        ctx.ntt(np.zeros(8, dtype=np.uint64), 1, 3, coset_shift=P)


def test_loose_field_primitives_on_edge_words(ctx):
    """The butterflies work on "loose" words (any uint64 stands for its residue).  Their add / subtract take the probable
    correction from the carry-out and the improbable second one (both operands >= p, or a borrow below 2^32) on a wave-uniform
    branch that field data never takes: here every combination of edge words takes it, next to random words, for all lanes of a
    wave and for single lanes of a wave (the branch is per wave, the fix per lane)."""
    M = (1 << 64) - 1
    edge = [0, 1, 2, 0xFFFFFFFF, 0x100000000, P - 2, P - 1, P, P + 1, P + 0xFFFFFFFE, M - 1, M, 0xFFFFFFFF00000000, 0x7FFFFFFFFFFFFFFF, 1 << 63]
    pairs = [(x, y) for x in edge for y in edge]
    rng = np.random.default_rng(5)
    # a wave whose only double-wrapping lane is lane 37, then whole waves of edge pairs, then random words
    a = [int(v) for v in rng.integers(0, P, 64, dtype=np.uint64)] + [x for x, _ in pairs] + [int(v) for v in rng.integers(0, 1 << 64, 500, dtype=np.uint64)]
    b = [int(v) for v in rng.integers(0, P, 64, dtype=np.uint64)] + [y for _, y in pairs] + [int(v) for v in rng.integers(0, 1 << 64, 500, dtype=np.uint64)]
    a[37], b[37] = M, M
    a[5], b[5] = 3, M          # second borrow: 3 - (2^64 - 1)
    # products whose low 64 bits are smaller than their top 32 bits (a b = X 2^96 + small): the borrow of lo - h1 in the reduction,
    # which field data meets with probability < 2^-32 per product and which sits behind a never-taken branch
    for i, (x, y) in enumerate([(1 << 63, 0x7FFFFFFF << 33), (1 << 48, 0xFFFF << 48), (1 << 48, 1 << 48), ((1 << 48) + 1, 0xABCD << 48),
                                (0xFFFFFFFF << 32, 0xFFFFFFFF << 32), (M, M), (P - 1, P - 1), (1 << 32, 1 << 32), (0x1234 << 48, 1)]):
        a[10 + i], b[10 + i] = x, y
    # products with the middle carry c (2^96 c == -c, fed into the reduction as the borrow-in of lo - h1' - c) AND a low word
    # below h1' + c: pick bl odd and al with al bl = small (mod 2^32), bh large, and ah so that the middle word of lo is zero
    slot, tries = 30, 0
    while slot < 62:
        tries += 1
        bl = int(rng.integers(0, 1 << 31)) * 2 + 1
        small = 1 if slot % 2 else int(rng.integers(0, 1 << 16))
        al = small * pow(bl, -1, 1 << 32) % (1 << 32)
        bh = int(rng.integers(1 << 31, 1 << 32))
        m1 = al * bh + ((al * bl) >> 32)
        ah = (-m1 * pow(bl, -1, 1 << 32)) % (1 << 32)
        m2 = ah * bl + m1
        x, y = (ah << 32) | al, (bh << 32) | bl
        if (m2 >> 64) == 1 and (x * y) & M == small:
            a[slot], b[slot] = x, y
            slot += 1
        assert tries < 10000
    got = ctx.field_selftest(a, b)
    want = [[(x + y) % P for x, y in zip(a, b)], [(x - y) % P for x, y in zip(a, b)], [x * (1 << 24) % P for x in a],
            [x * (1 << 48) % P for x in a], [x * (1 << 72) % P for x in a], [x * y % P for x, y in zip(a, b)], [x * y % P for x, y in zip(a, b)]]
    for row, w, name in zip(got, want, ("add", "sub", "2^24", "2^48", "2^72", "mul", "mul (branch-free)")):
        bad = [i for i in range(len(a)) if int(row[i]) != w[i]]
        assert not bad, (name, bad[:5], hex(a[bad[0]]), hex(b[bad[0]]))
