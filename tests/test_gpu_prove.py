"""GPU parity: prove_single_table through the C ABI vs the CPU oracle (bit-exact proofs), and acceptance of
GPU proofs by the oracle's verifier at sizes the oracle prover cannot reach in seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def fake_ctl_aux(n):
    # poseidon_benchmark's fake CTL data (poseidon_stark.rs:786-799): zero helper column + zero Z, per challenge
    return np.zeros(4 * n, dtype=np.uint64), [1, 1]


@pytest.mark.parametrize("log_n", [5, 6, 9])
def test_quotient_matches_oracle(ctx, zkm, oracle, log_n):
    n = 1 << log_n
    trace = oracle.poseidon_trace(11, n - 2, log_n)
    rng = np.random.default_rng(log_n)
    # non-trivial aux columns so the CTL terms are exercised (the quotient need not be a polynomial identity here)
    aux = rng.integers(0, P, 4 * n, dtype=np.uint64)
    tb = zkm.PolynomialBatch.from_values(ctx, trace, 262, log_n)
    ab = zkm.PolynomialBatch.from_values(ctx, aux, 4, log_n)
    otb, oab = oracle.batch_from_values(trace, 262, log_n), oracle.batch_from_values(aux, 4, log_n)
    alphas = rng.integers(0, P, 2, dtype=np.uint64)
    got = ctx.quotient(tb, ab, [1, 1], alphas)
    want = oracle.quotient_poseidon(otb, oab, [1, 1], alphas)
    assert (got == want).all()
    # 3 helper columns on one Z, 1 challenge
    aux5 = rng.integers(0, P, 4 * n, dtype=np.uint64)
    ab5 = zkm.PolynomialBatch.from_values(ctx, aux5, 4, log_n)
    got = ctx.quotient(tb, ab5, [3], alphas[:1])
    want = oracle.quotient_poseidon(otb, oracle.batch_from_values(aux5, 4, log_n), [3], alphas[:1])
    assert (got == want).all()
    for b in (tb, ab, ab5):
        b.free()


def test_eval_openings(ctx, zkm, oracle):
    rng = np.random.default_rng(5)
    for log_n, ncols in ((3, 2), (8, 5), (15, 3)):
        n = 1 << log_n
        vals = rng.integers(0, P, ncols * n, dtype=np.uint64)
        b = zkm.PolynomialBatch.from_coeffs(ctx, vals, ncols, log_n)
        zeta = [int(x) for x in rng.integers(0, P, 2, dtype=np.uint64)]
        got = ctx.eval_openings(b, zeta).reshape(ncols, 2)
        # Horner in F2 = F[X]/(X^2 - 7) with python ints
        for c in range(ncols):
            a0 = a1 = 0
            for k in range(n - 1, -1, -1):
                a0, a1 = (a0 * zeta[0] + 7 * a1 * zeta[1] + int(vals[c * n + k])) % P, (a0 * zeta[1] + a1 * zeta[0]) % P
            assert [int(got[c][0]), int(got[c][1])] == [a0, a1]
        b.free()


@pytest.mark.parametrize("log_n", [5, 7, 8, 10, 12])
def test_proof_is_bit_exact(ctx, zkm, oracle, log_n):
    n = 1 << log_n
    trace = oracle.poseidon_trace(seed=log_n, num_perms=n - 5, log_n=log_n)
    aux, nh = fake_ctl_aux(n)
    want = oracle.prove(trace, log_n, aux, nh)
    got = ctx.prove_single_table(trace, log_n, aux, nh)
    assert got.size == want.size
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing proof word: %d" % bad[0]
    assert oracle.verify(got, 4, nh) == 0


def test_proof_with_existing_commitment_and_transcript(ctx, zkm, oracle):
    log_n = 7
    n = 1 << log_n
    trace = oracle.poseidon_trace(9, n, log_n)
    aux, nh = fake_ctl_aux(n)
    och = oracle.challenger()
    oracle.observe(och, [5, 6, 7])
    want = oracle.prove(trace, log_n, aux, nh, challenger=och)
    tb = zkm.PolynomialBatch.from_values(ctx, trace, 262, log_n)
    ch = zkm.challenger_new()
    zkm.challenger_observe(ch, [5, 6, 7])
    got = ctx.prove_single_table(None, log_n, aux, nh, challenger=ch, trace_batch=tb)
    assert (got == want).all()
    assert list(ch.state) == list(och.state) and ch.n_in == och.n_in and ch.n_out == och.n_out
    tb.free()


def test_device_generated_trace_proof_verifies_2_16(ctx, zkm, oracle):
    # config 1 size (2^16 rows): GPU witness + GPU proof, accepted by the oracle's verifier
    log_n = 16
    n = 1 << log_n
    trace = ctx.poseidon_trace(1, n, log_n)
    aux = ctx.alloc(4 * n).upload(np.zeros(4 * n, dtype=np.uint64))
    proof = ctx.prove_single_table(trace, log_n, aux, [1, 1])
    assert oracle.verify(proof, 4, [1, 1]) == 0
    bad = proof.copy()
    bad[2000] ^= 1
    assert oracle.verify(bad, 4, [1, 1]) != 0
    trace.free()
    aux.free()


def test_full_size_proof_verifies_2_20(ctx, zkm, oracle):
    # BASELINE config 2 (2^20 rows x 262 columns): size-independent acceptance check
    log_n = 20
    n = 1 << log_n
    trace = ctx.poseidon_trace(2, n, log_n)
    aux = ctx.alloc(4 * n).upload(np.zeros(4 * n, dtype=np.uint64))
    proof = ctx.prove_single_table(trace, log_n, aux, [1, 1])
    assert oracle.verify(proof, 4, [1, 1]) == 0
    trace.free()
    aux.free()


def test_invalid_witness_is_rejected(ctx, zkm, oracle):
    log_n = 6
    n = 1 << log_n
    trace = oracle.poseidon_trace(4, n, log_n).copy()
    trace[50 * n + 7] = (int(trace[50 * n + 7]) + 1) % P
    aux, nh = fake_ctl_aux(n)
    got = ctx.prove_single_table(trace, log_n, aux, nh)
    assert (got == oracle.prove(trace, log_n, aux, nh)).all()  # same bytes as the oracle ...
    assert oracle.verify(got, 4, nh) != 0                      # ... and rejected like the oracle's


def test_bad_arguments(ctx, zkm):
    n = 32
    with pytest.raises(zkm.ZkmError):
        ctx.prove_single_table(np.zeros(262 * n, dtype=np.uint64), 5, np.zeros(3 * n, dtype=np.uint64), [1, 1])
    with pytest.raises(zkm.ZkmError):
        ctx.prove_single_table(np.zeros(100 * n, dtype=np.uint64), 5, np.zeros(4 * n, dtype=np.uint64), [1, 1], ncols=100)


@pytest.mark.parametrize("log_n", [5, 9, 13])
def test_prove_openings_alone_is_bit_exact(ctx, zkm, oracle, log_n):
    # BASELINE config 4 shape (13 + 4 + 4 polynomials, random), small sizes: GPU == oracle bytes, FRI verifies
    rng = np.random.default_rng(40 + log_n)
    n = 1 << log_n
    tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (13, 4, 4))
    otb, oab, oqb = oracle.batch_from_values(tv, 13, log_n), oracle.batch_from_values(av, 4, log_n), oracle.batch_from_coeffs(qc, 4, log_n)
    tb, ab = zkm.PolynomialBatch.from_values(ctx, tv, 13, log_n), zkm.PolynomialBatch.from_values(ctx, av, 4, log_n)
    qb = zkm.PolynomialBatch.from_coeffs(ctx, qc, 4, log_n)
    och, ch = oracle.challenger(), zkm.challenger_new()
    oracle.observe(och, [42])
    zkm.challenger_observe(ch, [42])
    want = oracle.prove_openings(otb, oab, oqb, 2, challenger=och)
    got = ctx.prove_openings(tb, ab, qb, 2, challenger=ch)
    assert (got == want).all()
    vch = oracle.challenger()
    oracle.observe(vch, [42])
    assert oracle.verify_openings(got, 13, 4, 2, challenger=vch) == 0
    for b in (tb, ab, qb):
        b.free()


def test_full_fri_on_2_22_rows_verifies(ctx, zkm, oracle):
    # BASELINE config 4: full FRI commit + fold on a 2^22-row trace (13 columns + 4 aux + 4 quotient chunks):
    # final polynomial 2^22 F2 coefficients, LDE 2^24, folds [4,4,4,4,4], final poly 4 coefficients
    log_n = 22
    n = 1 << log_n
    rng = np.random.default_rng(4)
    def dev_random(cols):
        buf = ctx.alloc(cols * n)
        buf.upload(rng.integers(0, P, cols * n, dtype=np.uint64))
        return buf
    tvals, avals, qco = dev_random(13), dev_random(4), dev_random(4)
    tb = zkm.PolynomialBatch.from_values(ctx, tvals, 13, log_n)
    ab = zkm.PolynomialBatch.from_values(ctx, avals, 4, log_n)
    qb = zkm.PolynomialBatch.from_coeffs(ctx, qco, 4, log_n)
    proof = ctx.prove_openings(tb, ab, qb, 2)
    assert int(proof[7]) == 5 and int(proof[8]) == 4  # 5 FRI layers, 4 final coefficients
    assert oracle.verify_openings(proof, 13, 4, 2) == 0
    bad = proof.copy()
    bad[-5] ^= 1
    assert oracle.verify_openings(bad, 13, 4, 2) != 0
    for b in (tb, ab, qb):
        b.free()
    for b in (tvals, avals, qco):
        b.free()


def test_no_device_memory_is_leaked(zkm, oracle):
    """Live allocator bytes return to the baseline after proofs, multi-table proofs and failing calls (every error path releases
    its scratch); only the reuse cache grows."""
    from zkm_amd import tables as T
    from . import logic_fixtures
    c = zkm.Context(0)
    trace = oracle.poseidon_trace(3, 100, 7)
    aux = np.zeros(4 << 7, dtype=np.uint64)
    def transient():   # live bytes that are not resident tables (twiddles / power tables stay by design; the commit lanes of
        live, _ = c.memory()   # prove_with_traces build their own on first use, whichever lane a table lands on)
        return live - c.resident_bytes()
    c.prove_single_table(trace, 7, aux, [1, 1])
    base, _ = c.memory()
    assert transient() == 0
    for _ in range(3):
        c.prove_single_table(trace, 7, aux, [1, 1])
    assert c.memory()[0] == base and transient() == 0
    with pytest.raises(zkm.ZkmError):
        c.prove_single_table(trace, 7, aux, [1, 1], table_id=99)
    with pytest.raises(zkm.ZkmError):
        c.prove_single_table(trace, 7, np.zeros(3 << 7, dtype=np.uint64), [1, 1])        # aux does not match the CTL description
    assert transient() == 0
    t4, c4, _ = logic_fixtures.build4(oracle, log_sponge=3)
    for _ in range(3):
        c.prove_with_traces(t4, c4)
        assert transient() == 0
    # a multi-table call that fails late -- all trace and auxiliary commitments exist, the last table has no constraint kernel --
    # releases everything as well (commitments built on the lanes included)
    bad = list(t4)
    bad[-1] = (99,) + tuple(bad[-1][1:])
    with pytest.raises(zkm.ZkmError):
        c.prove_with_traces(bad, c4)
    assert transient() == 0
    live, _ = c.memory()
    # the only growth since the first baseline is resident tables (for the new sizes, on the context and its lanes)
    assert live - base < 64 << 20 and live - base <= c.resident_bytes()
    c.close()


@pytest.mark.parametrize("log_n", [9, 18])
def test_generic_fri_instance_equals_the_stark_instance(ctx, zkm, oracle, log_n):
    """zkm_fri_prove takes the FriInstanceInfo as data (oracles + batches of (oracle, polynomial) at a point).  Fed the three oracles
    and three batches of the STARK instance (stark.rs:91-148) and the transcript state prove_single_table has after observing the
    openings, it must reproduce the FRI part of zkm_prove_openings' blob word for word; a second, different instance (two oracles,
    one batch, other point) must at least be self-consistent with the oracle-checked STARK path on its shared pieces."""
    import ctypes as C
    W, A, Q, Z = 13, 4, 4, 2            # (log_n 18: the batches keep their coefficients in the digit layout of the two-pass inverse transform)
    rng = np.random.default_rng(91)
    n = 1 << log_n
    tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A, Q))
    tb, ab = zkm.PolynomialBatch.from_values(ctx, tv, W, log_n), zkm.PolynomialBatch.from_values(ctx, av, A, log_n)
    qb = zkm.PolynomialBatch.from_coeffs(ctx, qc, Q, log_n)
    ch1 = zkm.challenger_new()
    zkm.challenger_observe(ch1, [7, 8, 9])
    blob = ctx.prove_openings(tb, ab, qb, Z, challenger=ch1)
    lay, q = zkm.proof_layout(blob)
    # replay the transcript up to the point where prove_openings draws alpha: compact, zeta, observe_openings (proof.rs:336-367)
    ch2 = zkm.challenger_new()
    zkm.challenger_observe(ch2, [7, 8, 9])
    st = np.zeros(12, dtype=np.uint64)
    zkm.load().zkm_challenger_compact(C.byref(ch2), st.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert (st == blob[lay.init_challenger_state:lay.init_challenger_state + 12]).all()
    zeta = (int(zkm.challenger_get(ch2)), int(zkm.challenger_get(ch2)))
    g = oracle.root_of_unity(log_n)
    zeta_next = (zeta[0] * g % P, zeta[1] * g % P)
    sec = lambda off, words: blob[off:off + words]
    for off, words in ((lay.local_values, 2 * W), (lay.aux_polys, 2 * A), (lay.quotient_polys_open, 2 * Q), (lay.next_values, 2 * W),
                       (lay.aux_polys_next, 2 * A)):
        zkm.challenger_observe(ch2, sec(off, words))
    for i in range(Z):
        zkm.challenger_observe(ch2, [int(blob[lay.ctl_zs_first + i]), 0])
    polys0 = [(0, c) for c in range(W)] + [(1, c) for c in range(A)] + [(2, c) for c in range(Q)]
    polys1 = [(0, c) for c in range(W)] + [(1, c) for c in range(A)]
    polys2 = [(1, c) for c in range(A - Z, A)]
    fri = ctx.fri_prove([tb, ab, qb], [(zeta, polys0), (zeta_next, polys1), ((1, 0), polys2)], ch2)
    assert int(fri[0]) == int.from_bytes(b"ZKMFRIPF", "little") and [int(x) for x in fri[1:9]] == [log_n, 3, 4, int(lay.fri_layers),
                                                                                              int(lay.final_poly_len), 37, 2, 4]
    assert [int(x) for x in fri[16:19]] == [W, A, Q]
    assert (fri[24:] == blob[lay.commit_phase_merkle_caps:]).all()       # caps, final polynomial, PoW witness, all 37 query rounds
    assert list(ch2.state) == list(ch1.state) and ch2.n_in == ch1.n_in and ch2.n_out == ch1.n_out
    # another instance: two oracles, one batch at another point -- accepted shapes and error paths
    fri2 = ctx.fri_prove([ab, qb], [((5, 6), [(1, 3), (0, 0), (1, 0)])], zkm.challenger_new())
    assert int(fri2[2]) == 2 and [int(x) for x in fri2[16:18]] == [A, Q]
    with pytest.raises(zkm.ZkmError):
        ctx.fri_prove([ab, qb], [((5, 6), [(2, 0)])], zkm.challenger_new())          # oracle index out of range
    with pytest.raises(zkm.ZkmError):
        ctx.fri_prove([ab, qb], [((P, 0), [(0, 0)])], zkm.challenger_new())          # non-canonical point
    for b in (tb, ab, qb):
        b.free()


@pytest.mark.parametrize("log_n", [10, 16])
def test_fri_instance_with_eight_batches(zkm, log_n):
    """zkm_fri_prove admits up to FRI_MAX_BATCHES = 8 opening batches, and the one-launch bottom level of the division by
    (X - z) (k_seg_scan_combine) keeps two LDS tiles per batch: eight batches are 73,728 B, over the 64 KB a launch gets without a
    raised limit.  The instance must prove, and the blob must equal the one the two-launch path (k_seg_scan + k_seg_combine,
    fri_scan_combine 0) makes -- the path the oracle-checked tests of round 4 ran on.  A ninth batch is refused."""
    W, A = 6, 4
    rng = np.random.default_rng(95)
    n = 1 << log_n
    tv, av = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A))
    batches = [((3 + b, 10 * b + 1), [(b & 1, c) for c in range(1 + b % 4)] + [((b + 1) & 1, b % 4)]) for b in range(8)]
    blobs = []
    for combine in (1, 0):
        c = zkm.Context(0)
        c.set_tuning("fri_scan_combine", combine)
        tb, ab = zkm.PolynomialBatch.from_values(c, tv, W, log_n), zkm.PolynomialBatch.from_values(c, av, A, log_n)
        ch = zkm.challenger_new()
        zkm.challenger_observe(ch, [4, 2])
        blobs.append(c.fri_prove([tb, ab], batches, ch))
        if combine:
            with pytest.raises(zkm.ZkmError):
                c.fri_prove([tb, ab], batches + [((9, 9), [(0, 0)])], zkm.challenger_new())
        tb.free()
        ab.free()
        c.close()
    assert int(blobs[0][0]) == int.from_bytes(b"ZKMFRIPF", "little")
    assert blobs[0].size == blobs[1].size and (blobs[0] == blobs[1]).all()


def test_concurrent_contexts_are_bit_exact(ctx, zkm):
    """bench.py's throughput mode: k contexts on one GPU, one host thread each, proving independent segments side by side
    (zkm_amd.dist.run_workers).  Every proof made that way must equal the proof the same segment gets from one context working
    alone -- shared traces are read-only, twiddle / power tables, allocator, stream and transcript are per context."""
    from zkm_amd.dist import run_workers
    log_n = 14
    n = 1 << log_n
    seeds = list(range(300, 310))
    traces = {s: ctx.poseidon_trace(s, n - 7, log_n) for s in seeds}
    aux_host, nh = fake_ctl_aux(n)
    aux = ctx.alloc(aux_host.size).upload(aux_host)
    alone = {s: ctx.prove_single_table(traces[s], log_n, aux, nh) for s in seeds}
    ctx.synchronize()
    workers = [zkm.Context(0) for _ in range(3)]
    try:
        for rep in range(2):       # second round: warm allocators, different interleaving
            got = run_workers(lambda s, w: workers[w].prove_single_table(traces[s], log_n, aux, nh), seeds, len(workers))
            assert sorted(got) == seeds
            for s in seeds:
                assert (got[s] == alone[s]).all(), (rep, s)
    finally:
        for w in workers:
            w.close()
    for t in traces.values():
        t.free()
    aux.free()


# ---- zkm_ctx_set_tuning: both sides of every kernel-selection threshold produce the same words
@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [9, 13])
def test_tuning_fri_division_variants_agree(zkm, oracle, log_n):
    """Division by (X - z): one batch per workgroup column + a combine launch (the default at every size since round 4) against all
    batches in one workgroup (k_seg_scan_final, behind the "fri_fused_division_min" threshold) -- forced both ways at small sizes."""
    W, A, Q, Z = 13, 4, 4, 2
    rng = np.random.default_rng(92)
    n = 1 << log_n
    tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A, Q))
    blobs = []
    # (third variant, round 5's default: the bottom level's scan and weighted sum in one launch, "fri_scan_combine"; 0 = two launches)
    for threshold, fused_bottom in ((1 << 30, 1), (1, 1), (1 << 30, 0)):
        c = zkm.Context(0)
        c.set_tuning("fri_scan_combine", fused_bottom)
        c.set_tuning("fri_fused_division_min", threshold)
        tb, ab = zkm.PolynomialBatch.from_values(c, tv, W, log_n), zkm.PolynomialBatch.from_values(c, av, A, log_n)
        qb = zkm.PolynomialBatch.from_coeffs(c, qc, Q, log_n)
        ch = zkm.challenger_new()
        zkm.challenger_observe(ch, [7, 8, 9])
        blobs.append(c.prove_openings(tb, ab, qb, Z, challenger=ch))
        for b in (tb, ab, qb):
            b.free()
        c.close()
    assert (blobs[0] == blobs[1]).all() and (blobs[0] == blobs[2]).all()
    otb, oab = oracle.batch_from_values(tv, W, log_n), oracle.batch_from_values(av, A, log_n)
    oqb = oracle.batch_from_coeffs(qc, Q, log_n)
    och = oracle.challenger()
    oracle.observe(och, [7, 8, 9])
    assert (blobs[0] == oracle.prove_openings(otb, oab, oqb, Z, och)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nalphas", [1, 2])
def test_tuning_keccak_quotient_variants_agree(zkm, oracle, nalphas):
    """KeccakStark constraints: 25 threads per point (default up to 2^15 points) against one thread per point (default beyond;
    verified at 2^15 rows by test_large_tables_gpu_proof_verifies) -- both forced at 2^6 rows, both equal to the oracle."""
    from zkm_amd import tables as T
    from .test_gpu_tables import table_trace
    log_n, W = 6, T.WIDTH[T.TABLE_KECCAK]
    trace = table_trace(oracle, T.TABLE_KECCAK, log_n)
    rng = np.random.default_rng(6)
    aux = rng.integers(0, 1 << 63, 3 << log_n, dtype=np.uint64)
    alphas = [int(x) for x in rng.integers(1, 1 << 62, nalphas)]
    want = oracle.quotient(oracle.batch_from_values(trace, W, log_n), oracle.batch_from_values(aux, 3, log_n), [2], alphas, table_id=T.TABLE_KECCAK)
    for max_points in (1 << 15, 0):
        c = zkm.Context(0)
        c.set_tuning("keccak_parts_max_points", max_points)
        tb, ab = zkm.PolynomialBatch.from_values(c, trace, W, log_n), zkm.PolynomialBatch.from_values(c, aux, 3, log_n)
        assert (c.quotient(tb, ab, [2], alphas, table_id=T.TABLE_KECCAK) == want).all(), max_points
        tb.free(); ab.free()
        c.close()


@pytest.mark.gpu
def test_tuning_rejects_unknown_keys(ctx, zkm):
    with pytest.raises(zkm.ZkmError):
        ctx.set_tuning("no_such_threshold", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,W", [(8, 70), (12, 9)])
def test_tuning_poseidon_forms_agree(zkm, oracle, log_n, W):
    """One lane, four lanes or sixteen lanes per hash (leaves, tree levels, FRI layers): every choice of the two thresholds gives the
    commitments and the openings proof of the oracle."""
    A, Q, Z = 4, 4, 2
    rng = np.random.default_rng(93)
    n = 1 << log_n
    tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A, Q))
    otb, oab, oqb = oracle.batch_from_values(tv, W, log_n), oracle.batch_from_values(av, A, log_n), oracle.batch_from_coeffs(qc, Q, log_n)
    want = oracle.prove_openings(otb, oab, oqb, Z)
    for wide, quad in ((0, 0), (0, 1 << 30), (1 << 30, 1 << 30), (64, 512), (1024, 32768)):
        c = zkm.Context(0)
        c.set_tuning("wide_max_hashes", wide)
        c.set_tuning("quad_max_hashes", quad)
        tb, ab = zkm.PolynomialBatch.from_values(c, tv, W, log_n), zkm.PolynomialBatch.from_values(c, av, A, log_n)
        qb = zkm.PolynomialBatch.from_coeffs(c, qc, Q, log_n)
        assert (tb.cap() == otb.cap()).all() and (qb.cap() == oqb.cap()).all(), (wide, quad)
        for lvl in range(0, log_n + 2 - 4, 3):
            assert (tb.digest_layer(lvl) == otb.digest_layer(lvl)).all(), (wide, quad, lvl)
        assert (c.prove_openings(tb, ab, qb, Z) == want).all(), (wide, quad)
        for b in (tb, ab, qb):
            b.free()
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,W", [(5, 9), (7, 13), (9, 70), (11, 33), (13, 9), (14, 5)])
def test_tuning_single_launch_small_paths_agree(zkm, oracle, log_n, W):
    """Round 4's single-launch paths for short tables -- whole transforms of 2^9 .. 2^13 points in one workgroup (k_ntt_small: inverse
    transform natural -> natural, coset LDE as four blocks) and the tree tail that climbs from <= 2^11 nodes to the cap and delivers it
    to the host itself (k_merkle_tail) -- against the two-pass transforms and the level kernels + download: coefficients, every digest
    layer, cap and the openings proof, both ways, equal to the oracle's.  Heights on both sides of each size limit."""
    A, Q, Z = 4, 4, 2
    rng = np.random.default_rng(94)
    n = 1 << log_n
    tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A, Q))
    otb, oab, oqb = oracle.batch_from_values(tv, W, log_n), oracle.batch_from_values(av, A, log_n), oracle.batch_from_coeffs(qc, Q, log_n)
    want = oracle.prove_openings(otb, oab, oqb, Z)
    for small_ntt, tree_tail in ((1, 1), (0, 0), (1, 0), (0, 1)):
        c = zkm.Context(0)
        c.set_tuning("small_ntt", small_ntt)
        c.set_tuning("tree_tail", tree_tail)
        tb, ab = zkm.PolynomialBatch.from_values(c, tv, W, log_n), zkm.PolynomialBatch.from_values(c, av, A, log_n)
        qb = zkm.PolynomialBatch.from_coeffs(c, qc, Q, log_n)
        key = (small_ntt, tree_tail)
        assert (tb.coeffs() == otb.coeffs()).all(), key
        assert (tb.cap() == otb.cap()).all() and (ab.cap() == oab.cap()).all() and (qb.cap() == oqb.cap()).all(), key
        for lvl in range(0, log_n + 2 - 4 + 1):
            assert (tb.digest_layer(lvl) == otb.digest_layer(lvl)).all(), (key, lvl)
        for i in (0, 1, 4 * n - 1):
            assert (tb.lde_row(i) == otb.lde_row(i)).all(), (key, i)
        assert (c.prove_openings(tb, ab, qb, Z) == want).all(), key
        for b in (tb, ab, qb):
            b.free()
        live, _ = c.memory()
        assert live == c.resident_bytes(), key          # (the tail kernel's ticket word is resident, nothing else stays behind)
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arity_bits,final_poly_bits,log_n", [(2, 3, 9), (3, 2, 10), (5, 4, 12), (6, 0, 13), (5, 5, 11)])
def test_every_admitted_fri_arity_proves_like_the_oracle(ctx, zkm, oracle, arity_bits, final_poly_bits, log_n):
    """validate_config admits arity_bits 2 .. 6, and the LDS-tiled fold kernel of round 4 stopped at arity 16 -- a config with
    arity_bits 5 or 6 passed validation and failed after all commitments.  The fold tile is now sized by the arity (dynamic LDS): every
    admitted arity proves, word for word like the oracle (FriReductionStrategy::ConstantArityBits, config.rs:25), and verifies."""
    n = 1 << log_n
    cfg = ctx.standard_config()
    cfg.arity_bits, cfg.final_poly_bits = arity_bits, final_poly_bits
    ocfg = oracle.standard_config()
    ocfg.arity_bits, ocfg.final_poly_bits = arity_bits, final_poly_bits
    trace_dev = ctx.poseidon_trace(seed=40 + arity_bits, num_perms=n - 5, log_n=log_n)
    trace = trace_dev.download()
    aux = np.zeros(4 * n, dtype=np.uint64)
    got = ctx.prove_single_table(trace_dev, log_n, aux, [1, 1], cfg=cfg)
    want = oracle.prove(trace, log_n, aux, [1, 1], cfg=ocfg)
    assert got.size == want.size and (got == want).all()
    assert oracle.verify(got, 4, [1, 1], cfg=ocfg) == 0
    assert int(got[11]) == arity_bits
    trace_dev.free()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,K", [(6, 2), (11, 3), (14, 2), (18, 2), (20, 4)])
def test_single_table_proofs_in_lockstep_equal_single_proofs(ctx, zkm, oracle, oracle_proof_2_20, log_n, K):
    """zkm_prove_single_tables: K PoseidonStark proofs (own witness seed, own transcript each) as ONE lock-step call -- stacked trace /
    auxiliary / quotient commitments, the device-resident traces transformed where they lie -- against K zkm_prove_single_table calls,
    and one of them against the oracle.  2^18 rows: the digit coefficient layout.  (20, 4): the call bench.py's timed region makes
    (four 262 x 2^20 proofs per call) -- proof 0 of THAT call is bench.py's segment 0 (witness seed 100, a fresh transcript) and is held
    against the oracle's 2^20-row proof directly, not only through the single-proof path."""
    n = 1 << log_n
    bench0 = log_n == 20                  # proof 0 = the oracle_proof_2_20 fixture's workload
    traces = [ctx.poseidon_trace(seed=100, num_perms=n, log_n=log_n) if (bench0 and k == 0) else
              ctx.poseidon_trace(seed=60 + k, num_perms=n - 1 - k, log_n=log_n) for k in range(K)]
    aux = np.zeros(4 * n, dtype=np.uint64)

    def transcript(k):
        ch = zkm.challenger_new()
        if not (bench0 and k == 0):
            zkm.challenger_observe(ch, [k, 17])
        return ch
    chs = [transcript(k) for k in range(K)]
    got = ctx.prove_single_tables(traces, log_n, aux, [1, 1], challengers=chs)
    if bench0 and oracle_proof_2_20 is not None:
        ref = oracle_proof_2_20["proof"]
        bad = np.nonzero(got[0] != ref)[0] if got[0].size == ref.size else np.array([-1])
        assert bad.size == 0, "proof 0 of the lock-step call vs the ORACLE's 2^20-row proof: first differing word %d of %d" % (bad[0], ref.size)
    for k in range(K):
        ch = transcript(k)
        want = ctx.prove_single_table(traces[k], log_n, aux, [1, 1], challenger=ch)
        assert (got[k] == want).all(), k
        assert list(ch.state) == list(chs[k].state) and ch.n_in == chs[k].n_in and ch.n_out == chs[k].n_out   # same transcript state
    if log_n <= 14:
        och = oracle.challenger()
        oracle.observe(och, [1, 17])
        ref = oracle.prove(traces[1].download(), log_n, aux, [1, 1], challenger=och)
        assert (got[1] == ref).all()
    for t in traces:
        t.free()
    if log_n >= 20:
        ctx.trim()      # (four proofs' worth of cached buffers: give them back before the next test)
