"""GPU parity at the sizes bench.py runs: word-for-word against the CPU oracle.

VERDICT r01 weak #3: the 3-pass NTT plan (lengths >= 2^17), the 2^17-row x 262-column commitment and a
full 2^16-row proof (BASELINE config 1 size) were only round-tripped / verifier-accepted.  Here they are compared
with the oracle element by element.  The oracle NTT is O(n log n) (oracle/commit.c), a 2^22 transform takes
well under a second per column on the host.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
G = 14293326489335486720


def rand_field(rng, size):
    return rng.integers(0, P, size, dtype=np.uint64)


@pytest.mark.parametrize("log_n", [17, 18, 19, 20, 21, 22, 24, 25])
def test_large_ntt_matches_oracle(ctx, oracle, log_n):
    """Every pass plan the library selects for 2^17..2^24 (the LDE of a 2^22-row trace is a 2^24 transform):
    forward, inverse, coset forward, coset inverse -- all natural -> natural through the C ABI.  2^25 is past the pass kernels'
    range: the one-stage-per-launch radix-2 kernels take it (the same path as transforms of fewer than 8 points)."""
    rng = np.random.default_rng(1000 + log_n)
    ncols = 2 if log_n <= 22 else 1
    x = rand_field(rng, ncols << log_n)
    x[:4] = [0, 1, P - 1, 0xFFFFFFFF00000000]
    # (2^25: two of the four variants -- the oracle transform is what takes the time, and the radix-2 path is the same code for all)
    for inverse, shift in ((False, 0), (True, 0), (False, G), (True, G)) if log_n < 25 else ((False, 0), (True, G)):
        got = ctx.ntt(x.copy(), ncols, log_n, inverse=inverse, coset_shift=shift)
        want = oracle.ntt(x, log_n, inverse=inverse, coset_shift=shift)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (log_n, inverse, shift, int(bad[0]), bad.size)


@pytest.mark.parametrize("log_n,ncols", [(15, 7), (17, 262), (18, 5), (18, 70), (19, 2), (20, 3), (21, 1), (22, 1)])
def test_large_commit_matches_oracle(ctx, zkm, oracle, log_n, ncols):
    """from_values at bench-sized transforms: coefficients (all of them), cap, digest layers, sampled LDE rows /
    leaves / Merkle paths.  (17, 262) is the bench's column count on the 3-pass iNTT + 3-pass LDE plan;
    (20, 3) and (22, 1) are the bench's / config 4's row counts (LDE lengths 2^22 / 2^24); 18, 19, 20 are the three stage
    counts (6, 7, 8) of the block-twiddle first LDE pass (k_lde_upper) AND of the two-pass inverse transform with its digit coefficient
    layout (round 3: the coefficients are compared in natural order through zkm_batch_coeffs); (18, 70) takes that path through the
    pipelined column-chunk ingest (host values, >= 64 columns); 15 and 17 are the plain DIF first pass (3 and 5 stages), 21 and 22 the
    2^13-element block kernel with 8 and 9 upper stages."""
    rng = np.random.default_rng(2000 + log_n)
    vals = rand_field(rng, ncols << log_n)
    b = zkm.PolynomialBatch.from_values(ctx, vals, ncols, log_n)
    ob = oracle.batch_from_values(vals, ncols, log_n)
    assert (b.coeffs() == ob.coeffs()).all()
    assert (b.cap() == ob.cap()).all()
    N = 4 << log_n
    idx = [0, 1, 2, N // 4 - 1, N // 4, N // 2 + 3, N - 2, N - 1] + [int(i) for i in rng.integers(0, N, 24)]
    for i in idx:
        assert (b.lde_row(i) == ob.lde_row(i)).all(), i
        assert (b.leaf(i) == ob.leaf(i)).all(), i
        assert (b.merkle_path(i) == ob.merkle_path(i)).all(), i
    for level in (0, 1, 2, 7, b.lde_bits - b.cap_height - 1, b.lde_bits - b.cap_height):
        assert (b.digest_layer(level) == ob.digest_layer(level)).all(), level
    b.free()


def test_proof_is_bit_exact_2_16(ctx, zkm, oracle):
    """BASELINE config 1 size: one full prove_single_table of PoseidonStark 262 x 2^16, GPU bytes == oracle bytes
    (the oracle prover takes a few seconds on the GPU box's host cores)."""
    log_n = 16
    n = 1 << log_n
    trace_dev = ctx.poseidon_trace(16, n - 5, log_n)
    trace = trace_dev.download()
    assert (trace == oracle.poseidon_trace(16, n - 5, log_n)).all()
    aux = np.zeros(4 * n, dtype=np.uint64)
    want = oracle.prove(trace, log_n, aux, [1, 1])
    got = ctx.prove_single_table(trace_dev, log_n, aux, [1, 1])
    assert got.size == want.size
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing proof word: %d" % bad[0]
    assert oracle.verify(got, 4, [1, 1]) == 0
    trace_dev.free()


def test_proof_is_bit_exact_2_20(ctx, zkm, oracle_proof_2_20):
    """The bench workload itself (BASELINE config 2: PoseidonStark 262 x 2^20, the proof bench.py times): GPU proof bytes == oracle
    proof bytes, every word.  The oracle's proof comes from the session fixture (computed once, about a minute on 64 host threads);
    the trace is compared through a checksum and a sample (the oracle regenerates it from the seed)."""
    if oracle_proof_2_20 is None:
        # visible in the summary line ("1 xfailed"), not a silent skip: on such a host the word-for-word claim at 2^20 rows is NOT made
        pytest.xfail("host shows fewer than 32 CPUs: the 2^20-row oracle proof would take the suite's time limit -- the full-size parity "
                     "claim is not made on this host (2^16 rows are compared word for word above)")
    log_n = 20
    n = 1 << log_n
    trace_dev = ctx.poseidon_trace(100, n, log_n)             # bench.py's segment 0 (seed 100)
    trace = trace_dev.download()
    assert int(np.bitwise_xor.reduce(trace)) == oracle_proof_2_20["xor"] and (trace[::4099] == oracle_proof_2_20["sample"]).all()
    del trace
    aux = np.zeros(4 * n, dtype=np.uint64)
    got = ctx.prove_single_table(trace_dev, log_n, aux, [1, 1])
    want = oracle_proof_2_20["proof"]
    assert got.size == want.size
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing proof word: %d of %d" % (bad[0], got.size)
    trace_dev.free()
    # the full-size CPU time next to the GPU's: recorded where the round's profile script can pick it up
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        import json
        json.dump({"what": "CPU oracle (scalar C restatement, OpenMP), one full prove_single_table of PoseidonStark 262 x 2^20 (seed 100)",
                   "full_size_s": oracle_proof_2_20["seconds"], "threads": oracle_proof_2_20["threads"],
                   "wide_threads": oracle_proof_2_20.get("wide_threads"), "cpu_quota": oracle_proof_2_20.get("cpu_quota"), "host_cores": os.cpu_count(),
                   "stage_s": dict(zip(["compute trace commitment", "compute auxiliary polynomials commitment", "compute quotient polys",
                                        "compute quotient commitment", "openings (StarkOpeningSet::new)", "compute openings proof: combine + final LDE",
                                        "compute openings proof: commit phase + PoW", "compute openings proof: query rounds"], oracle_proof_2_20["stage_s"]))},
                  open(os.path.join(out, "cpu_oracle_full_size_64_threads.json"), "w"), indent=1)


@pytest.mark.parametrize("log_n", [17, 18, 19, 22])
def test_prove_openings_bit_exact(ctx, zkm, oracle, log_n):
    """FRI on the 3-pass plans (final polynomial 2^17 coefficients, LDE 2^19), on the digit coefficient layouts of 2^18 and 2^19 rows
    (openings from the exponent-indexed power table, combination converted to natural order; 2^20 is the bench proof's test) and
    BASELINE config 4 literally (13 + 4 + 4 polynomials of 2^22 coefficients, LDE 2^24, folds [4,4,4,4,4], 4 final coefficients):
    GPU bytes == oracle bytes."""
    rng = np.random.default_rng(4000 + log_n)
    n = 1 << log_n
    tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (13, 4, 4))
    old = oracle.get_threads()
    oracle.set_threads(min(64, os.cpu_count() or 1, __import__("bench").cpu_quota() or 64))   # (the GPU boxes grant 16 CPUs of the 256 they show)
    try:
        otb, oab, oqb = oracle.batch_from_values(tv, 13, log_n), oracle.batch_from_values(av, 4, log_n), oracle.batch_from_coeffs(qc, 4, log_n)
        want = oracle.prove_openings(otb, oab, oqb, 2)
    finally:
        oracle.set_threads(old)
    tb, ab = zkm.PolynomialBatch.from_values(ctx, tv, 13, log_n), zkm.PolynomialBatch.from_values(ctx, av, 4, log_n)
    qb = zkm.PolynomialBatch.from_coeffs(ctx, qc, 4, log_n)
    assert (tb.cap() == otb.cap()).all() and (qb.cap() == oqb.cap()).all()
    got = ctx.prove_openings(tb, ab, qb, 2)
    assert got.size == want.size
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d of %d" % (bad[0], got.size)
    for b in (tb, ab, qb):
        b.free()


def test_sha_witness_kernels_reference_vectors(ctx):
    """The SHA known answers the reference's own table tests hold (sha_compress_stark.rs:958-965,
    sha_compress_sponge_stark.rs:420-448, sha_extend_stark.rs:443-476) through the GPU witness kernels."""
    from .test_oracle_tables import REF_SHA_H, REF_SHA_W, REF_SHA_OUTPUT_HX, _le4
    meta = np.zeros((1, 8), dtype=np.uint64)
    tr = ctx.sha_compress_trace([REF_SHA_H], [REF_SHA_W], meta, 7).download().reshape(224, 128)
    assert _le4(tr, 140, 0) == 4228417613 and _le4(tr, 134, 0) == 2563236514
    tr = ctx.sha_compress_sponge_trace([REF_SHA_H], [REF_SHA_W], meta, 3).download().reshape(127, 8)
    assert [_le4(tr, 64 + 6 * q, 0) for q in range(8)] == REF_SHA_OUTPUT_HX
    inp = np.array([0, 1, 2, 3], dtype="<u4").view(np.uint8)
    tr = ctx.sha_extend_trace(inp, [0], 2).download().reshape(78, 4)
    assert _le4(tr, 0, 0) == 40965


def test_pipelined_host_ingest_equals_monolithic(zkm, oracle):
    """Host-resident values (the reference's Vec<PolynomialValues>, prover.rs:144-167) are uploaded in column chunks on a copy
    stream and absorbed chunk by chunk (k_merkle_leaves_chunk); the commitment must not depend on the chunking, the source
    memory kind (pageable / pinned) or the ragged last chunk."""
    rng = np.random.default_rng(77)
    log_n = 13
    for ncols in (64, 77, 262):
        vals = rand_field(rng, ncols << log_n)
        want = oracle.batch_from_values(vals, ncols, log_n)
        caps = []
        for chunk in (0, 8, 32):  # (log_n 13: the smallest height the pipelined path takes)
            c = zkm.Context(0)
            c.set_tuning("ingest_chunk_cols", chunk)
            b = zkm.PolynomialBatch.from_values(c, vals, ncols, log_n)
            assert (b.cap() == want.cap()).all() and (b.coeffs() == want.coeffs()).all(), (ncols, chunk)
            for i in (0, 5, (4 << log_n) - 1):
                assert (b.leaf(i) == want.leaf(i)).all() and (b.merkle_path(i) == want.merkle_path(i)).all()
            b.free()
            if chunk == 32:
                pinned = c.pinned_array(vals.size)
                pinned[:] = vals
                b = zkm.PolynomialBatch.from_values(c, pinned, ncols, log_n)
                assert (b.cap() == want.cap()).all()
                b.free()
            c.close()


@pytest.mark.parametrize("log_n", [9, 14])
def test_staged_traces_prove_like_host_and_device_traces(zkm, oracle, log_n):
    """zkm_trace_stage[_columns] (round 6: the NEXT proof's upload behind the CURRENT proof, include/zkm_hip.h "staged traces"): a proof
    from a staged trace -- one block or one pointer per column, pageable or pinned memory, words vouched canonical or not, consumed by
    zkm_prove_single_table or as one of K proofs of a lock-step call, the next trace being staged while the current proof runs --
    equals the proof from the host array and the oracle's, word for word; ready / free behave."""
    n = 1 << log_n
    c = zkm.Context(0)
    try:
        traces = []
        for k in range(3):
            d = c.poseidon_trace(seed=70 + k, num_perms=n - 2 - k, log_n=log_n)
            traces.append(d.download())
            d.free()
        aux = np.zeros(4 * n, dtype=np.uint64)
        want = [oracle.prove(t, log_n, aux, [1, 1]) for t in traces]
        for k in range(3):
            assert (c.prove_single_table(traces[k], log_n, aux, [1, 1]) == want[k]).all()
        # a loop as bench.py's host_resident block runs it: stage the next, prove the current, free it
        pinned = [c.pinned_array(262 * n) for _ in range(3)]
        for k in range(3):
            pinned[k][:] = traces[k]
        cur = c.stage_trace(pinned[0], 262, log_n)
        for k in range(3):
            nxt = c.stage_trace(pinned[k + 1], 262, log_n) if k < 2 else None
            got = c.prove_single_table(cur, log_n, aux, [1, 1])
            assert (got == want[k]).all(), k
            assert cur.ready() is True            # the proof consumed it: the upload has landed
            cur.free()
            cur = nxt
        # one pointer per column, pageable memory, NOT vouched canonical: every word + p where that still fits 64 bits
        cols = [np.ascontiguousarray(traces[1].reshape(262, n)[i]).copy() for i in range(262)]
        for col in cols[::7]:
            small = col < (1 << 32) - 1
            col[small] += np.uint64(P)
        st = c.stage_trace(cols, 262, log_n, canonical=False)
        assert st.ready(wait=True) is True
        assert (c.prove_single_table(st, log_n, aux, [1, 1]) == want[1]).all()
        # K proofs in lock-step from staged traces
        sts = [c.stage_trace(pinned[k], 262, log_n) for k in range(3)]
        got = c.prove_single_tables(sts, log_n, aux, [1, 1])
        for k in range(3):
            assert (got[k] == want[k]).all(), k
        for s_ in sts + [st]:
            s_.free()
        for p_ in pinned:
            c.free_pinned(p_)
        c.synchronize()
        live, _ = c.memory()
        assert live == c.resident_bytes()          # every staged block went back to the allocator
        with pytest.raises(zkm.ZkmError, match="zkm_trace_stage"):
            c.stage_trace(np.zeros(0, dtype=np.uint64), 0, log_n)
    finally:
        c.close()


def test_staged_trace_at_bench_size_equals_the_oracle(zkm, oracle_proof_2_20):
    """The deployed input at the bench's own size: the 262 x 2^20 trace of witness seed 100 staged from pinned HOST memory
    (zkm_trace_stage: 2.2 GB over PCIe on the copy streams) and proven -- the proof equals the oracle's 2^20-row proof word for word, as the
    device-resident and the in-proof-pipelined paths do (test_proof_is_bit_exact_2_20, test_pipelined_host_ingest_equals_monolithic)."""
    log_n = 20
    n = 1 << log_n
    c = zkm.Context(0)
    try:
        dev = c.poseidon_trace(100, n, log_n)
        host = c.pinned_array(262 * n)
        host[:] = dev.download()
        dev.free()
        aux = np.zeros(4 * n, dtype=np.uint64)
        st = c.stage_trace(host, 262, log_n)
        got = c.prove_single_table(st, log_n, aux, [1, 1])
        st.free()
        piped = c.prove_single_table(host, log_n, aux, [1, 1])          # the same bytes through the in-proof pipeline (chunk kernel on the matrix core)
        assert (got == piped).all()
        if oracle_proof_2_20 is not None:
            want = oracle_proof_2_20["proof"]
            bad = np.nonzero(got != want)[0] if got.size == want.size else np.array([-1])
            assert bad.size == 0, "first differing proof word: %d of %d" % (bad[0], want.size)
        c.free_pinned(host)
    finally:
        c.close()
