"""CPU suite: NTT / commitment / prove -> verify of the oracle (sizes that finish in seconds)."""
import numpy as np
import pytest

P = 0xFFFFFFFF00000001
G = 14293326489335486720


def naive_dft(x, root):
    n = len(x)
    return [sum(int(x[i]) * pow(root, i * k, P) for i in range(n)) % P for k in range(n)]


def test_ntt_matches_naive_dft(oracle):
    rng = np.random.default_rng(3)
    for log_n in (1, 3, 5):
        n = 1 << log_n
        x = rng.integers(0, P, n, dtype=np.uint64)
        w = oracle.root_of_unity(log_n)
        assert [int(v) for v in oracle.ntt(x, log_n)] == naive_dft(x, w)
        # coset: coefficients scaled by shift^i first (plonky2 coset_fft)
        xs = np.array([int(x[i]) * pow(G, i, P) % P for i in range(n)], dtype=np.uint64)
        assert [int(v) for v in oracle.ntt(x, log_n, coset_shift=G)] == naive_dft(xs, w)


def test_ntt_roundtrip_and_linearity(oracle):
    rng = np.random.default_rng(4)
    log_n = 10
    x = rng.integers(0, P, 3 << log_n, dtype=np.uint64)
    y = oracle.ntt(x, log_n)
    assert (oracle.ntt(y, log_n, inverse=True) == x).all()
    yc = oracle.ntt(x, log_n, coset_shift=G)
    assert (oracle.ntt(yc, log_n, inverse=True, coset_shift=G) == x).all()


def test_batch_lde_is_evaluation_on_coset(oracle):
    rng = np.random.default_rng(5)
    log_n, ncols = 4, 3
    n, N = 1 << log_n, 4 << log_n
    vals = rng.integers(0, P, ncols * n, dtype=np.uint64)
    b = oracle.batch_from_values(vals, ncols, log_n)
    co = b.coeffs().reshape(ncols, n)
    w4 = oracle.root_of_unity(log_n + 2)
    for i in (0, 1, 7, N - 1):
        x = G * pow(w4, i, P) % P
        want = [sum(int(co[c][k]) * pow(x, k, P) for k in range(n)) % P for c in range(ncols)]
        assert [int(v) for v in b.lde_row(i)] == want
    # values are the evaluations on the subgroup: every 4th LDE point of the *unshifted* domain is not
    # available, but iNTT(values) == coeffs
    assert (oracle.ntt(vals, log_n, inverse=True) == co.reshape(-1)).all()
    # Merkle path of a leaf recomputes to the cap
    for leaf in (0, 5, N - 1):
        d = oracle.hash_or_noop(b.leaf(leaf))
        path = b.merkle_path(leaf).reshape(-1, 4)
        idx = leaf
        for s in path:
            d = oracle.two_to_one(s, d) if idx & 1 else oracle.two_to_one(d, s)
            idx >>= 1
        assert (b.cap().reshape(-1, 4)[idx] == d).all()


@pytest.mark.parametrize("log_n", [5, 6, 7])
def test_prove_then_verify(oracle, log_n):
    # the shape of poseidon_benchmark (poseidon_stark.rs:751-816): fake CTL data = zero helper + zero Z, x2
    n = 1 << log_n
    trace = oracle.poseidon_trace(seed=1, num_perms=n - 3, log_n=log_n)
    aux = np.zeros(4 * n, dtype=np.uint64)
    proof = oracle.prove(trace, log_n, aux, [1, 1])
    assert oracle.verify(proof, 4, [1, 1]) == 0
    # any flipped word must be rejected
    rng = np.random.default_rng(log_n)
    for pos in rng.integers(16, proof.size, 12):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert oracle.verify(bad, 4, [1, 1]) != 0, "tampered word %d accepted" % pos


def test_invalid_witness_fails_verification(oracle):
    log_n = 5
    n = 1 << log_n
    trace = oracle.poseidon_trace(seed=2, num_perms=n, log_n=log_n)
    trace = trace.copy()
    trace[40 * n + 3] = (int(trace[40 * n + 3]) + 1) % P  # break one s-box witness cell
    aux = np.zeros(4 * n, dtype=np.uint64)
    proof = oracle.prove(trace, log_n, aux, [1, 1])  # the reference would panic in trim_to_len (prover.rs:566-570)
    assert oracle.verify(proof, 4, [1, 1]) != 0


def test_transcript_chaining(oracle):
    # one Challenger threads through tables (prover.rs:240-419): a proof started from a non-empty transcript
    # verifies only from the same transcript state
    from oracle.oracle_py import Challenger
    log_n = 5
    n = 1 << log_n
    trace = oracle.poseidon_trace(seed=3, num_perms=n, log_n=log_n)
    aux = np.zeros(4 * n, dtype=np.uint64)
    ch = oracle.challenger()
    oracle.observe(ch, [1, 2, 3])
    proof = oracle.prove(trace, log_n, aux, [1, 1], challenger=ch)
    ch2 = oracle.challenger()
    oracle.observe(ch2, [1, 2, 3])
    assert oracle.verify(proof, 4, [1, 1], challenger=ch2) == 0
    assert oracle.verify(proof, 4, [1, 1], challenger=oracle.challenger()) != 0


def test_prove_openings_alone_then_verify(oracle):
    # BASELINE config 4 shape at a small size: 13 trace columns (MemoryStark width), 4 aux (2 CTL Zs), 4 quotient chunks of
    # seeded random polynomials; prove_openings on the three commitments, FRI-only verification
    rng = np.random.default_rng(44)
    for log_n in (5, 8):
        n = 1 << log_n
        tb = oracle.batch_from_values(rng.integers(0, P, 13 * n, dtype=np.uint64), 13, log_n)
        ab = oracle.batch_from_values(rng.integers(0, P, 4 * n, dtype=np.uint64), 4, log_n)
        qb = oracle.batch_from_coeffs(rng.integers(0, P, 4 * n, dtype=np.uint64), 4, log_n)
        proof = oracle.prove_openings(tb, ab, qb, 2)
        assert oracle.verify_openings(proof, 13, 4, 2) == 0
        # (the caps are not part of this truncated transcript, so a cap entry no query lands in is unconstrained here;
        #  in prove_single_table every cap is observed.  Tamper from the openings onwards.)
        for pos in rng.integers(16 + 12 + 3 * 64, proof.size, 8):
            bad = proof.copy()
            bad[pos] = (int(bad[pos]) + 1) % P
            assert oracle.verify_openings(bad, 13, 4, 2) != 0
