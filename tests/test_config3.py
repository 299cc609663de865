"""BASELINE config 3 on ONE GPU: "64 independent 2^20-row segments sharded across 8 MI355X (embarrassingly parallel)".

The reference proves segment files one after another (prover/examples/utils/src/utils.rs:57-68, 105-133); here the 64 segments
(PoseidonStark 262 x 2^20, witness seeds 100 .. 163 -- SURVEY 8d) go through zkm_amd.dist.prove_segments exactly as bench.py
--segments 64 drives them: round-robin over the ranks (one rank here), four contexts pulling from one queue.  Every proof must be
accepted by the oracle's verifier; segments 0, 31 and 63 must equal a proof made alone on a fresh context word for word (the
concurrent contexts share nothing but read-only traces); segment 0 must equal the CPU oracle's proof of the same trace.
The 8-GPU shape differs only in which rank's queue a segment lands in (tests/test_dist_cpu.py covers that under gloo).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LOG_N, NSEG, NCTX = 20, 64, 4


def test_config3_64_segments_four_contexts(zkm, oracle, oracle_proof_2_20):
    from zkm_amd import dist as zd
    n = 1 << LOG_N
    ctxs = [zkm.Context(0) for _ in range(NCTX)]
    try:
        # all 64 traces resident (64 x 2.05 GiB = 131 GiB of the 288): inputs are in HBM when the clock starts, as in bench.py
        traces = [ctxs[0].poseidon_trace(100 + s, n, LOG_N) for s in range(NSEG)]
        aux = ctxs[0].alloc(4 * n).upload(np.zeros(4 * n, dtype=np.uint64))
        ctxs[0].synchronize()

        def prove(s, w=0):
            return ctxs[w].prove_single_table(traces[s], LOG_N, aux, [1, 1])
        for w in range(NCTX):
            prove(w, w)                                           # warm-up: allocator, twiddles, power tables of every context
        proofs, elapsed = zd.prove_segments(prove, NSEG, sync_fn=lambda: [c.synchronize() for c in ctxs], gather=True, workers=NCTX)
        assert sorted(proofs) == list(range(NSEG))
        assert len({p.tobytes() for p in proofs.values()}) == NSEG          # 64 different traces -> 64 different proofs
        for s in range(NSEG):
            assert oracle.verify(proofs[s], 4, [1, 1]) == 0, "segment %d rejected by the oracle's verifier" % s
        solo = zkm.Context(0)
        for s in (0, 31, 63):
            alone = solo.prove_single_table(traces[s], LOG_N, aux, [1, 1])
            assert alone.size == proofs[s].size and (alone == proofs[s]).all(), "segment %d differs from its single-context proof" % s
        solo.close()
        if oracle_proof_2_20 is not None:                           # (None on hosts too small to run the oracle at 2^20 rows in time)
            want = oracle_proof_2_20["proof"]
            assert proofs[0].size == want.size and (proofs[0] == want).all(), "segment 0 (seed 100) differs from the CPU oracle's proof"
        rate = NSEG / elapsed
        print("config 3 on one MI355X: %d segments in %.2f s = %.2f proofs/s (%d contexts)" % (NSEG, elapsed, rate, NCTX))
        assert rate > 5.0                                           # sanity only (r02: 16 proofs/s); the number of record is bench.py's
    finally:
        for c in ctxs:
            c.close()
