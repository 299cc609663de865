"""Data formats either side of the proving path: field offsets of a proof blob and the ZKMTRACE segment image.
CPU tests use proofs made by the oracle; the GPU test proves from an image and compares with the in-memory path and the oracle."""
import numpy as np
import pytest

from tests.ctl_fixtures import build


def fake_aux(oracle, log_n, helpers=2):
    rng = np.random.default_rng(5)
    return rng.integers(0, 2**63, (helpers + 1) << log_n, dtype=np.uint64) % np.uint64(0xFFFFFFFF00000001)


def test_proof_layout_describes_an_oracle_proof(zkm, oracle):
    log_n = 5
    trace = oracle.poseidon_trace(3, 20, log_n)
    aux = fake_aux(oracle, log_n)
    proof, stages = oracle.prove(trace, log_n, aux, [2], want_stages=True)
    lay, q = zkm.proof_layout(proof)
    cfg = oracle.standard_config()
    assert lay.total_words == proof.size
    assert (lay.degree_bits, lay.trace_cols, lay.aux_cols, lay.ctl_zs) == (log_n, 262, 3, 1)
    assert lay.quotient_polys == 2 * cfg.num_challenges and lay.cap_height == cfg.cap_height and lay.num_queries == cfg.num_queries
    # recover_degree_bits (proof.rs:205-212)
    assert q.initial_siblings + lay.cap_height - lay.rate_bits == log_n
    # the caps at the reported offsets are the caps of the three committed batches
    tb = oracle.batch_from_values(trace, 262, log_n)
    cap = np.asarray(tb.cap()).reshape(-1)
    assert (proof[lay.trace_cap:lay.trace_cap + cap.size] == cap).all()
    ab = oracle.batch_from_values(aux, 3, log_n)
    cap = np.asarray(ab.cap()).reshape(-1)
    assert (proof[lay.aux_cap:lay.aux_cap + cap.size] == cap).all()
    # first query round: the trace leaf and its Merkle path authenticate against the trace cap
    base = lay.query_round_proofs
    leaf = proof[base + q.oracle_evals[0]: base + q.oracle_evals[0] + 262]
    sib = proof[base + q.oracle_siblings[0]: base + q.oracle_siblings[0] + 4 * q.initial_siblings].reshape(-1, 4)
    found = False
    lde = 1 << (log_n + int(lay.rate_bits))
    for i in range(lde):
        if (np.asarray(tb.leaf(i)) == leaf).all():
            assert (np.asarray(tb.merkle_path(i)).reshape(-1, 4) == sib).all()
            found = True
            break
    assert found
    # the fields tile the blob without gaps, in order
    order = [16, lay.init_challenger_state, lay.trace_cap, lay.aux_cap, lay.quotient_cap, lay.local_values, lay.next_values, lay.aux_polys,
             lay.aux_polys_next, lay.ctl_zs_first, lay.quotient_polys_open, lay.commit_phase_merkle_caps, lay.final_poly, lay.pow_witness,
             lay.query_round_proofs]
    assert order == sorted(order) and order[0] == order[1]
    assert lay.query_round_proofs + lay.num_queries * lay.query_round_words == proof.size
    assert lay.pow_witness + 1 == lay.query_round_proofs
    with pytest.raises(zkm.ZkmError):
        zkm.proof_layout(np.zeros(64, dtype=np.uint64))


def parse_image(img):
    assert img[0] == int.from_bytes(b"ZKMTRACE", "little") and img[1] == 1
    nt, npub = int(img[2]), int(img[3])
    pub = img[8:8 + npub]
    recs = img[8 + npub: 8 + npub + 8 * nt].reshape(nt, 8)
    return pub, recs


def test_segment_image_round_trip_and_sizing(zkm, oracle):
    tables, ctls = build(oracle)
    pub = [9, 8, 7]
    img = zkm.segment_image(tables, ctls, public_values=pub)
    got_pub, recs = parse_image(img)
    assert list(got_pub) == pub
    for (tid, tr, ncols, log_n, ct), r in zip(tables, recs):
        assert (int(r[0]), int(r[1]), int(r[2])) == (tid, ncols, log_n)
        off = int(r[3])
        assert (img[off: off + (ncols << log_n)] == np.asarray(tr).reshape(-1)).all()
    # sizing from the image alone (no GPU) agrees with the in-memory sizing
    import ctypes as C
    lib = zkm.load()
    cfg = zkm.StarkConfig()
    lib.zkm_standard_config(C.byref(cfg))
    offs = (C.c_size_t * (len(tables) + 1))()
    total = C.c_size_t()
    err = C.c_char_p()
    u64p = C.POINTER(C.c_uint64)
    assert lib.zkm_prove_segment_image(None, C.byref(cfg), img.ctypes.data_as(u64p), img.size, None, C.byref(total), offs, None, C.byref(err)) == 0
    wtotal, woffs = oracle.all_proof_words(tables, ctls)
    assert total.value == wtotal and list(offs) == list(woffs)
    # truncated or foreign images are rejected with a message
    for bad, n in ((img, img.size - 1), (img, 12), (np.zeros(32, dtype=np.uint64), 32)):
        err = C.c_char_p()
        assert lib.zkm_prove_segment_image(None, C.byref(cfg), bad.ctypes.data_as(u64p), n, None, C.byref(total), offs, None, C.byref(err)) != 0
        assert b"zkm_prove_segment_image" in err.value


def test_segment_image_hostile_headers_are_rejected(zkm, oracle):
    """every size in the image header is untrusted.  Values chosen so that the sums the old checks formed
    (o + npub + 8 ntables, 3 ncolumns + nterms + ..., ncols << log_n, looking_off + nlooking) wrap around 2^64."""
    import ctypes as C
    tables, ctls = build(oracle)
    img = zkm.segment_image(tables, ctls, public_values=[1, 2])
    lib = zkm.load()
    cfg = zkm.StarkConfig()
    lib.zkm_standard_config(C.byref(cfg))
    u64p = C.POINTER(C.c_uint64)
    nt, npub, nctls = int(img[2]), int(img[3]), int(img[4])
    th = 8 + npub                       # first table record
    M = (1 << 64) - 1

    def rejected(edit):
        bad = img.copy()
        edit(bad)
        offs = (C.c_size_t * (nt + 1))()
        total = C.c_size_t()
        err = C.c_char_p()
        rc = lib.zkm_prove_segment_image(None, C.byref(cfg), bad.ctypes.data_as(u64p), bad.size, None, C.byref(total), offs, None, C.byref(err))
        return rc != 0 and b"zkm_prove_segment_image" in (err.value or b"")

    def setw(i, v):
        def f(a):
            a[i] = np.uint64(v)
        return f
    assert rejected(setw(3, M - 15))                    # npub: o + npub + 8 ntables wraps to a small number
    assert rejected(setw(3, 1 << 63))
    assert rejected(setw(2, 4097))                      # ntables over the limit
    assert rejected(setw(4, M))                         # nctls
    assert rejected(setw(5, M - 3))                     # nsides: o + 2 nctls + nsides wraps
    for f, v in ((4, (M // 3) + 2), (4, M), (5, M), (5, M - 1), (6, (1 << 62) + 1), (7, M), (7, M - 1)):
        assert rejected(setw(th + f, v)), (f, v)        # ncolumns / nterms / ncolsets / nfilter_idx: the `need` sum wraps
    assert rejected(setw(th + 1, 1 << 60))              # ncols so large that ncols << log_n wraps
    assert rejected(setw(th + 1, 0))
    assert rejected(setw(th + 2, 41))                   # log_n
    assert rejected(setw(th + 2, 64))
    assert rejected(setw(th + 3, M))                    # trace offset
    assert rejected(setw(th + 3, img.size))
    # looking_off + nlooking wrapping in 32 bits / pointing past the sides array
    ctl0 = None
    o = 8 + npub + 8 * nt
    for t in range(nt):
        h = img[th + 8 * t: th + 8 * t + 8]
        o += 3 * int(h[4]) + (int(h[5]) + 1) // 2 + int(h[5]) + 4 * int(h[6]) + (int(h[7]) + 1) // 2
    ctl0 = o
    assert nctls > 0
    assert rejected(setw(ctl0, (0xFFFFFFFF << 32) | 2))          # looking_off = 2^32 - 1, nlooking = 2
    assert rejected(setw(ctl0, (1 << 32) | 0xFFFFFFFF))          # nlooking = 2^32 - 1
    assert not rejected(lambda a: None)                          # the untouched image still parses


@pytest.mark.gpu
def test_prove_from_image_equals_in_memory_path(ctx, zkm, oracle):
    tables, ctls = build(oracle)
    pub = [9, 8, 7]
    img = zkm.segment_image(tables, ctls, public_values=pub)
    got, chal, offs = ctx.prove_segment_image(img)
    want, wchal, woffs = ctx.prove_with_traces(tables, ctls, public_values=pub)
    assert offs == woffs and (chal == wchal).all() and (got == want).all()
    ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=pub)
    assert (got == ref).all()
    assert oracle.verify_all(tables, ctls, got, chal, public_values=pub) == 0
    for k in range(len(tables)):
        lay, _ = zkm.proof_layout(got[offs[k]:offs[k + 1]])
        assert lay.total_words == offs[k + 1] - offs[k] and lay.degree_bits == tables[k][3]


@pytest.mark.gpu
def test_prove_segment_uses_the_shipped_all_stark(ctx, zkm, oracle):
    """zkm_prove_segment proves the twelve tables from their traces alone (the AllStark lookups are compiled into the library,
    csrc/all_stark_ctl.inc): same bytes as zkm_prove_with_traces driven by the Python description, and as the oracle."""
    import os
    from zkm_amd import tables as T
    seg = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    traces = [seg["t%d" % i] for i in range(12)]
    got, chal, offs = ctx.prove_segment(traces, log_n, public_values=[1, 2, 3])
    ctl_tables, ctls = T.all_cross_table_lookups()
    tables = [(T.TABLE_ENUM_ORDER[i], traces[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    want, wchal, woffs = ctx.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    assert offs == woffs and (chal == wchal).all() and (got == want).all()
    ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    assert (got == ref).all() and (chal == rchal).all()
    # a full segment in any other table order is refused (the transcript would not match the reference's)
    swapped = [tables[1], tables[0]] + tables[2:]
    with pytest.raises(zkm.ZkmError, match="Table enum order"):
        ctx.prove_with_traces(swapped, ctls, public_values=[1, 2, 3])


@pytest.mark.gpu
def test_segment_at_2_16_cycle_heights_is_bit_exact(ctx, zkm, oracle):
    """The reference's default segment size (2^16 cycles, emulator/src/utils.rs:6): the committed twelve-table test segment tiled to the
    table heights tools/bench_segment.py times (CPU and Arithmetic 2^16, Memory 2^17, precompile tables 2^6 .. 2^13), proven by
    zkm_prove_segment and by the oracle's prove_with_traces -- all twelve proof blobs and the CTL challenges word for word.  (Tiled
    rows are not a valid witness across the seams; neither prover looks at validity.)"""
    import os
    from zkm_amd import tables as T
    from tools.bench_segment import HEIGHTS
    seg = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment12.npz"))
    base = [int(x) for x in seg["log_n"]]
    traces, log_n = [], []
    for i in range(12):
        w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
        L = max(HEIGHTS[16][i], base[i])
        traces.append(np.ascontiguousarray(np.tile(seg["t%d" % i].reshape(w, -1), (1, 1 << (L - base[i])))).reshape(-1))
        log_n.append(L)
    got, chal, offs = ctx.prove_segment(traces, log_n, public_values=[1, 2, 3])
    ctl_tables, ctls = T.all_cross_table_lookups()
    tables = [(T.TABLE_ENUM_ORDER[i], traces[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    old = oracle.get_threads()
    oracle.set_threads(min(64, os.cpu_count() or 1, __import__("bench").cpu_quota() or 64))   # (the GPU boxes grant 16 CPUs of the 256 they show)
    try:
        ref, rchal, roffs = oracle.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    finally:
        oracle.set_threads(old)
    assert list(offs) == list(roffs) and (chal == rchal).all()
    bad = np.nonzero(got != ref)[0]
    assert bad.size == 0, "first differing word %d (table %d)" % (bad[0], int(np.searchsorted(offs, bad[0], side="right")) - 1)
    # the throughput profile (one stream per context, latency forms of the permutation only for tiny launches: at these heights the
    # leaves of the 2^13 / 2^14-row matrices move from the quad to the one-lane form, those of 2^10 rows from 16 lanes to a quad, and
    # every commitment runs on the context's own stream) must give the same words
    c2 = zkm.Context(0)
    try:
        c2.set_tuning("throughput_profile", 1)
        got2, chal2, offs2 = c2.prove_segment(traces, log_n, public_values=[1, 2, 3])
        assert list(offs2) == list(offs) and (chal2 == chal).all() and (got2 == got).all()
        c2.set_tuning("throughput_profile", 0)
        c2.set_tuning("commit_lanes", 2)
        c2.set_tuning("small_ntt", 0)          # ... and the two-pass transforms / level kernels + download instead of round 4's
        c2.set_tuning("tree_tail", 0)          # single-launch paths for short tables
        c2.set_tuning("aux_pipeline", 0)       # all auxiliary commitments before the first table proof (the default builds tables 1..'s
        got3, _, _ = c2.prove_segment(traces, log_n, public_values=[1, 2, 3])    # behind the proofs of the earlier tables)
        assert (got3 == got).all()
        c2.set_tuning("commit_lanes", 8)       # ... and the pipelined form with seven lanes dealing the tables differently
        c2.set_tuning("aux_pipeline", 1)
        got4, _, _ = c2.prove_segment(traces, log_n, public_values=[1, 2, 3])
        assert (got4 == got).all()
    finally:
        c2.close()


@pytest.mark.gpu
def test_pure_c_caller_proves_the_segment_image(ctx, zkm, oracle, tmp_path):
    """tools/c_smoke.c (pedantic C99, linked against libzkmhip.so, no Python in the process) proves the twelve-table test segment from a
    ZKMTRACE file: its proof blobs equal the ctypes path's and the oracle's word for word."""
    import os
    import subprocess
    from zkm_amd import tables as T
    from tests.test_abi import build_c_smoke
    exe = build_c_smoke(tmp_path)
    seg = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    ctl_tables, ctls = T.all_cross_table_lookups()
    tables = [(T.TABLE_ENUM_ORDER[i], seg["t%d" % i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    img = zkm.segment_image(tables, ctls, public_values=[1, 2, 3])
    path = tmp_path / "segment.zkmtrace"
    img.tofile(path)
    out = tmp_path / "proofs.bin"
    r = subprocess.run([exe, str(path), str(out)], capture_output=True, text=True, timeout=300, env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=np.uint64)
    want, chal, offs = ctx.prove_segment_image(img)
    assert got.size == want.size and (got == want).all()
    ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    assert (got == ref).all()
    assert "ok %d words, 12 tables, beta0 %016x" % (want.size, int(chal[0])) in r.stdout
    assert "lockstep ok: 2 segments" in r.stdout      # zkm_prove_segments from plain C: both blobs == the single-segment proof
    assert "pool ok: 3 segments, 2 workers on device 0" in r.stdout   # zkm_pool_* from plain C: groups 2 + 1 on two workers, same words


@pytest.mark.gpu
def test_out_of_memory_retry_while_lanes_are_active(zkm):
    """the allocator's out-of-memory path trims caches -- its own, then the parent's and the sibling lanes' -- while the
    commit lanes of the same segment are allocating and releasing on their own threads.  The "debug_fail_allocs" hook makes device
    allocations fail on their first attempt (every second one also on the retry, which sends it into the family path); the proofs of
    such a segment must equal those of an undisturbed one, repeatedly, and the memory accounting must come back to zero."""
    import os
    seg = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    traces = [seg["t%d" % i] for i in range(12)]
    c = zkm.Context(0)
    os.environ.pop("ZKM_ENABLE_TEST_HOOKS", None)
    with pytest.raises(zkm.ZkmError, match="unknown key"):     # the hook is not a tuning: refused unless the process asked for test hooks
        c.set_tuning("debug_fail_allocs", 1)
    os.environ["ZKM_ENABLE_TEST_HOOKS"] = "1"
    try:
        want, wchal, woffs = c.prove_segment(traces, log_n, public_values=[1, 2, 3])     # also fills every cache
        for rounds, k in enumerate((40, 400, 100000)):
            c.set_tuning("debug_fail_allocs", k)
            got, chal, offs = c.prove_segment(traces, log_n, public_values=[1, 2, 3])
            assert offs == woffs and (chal == wchal).all() and (got == want).all(), (rounds, k)
        c.set_tuning("debug_fail_allocs", 0)
        got, chal, offs = c.prove_segment(traces, log_n, public_values=[1, 2, 3])
        assert (got == want).all()
        live, cached = c.memory()
        assert live == c.resident_bytes()
    finally:
        os.environ.pop("ZKM_ENABLE_TEST_HOOKS", None)
        c.close()


@pytest.mark.gpu
def test_segments_proven_side_by_side_are_bit_exact(ctx, zkm):
    """Twelve-table segments from three contexts working at the same time (host traces: each call uploads, commits and proves)
    equal the segment proof of one context working alone."""
    import os
    from zkm_amd.dist import run_workers
    seg = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    traces = [seg["t%d" % i] for i in range(12)]
    want, wchal, woffs = ctx.prove_segment(traces, log_n, public_values=[1, 2, 3])
    workers = [zkm.Context(0) for _ in range(3)]
    try:
        got = run_workers(lambda s, w: workers[w].prove_segment(traces, log_n, public_values=[1, 2, 3]), range(6), len(workers))
    finally:
        for w in workers:
            w.close()
    for s in range(6):
        proofs, chal, offs = got[s]
        assert offs == woffs and (chal == wchal).all() and (proofs == want).all(), s


@pytest.mark.gpu
def test_gpu_proof_blob_walks_through_the_layout(ctx, zkm, oracle):
    """N4 on a GPU-made proof: every field zkm_proof_get_layout / _query_layout report is where the batches say it is -- caps equal
    the commitments' caps, every query round's leaves and Merkle paths authenticate against them, FRI layer evals chain."""
    log_n = 8
    n = 1 << log_n
    trace = oracle.poseidon_trace(21, n - 9, log_n)
    aux = np.zeros(3 * n, dtype=np.uint64)          # the benchmark's fake CTL shape: two helper columns and a Z, all zero (valid)
    tb = zkm.PolynomialBatch.from_values(ctx, trace, 262, log_n)
    ab = zkm.PolynomialBatch.from_values(ctx, aux, 3, log_n)
    proof = ctx.prove_single_table(None, log_n, aux, [2], trace_batch=tb)
    assert oracle.verify(proof, 3, [2]) == 0
    lay, q = zkm.proof_layout(proof)
    assert lay.total_words == proof.size and (lay.degree_bits, lay.trace_cols, lay.aux_cols, lay.ctl_zs) == (log_n, 262, 3, 1)
    C4 = 4 << int(lay.cap_height)
    assert (proof[lay.trace_cap:lay.trace_cap + C4] == tb.cap().reshape(-1)).all()
    assert (proof[lay.aux_cap:lay.aux_cap + C4] == ab.cap().reshape(-1)).all()
    lde_bits = log_n + int(lay.rate_bits)
    N = 1 << lde_bits
    rows_t = {tuple(int(x) for x in tb.leaf(i)[:4]): i for i in range(N)}
    for r in range(int(lay.num_queries)):
        base = lay.query_round_proofs + r * lay.query_round_words
        leaf = proof[base + q.oracle_evals[0]: base + q.oracle_evals[0] + 262]
        x = rows_t[tuple(int(v) for v in leaf[:4])]                      # the queried leaf index, recovered from the data
        assert (tb.leaf(x) == leaf).all()
        sib = proof[base + q.oracle_siblings[0]: base + q.oracle_siblings[0] + 4 * q.initial_siblings]
        assert (tb.merkle_path(x).reshape(-1) == sib).all()
        aleaf = proof[base + q.oracle_evals[1]: base + q.oracle_evals[1] + 3]
        assert (ab.leaf(x) == aleaf).all()
        asib = proof[base + q.oracle_siblings[1]: base + q.oracle_siblings[1] + 4 * q.initial_siblings]
        assert (ab.merkle_path(x).reshape(-1) == asib).all()
        assert q.oracle_cols[2] == lay.quotient_polys and q.layer_siblings_count[0] == lde_bits - int(lay.arity_bits) - int(lay.cap_height)
    # bulk LDE accessor == row-at-a-time accessor (get_lde_values_packed, prover.rs:687)
    rows = tb.lde_rows(5, 2, 7)
    for i in range(7):
        assert (rows[i] == tb.lde_row((5 + i) * 2)).all()
    # out-of-range and wrapping arguments are refused, not wrapped around
    N = 1 << lde_bits
    assert (tb.lde_rows(N // 2 - 3, 2, 3)[2] == tb.lde_row(N - 2)).all()
    for start, step, count in ((N // 2 - 2, 2, 3), (N, 1, 1), (0, 0, 1), ((1 << 64) - 1, 1, 2), (1, 1, (1 << 64) - 1)):
        assert tb.ctx.L.zkm_batch_lde_rows(tb.h, start, step, count, None) != 0, (start, step, count)
    tb.free()
    ab.free()


P = 0xFFFFFFFF00000001


def _split_columns(flat, ncols, rng=None):
    """One separately allocated array per column (the reference's Vec<PolynomialValues<F>>); with rng, words below 2^32 - 1 are
    replaced by their non-canonical representative v + p here and there (a plonky2 GoldilocksField may hold either)."""
    cols = [np.array(c, dtype=np.uint64) for c in np.asarray(flat, dtype=np.uint64).reshape(ncols, -1)]
    if rng is not None:
        for c in cols:
            pick = (c < np.uint64(0xFFFFFFFF)) & (rng.integers(0, 3, c.size) == 0)
            c[pick] += np.uint64(P)
    return cols


def test_segment_image_from_column_pointers(zkm, oracle):
    """zkm_table_input.columns: the image written from one pointer per column equals the image written from the flat block."""
    from zkm_amd import ctl as zc
    import ctypes as C
    tables, ctls = build(oracle)
    img = zkm.segment_image(tables, ctls, public_values=[4, 5])
    lib = zkm.load()
    keep = [_split_columns(tr, ncols) for (_, tr, ncols, _, _) in tables]
    packed = [(tid, [c.ctypes.data for c in cols], ncols, log_n, ct) for (tid, _, ncols, log_n, ct), cols in zip(tables, keep)]
    tarr, keep2 = zc.pack_tables(packed)
    carr, sides = zc.pack_ctls(ctls)
    pub = np.array([4, 5], dtype=np.uint64)
    words = lib.zkm_segment_image_words(tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), pub.size)
    assert words == img.size
    out = np.zeros(words, dtype=np.uint64)
    err = C.c_char_p()
    u64p = C.POINTER(C.c_uint64)
    assert lib.zkm_segment_image_write(tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), pub.ctypes.data_as(u64p), pub.size,
                                       out.ctypes.data_as(u64p), C.byref(err)) == 0, err.value
    assert (out == img).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ncols,log_n", [(5, 9), (77, 13), (262, 13)])
def test_commit_columns_equals_commit_values(ctx, zkm, oracle, ncols, log_n):
    """zkm_batch_commit_columns (one pointer per column): same coefficients, cap, leaves and paths as the flat
    from_values / from_coeffs -- through the monolithic upload (5 x 2^9) and the pipelined column-chunk ingest (>= 64 columns x 2^13)
    -- also when some words are the non-canonical representative v + p of their field element."""
    rng = np.random.default_rng(600 + ncols)
    vals = rng.integers(0, P, ncols << log_n, dtype=np.uint64)
    vals[::7] = rng.integers(0, 1 << 32, vals[::7].size, dtype=np.uint64)
    want = oracle.batch_from_values(vals, ncols, log_n)
    for noncanon in (None, rng):
        cols = _split_columns(vals, ncols, noncanon)
        b = zkm.PolynomialBatch.from_columns(ctx, cols, log_n)
        assert (b.cap() == want.cap()).all() and (b.coeffs() == want.coeffs()).all()
        for i in (0, 3, (4 << log_n) - 1):
            assert (b.leaf(i) == want.leaf(i)).all() and (b.merkle_path(i) == want.merkle_path(i)).all()
        b.free()
    wc = oracle.batch_from_coeffs(vals, ncols, log_n)
    b = zkm.PolynomialBatch.from_columns(ctx, _split_columns(vals, ncols), log_n, values=False)
    assert (b.cap() == wc.cap()).all() and (b.coeffs() == vals).all()
    b.free()
    with pytest.raises(zkm.ZkmError, match="null"):
        import ctypes as C
        h, err = C.c_void_p(), C.c_char_p()
        ptrs = (C.c_void_p * 2)(None, None)
        rc = ctx.L.zkm_batch_commit_columns(ctx.h, ptrs, 2, 4, 1, 2, 4, C.byref(h), C.byref(err))
        assert rc != 0
        raise zkm.ZkmError(err.value.decode())


@pytest.mark.gpu
def test_prove_segment_from_column_pointers(ctx, zkm):
    """zkm_prove_segment_columns: the twelve traces as one pointer per column (some words non-canonical) give the proof of the flat,
    canonical traces word for word -- commitment, CTL data and lookup columns all read the same field elements."""
    import os
    from zkm_amd import tables as T
    seg = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    traces = [seg["t%d" % i] for i in range(12)]
    want, wchal, woffs = ctx.prove_segment(traces, log_n, public_values=[1, 2, 3])
    rng = np.random.default_rng(12)
    cols = [_split_columns(traces[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], rng) for i in range(12)]
    got, chal, offs = ctx.prove_segment(cols, log_n, public_values=[1, 2, 3])
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    # the generic entry point with zkm_table_input.columns
    ctl_tables, ctls = T.all_cross_table_lookups()
    tables = [(T.TABLE_ENUM_ORDER[i], cols[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    got2, chal2, offs2 = ctx.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    assert offs2 == woffs and (got2 == want).all()
