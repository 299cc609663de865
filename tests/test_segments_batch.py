"""zkm_prove_segments: K independent segments in lock-step (one launch per stage for all segments whose table has the same height).
Every proof blob and every CTL challenge must be word for word what the single-segment path (zkm_prove_segment) and the CPU oracle's
prove_with_traces (restating prover/src/prover.rs:130-438) produce for that segment alone."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _segment(variant, heights=None):
    """The committed twelve-table test segment, rows rotated by `variant` (filters stay binary: neither prover looks at validity across
    the seam) and tiled to `heights` (log2 rows per table) where given: (traces, log_n)."""
    from zkm_amd import tables as T
    seg = np.load(os.path.join(ROOT, "tests", "golden", "segment12.npz"))
    base = [int(x) for x in seg["log_n"]]
    traces, log_n = [], []
    for i in range(12):
        w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
        t = seg["t%d" % i].reshape(w, -1)
        if variant:
            t = np.roll(t, variant * (i + 1), axis=1)
        L = base[i] if heights is None else max(heights[i], base[i])
        traces.append(np.ascontiguousarray(np.tile(t, (1, 1 << (L - base[i])))).reshape(-1))
        log_n.append(L)
    return traces, log_n


@pytest.mark.gpu
def test_lockstep_segments_equal_single_segment_proofs_and_oracle(ctx, zkm, oracle):
    """Four segments of the base heights (one group of four per table): lock-step == one at a time == oracle."""
    from zkm_amd import tables as T
    segs = []
    for v in range(4):
        tr, lg = _segment(v)
        segs.append((tr, lg, [1 + v, 2, 3 + 7 * v]))
    got = ctx.prove_segments(segs)
    ctl_tables, ctls = T.all_cross_table_lookups()
    for v, (tr, lg, pub) in enumerate(segs):
        one, chal1, offs1 = ctx.prove_segment(tr, lg, public_values=pub)
        proofs, chal, offs = got[v]
        assert list(offs) == list(offs1) and (chal == chal1).all()
        bad = np.nonzero(proofs != one)[0]
        assert bad.size == 0, "segment %d: first differing word %d (table %d)" % (v, bad[0], int(np.searchsorted(offs, bad[0], side="right")) - 1)
        tables = [(T.TABLE_ENUM_ORDER[i], tr[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], lg[i], ctl_tables[i]) for i in range(12)]
        ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=pub)
        assert (proofs == ref).all() and (chal == rchal).all()
        if v == 0:      # (the rotated variants are not valid witnesses across the seam: the provers do not care, the verifier does)
            assert oracle.verify_all(tables, ctls, proofs, chal, public_values=pub) == 0


@pytest.mark.gpu
def test_lockstep_segments_from_column_pointers(ctx, zkm):
    """zkm_prove_segments_columns: every table of every segment as one pointer per column (K of the reference's
    [Vec<PolynomialValues<F>>; NUM_TABLES], prover.rs:130-142): the same words as the block form."""
    from zkm_amd import tables as T
    segs, col_segs = [], []
    for v in range(3):
        tr, lg = _segment(v)
        segs.append((tr, lg, [9, v]))
        cols = []
        for i in range(12):
            w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
            m = tr[i].reshape(w, -1)
            cols.append([np.ascontiguousarray(m[k]) for k in range(w)])
        col_segs.append((cols, lg, [9, v]))
    want = ctx.prove_segments(segs)
    got = ctx.prove_segments(col_segs)
    for v in range(3):
        assert list(got[v][2]) == list(want[v][2]) and (got[v][1] == want[v][1]).all() and (got[v][0] == want[v][0]).all()


@pytest.mark.gpu
def test_lockstep_segments_of_ragged_heights_and_small_stacks(ctx, zkm):
    """Segments whose tables differ in height form one group per height; max_stack splits groups; every stacking gives the words
    of the single-segment path."""
    base = _segment(0)[1]
    hs = [list(base), [h + 1 for h in base], list(base), [h + (i % 2) for i, h in enumerate(base)], [h + 1 for h in base]]
    segs = []
    for v, h in enumerate(hs):
        tr, lg = _segment(v, h)
        segs.append((tr, lg, [v, v + 1]))
    want = [ctx.prove_segment(tr, lg, public_values=pub) for tr, lg, pub in segs]
    c2 = zkm.Context(0)
    try:
        for stack in (32, 2, 3):
            c2.set_tuning("max_stack", stack)
            got = c2.prove_segments(segs)
            for v in range(len(segs)):
                assert list(got[v][2]) == list(want[v][2]) and (got[v][1] == want[v][1]).all()
                bad = np.nonzero(got[v][0] != want[v][0])[0]
                assert bad.size == 0, "max_stack %d, segment %d: first differing word %d" % (stack, v, bad[0])
        c2.set_tuning("max_stack", 32)
        c2.set_tuning("throughput_profile", 1)        # one stream, latency forms for tiny launches only
        got = c2.prove_segments(segs)
        for v in range(len(segs)):
            assert (got[v][0] == want[v][0]).all()
        c2.set_tuning("throughput_profile", 0)
        c2.set_tuning("small_ntt", 0)
        c2.set_tuning("tree_tail", 0)
        c2.set_tuning("aux_pipeline", 0)
        got = c2.prove_segments(segs[:3])
        for v in range(3):
            assert (got[v][0] == want[v][0]).all()
    finally:
        c2.close()


@pytest.mark.gpu
def test_lockstep_segments_from_device_buffers_at_2_16_cycle_heights(ctx, zkm, oracle):
    """Three segments at the table heights of a 2^16-cycle segment (tools/bench_segment.py: the shape bench.py's segment_2_16.lockstep
    times), traces resident in HBM: every table's group of three == the single-segment path, and -- VERDICT r05 #7 -- EVERY segment of the
    lock-step call == the oracle's prove_with_traces of that segment, all twelve blobs and the CTL challenges word for word (not only
    the single-segment path and not only transitively)."""
    import os
    from zkm_amd import tables as T
    from tools.bench_segment import HEIGHTS
    segs, bufs, host = [], [], []
    for v in range(3):
        tr, lg = _segment(v, HEIGHTS[16])
        d = [ctx.alloc(t.size).upload(t) for t in tr]
        bufs.append(d)
        host.append(tr)
        segs.append((d, lg, [5, v]))
    try:
        want = [ctx.prove_segment(d, lg, public_values=pub) for d, lg, pub in segs]
        got = ctx.prove_segments(segs)
        for v in range(3):
            assert list(got[v][2]) == list(want[v][2]) and (got[v][1] == want[v][1]).all()
            bad = np.nonzero(got[v][0] != want[v][0])[0]
            assert bad.size == 0, "segment %d: first differing word %d (table %d)" % (v, bad[0], int(np.searchsorted(want[v][2], bad[0], side="right")) - 1)
        ctl_tables, ctls = T.all_cross_table_lookups()
        old = oracle.get_threads()
        oracle.set_threads(min(64, os.cpu_count() or 1, __import__("bench").cpu_quota() or 64))
        try:
            for v, (_, lg, pub) in enumerate(segs):
                tables = [(T.TABLE_ENUM_ORDER[i], host[v][i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], lg[i], ctl_tables[i]) for i in range(12)]
                ref, rchal, roffs = oracle.prove_with_traces(tables, ctls, public_values=pub)
                assert list(got[v][2]) == list(roffs) and (got[v][1] == rchal).all()
                bad = np.nonzero(got[v][0] != ref)[0]
                assert bad.size == 0, "segment %d vs the ORACLE: first differing word %d (table %d)" % (
                    v, bad[0], int(np.searchsorted(roffs, bad[0], side="right")) - 1)
        finally:
            oracle.set_threads(old)
    finally:
        for d in bufs:
            for b in d:
                b.free()


@pytest.mark.gpu
def test_a_bad_segment_of_a_group_is_named(ctx, zkm):
    """A non-binary filter value in ONE segment of a lock-step group (cross_table_lookup.rs:741 "Non-binary filter?"): the call fails
    with the reference's message and names the caller's segment and the table; the context proves the good segments afterwards."""
    from zkm_amd import tables as T
    segs = []
    for v in range(3):
        tr, lg = _segment(v)
        segs.append((tr, lg, [v]))
    want = ctx.prove_segments(segs)
    # Logic table (Table::all()[10]): its operation flags are the filter of its CTL column set (logic.rs ctl_filter); 2 is not a filter value
    bad_tr = [t.copy() for t in segs[1][0]]
    w = T.WIDTH[T.TABLE_ENUM_ORDER[10]]
    bad_tr[10].reshape(w, -1)[0][0] = 2
    with pytest.raises(zkm.ZkmError, match=r"segment 1, table 10: Non-binary filter"):
        ctx.prove_segments([segs[0], (bad_tr, segs[1][1], segs[1][2]), segs[2]])
    got = ctx.prove_segments(segs)
    for v in range(3):
        assert (got[v][0] == want[v][0]).all()


@pytest.mark.gpu
def test_lockstep_edge_cases_memory_and_allocation_failures(zkm):
    """Zero segments, one segment, more segments than a group holds (34 > ZKM_MAX_SEG = 32: two groups per table), the allocator's
    out-of-memory retry while a stack is being built (the "debug_fail_allocs" test hook), and the memory accounting afterwards: every
    blob equals the single-segment path's, nothing stays allocated."""
    import os
    c = zkm.Context(0)
    os.environ["ZKM_ENABLE_TEST_HOOKS"] = "1"
    try:
        assert c.prove_segments([]) == []
        tr, lg = _segment(0)
        one = c.prove_segment(tr, lg, public_values=[4, 2])
        got = c.prove_segments([(tr, lg, [4, 2])])
        assert len(got) == 1 and (got[0][0] == one[0]).all() and (got[0][1] == one[1]).all() and list(got[0][2]) == list(one[2])
        segs = [(tr, lg, [v, 1]) for v in range(34)]               # same traces, 34 transcripts
        many = c.prove_segments(segs)
        for v in (0, 31, 32, 33):
            want = c.prove_segment(tr, lg, public_values=[v, 1])
            assert (many[v][0] == want[0]).all() and (many[v][1] == want[1]).all(), v
        # a memory budget that holds two of these segments (~370 MB each by the library's estimate): five segments go in three waves of 2, 2, 1
        five = [(_segment(v)[0], lg, [v, 9]) for v in range(5)]
        want5 = c.prove_segments(five)
        c.set_tuning("segments_memory_budget", 800 << 20)
        got5 = c.prove_segments(five)
        c.set_tuning("segments_memory_budget", 1)          # not even one: one segment per wave
        got1 = c.prove_segments(five)
        c.set_tuning("segments_memory_budget", 0)
        for v in range(5):
            assert (got5[v][0] == want5[v][0]).all() and (got1[v][0] == want5[v][0]).all() and (got5[v][1] == want5[v][1]).all()
        few = [(_segment(v)[0], lg, [v]) for v in range(3)]
        want = c.prove_segments(few)
        for k in (25, 300, 100000):
            c.set_tuning("debug_fail_allocs", k)
            got = c.prove_segments(few)
            for v in range(3):
                assert (got[v][0] == want[v][0]).all(), (k, v)
        c.set_tuning("debug_fail_allocs", 0)
        c.synchronize()
        live, cached = c.memory()
        assert live == c.resident_bytes()
    finally:
        os.environ.pop("ZKM_ENABLE_TEST_HOOKS", None)
        c.close()


@pytest.mark.gpu
def test_lockstep_groups_of_tall_tables_in_the_digit_coefficient_layout(ctx, zkm, oracle):
    """Tables of 2^18 .. 2^20 rows keep their coefficients in the digit layout of the two-pass inverse transform (zkm_coeff_exponent):
    the stacked forms of the layout conversions, the run-reading LDE pass and the exponent-indexed openings.  Three segments whose
    Arithmetic, Logic and Memory tables are tiled to 2^18 / 2^18 / 2^19 rows (the narrow tables: the case stays small): lock-step ==
    one at a time, and one of them == the oracle."""
    import os
    from zkm_amd import tables as T
    base = _segment(0)[1]
    heights = list(base)
    heights[0], heights[10], heights[11] = 18, 18, 19          # Arithmetic, Logic, Memory (Table::all() positions)
    segs = []
    for v in range(3):
        tr, lg = _segment(v, heights)
        segs.append((tr, lg, [v, 77]))
    got = ctx.prove_segments(segs)
    for v, (tr, lg, pub) in enumerate(segs):
        want, wchal, woffs = ctx.prove_segment(tr, lg, public_values=pub)
        assert list(got[v][2]) == list(woffs) and (got[v][1] == wchal).all()
        bad = np.nonzero(got[v][0] != want)[0]
        assert bad.size == 0, "segment %d: first differing word %d (table %d)" % (v, bad[0], int(np.searchsorted(woffs, bad[0], side="right")) - 1)
    ctl_tables, ctls = T.all_cross_table_lookups()
    tr, lg, pub = segs[1]
    tables = [(T.TABLE_ENUM_ORDER[i], tr[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], lg[i], ctl_tables[i]) for i in range(12)]
    old = oracle.get_threads()
    oracle.set_threads(min(64, os.cpu_count() or 1, __import__("bench").cpu_quota() or 64))
    try:
        ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=pub)
    finally:
        oracle.set_threads(old)
    assert (got[1][0] == ref).all() and (got[1][1] == rchal).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fields", [dict(num_challenges=1), dict(cap_height=2, num_queries=11), dict(arity_bits=3, final_poly_bits=2, pow_bits=9),
                                    dict(cap_height=0, arity_bits=2, final_poly_bits=1, num_queries=5)])
def test_lockstep_groups_under_other_stark_configs(ctx, zkm, oracle, fields):
    """Every field of StarkConfig (config.rs:4-34) that the stacked kernels index by segment -- one or two constraint challenges, the
    cap height (caps per segment in one download), the FRI arity / final polynomial length, the proof-of-work bits, the number of query
    rounds: lock-step == one at a time == the oracle under a non-standard config."""
    from zkm_amd import tables as T
    cfg, ocfg = ctx.standard_config(), oracle.standard_config()
    for k, v in fields.items():
        setattr(cfg, k, v)
        setattr(ocfg, k, v)
    segs = []
    for v in range(3):
        tr, lg = _segment(v)
        segs.append((tr, lg, [v, 5]))
    got = ctx.prove_segments(segs, cfg=cfg)
    for v, (tr, lg, pub) in enumerate(segs):
        want, wchal, woffs = ctx.prove_segment(tr, lg, public_values=pub, cfg=cfg)
        assert list(got[v][2]) == list(woffs) and (got[v][1] == wchal).all() and (got[v][0] == want).all(), (fields, v)
    ctl_tables, ctls = T.all_cross_table_lookups()
    tr, lg, pub = segs[2]
    tables = [(T.TABLE_ENUM_ORDER[i], tr[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], lg[i], ctl_tables[i]) for i in range(12)]
    ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=pub, cfg=ocfg)
    assert (got[2][0] == ref).all() and (got[2][1] == rchal).all()


@pytest.mark.gpu
def test_lockstep_segments_from_staged_segments(ctx, zkm):
    """zkm_segment_stage[_columns] (round 6): all twelve tables of a segment staged in ONE call -- one device block, one pair of events --
    while the previous call proves; zkm_staged_segment_ptrs gives the twelve device matrices.  Three staged segments (block form, column
    pointers, one of them not vouched canonical) through one lock-step call == the same segments from host arrays; the next call is staged
    before the current one is proven, as a driver would."""
    from zkm_amd import tables as T
    P = 0xFFFFFFFF00000001
    segs = []
    for v in range(3):
        tr, lg = _segment(v)
        segs.append((tr, lg, [v, 3]))
    want = ctx.prove_segments(segs)
    c = zkm.Context(0)
    try:
        def stage(v, form):
            tr, lg, _ = segs[v]
            if form == "columns":
                cols = [[np.ascontiguousarray(tr[i].reshape(T.WIDTH[T.TABLE_ENUM_ORDER[i]], -1)[k]) for k in range(T.WIDTH[T.TABLE_ENUM_ORDER[i]])] for i in range(12)]
                return c.stage_segment(cols, lg)
            if form == "loose":          # every small word + p: the staged copy is canonicalised when first consumed
                loose = [t.copy() for t in tr]
                for t in loose[::3]:
                    small = t < (1 << 32) - 1
                    t[small] += np.uint64(P)
                return c.stage_segment(loose, lg, canonical=False)
            return c.stage_segment(tr, lg)
        cur = [stage(0, "block"), stage(1, "columns"), stage(2, "loose")]
        nxt = [stage(v, "block") for v in range(3)]                      # the next call, staged before the current one is proven
        for call in (cur, nxt):
            got = c.prove_segments([(st.tables(), segs[v][1], segs[v][2]) for v, st in enumerate(call)])
            for v in range(3):
                assert list(got[v][2]) == list(want[v][2]) and (got[v][1] == want[v][1]).all()
                bad = np.nonzero(got[v][0] != want[v][0])[0]
                assert bad.size == 0, "segment %d: first differing word %d" % (v, bad[0])
            for st in call:
                assert st.ready() is True
                st.free()
        c.synchronize()
        live, _ = c.memory()
        assert live == c.resident_bytes()
        single = c.stage_trace(segs[0][0][0], T.WIDTH[T.TABLE_ENUM_ORDER[0]], segs[0][1][0])
        with pytest.raises(zkm.ZkmError, match="not a staged segment"):
            single.tables()
        single.free()
    finally:
        c.close()
