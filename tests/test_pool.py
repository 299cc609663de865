"""zkm_pool: one process, many devices (include/zkm_hip.h) -- the reference's one-process segment loop
(prover/examples/utils/src/utils.rs:57-68, 105-133) as a queue of lock-step groups served by one worker thread per context.
CPU part: the plan, the argument checks, the loud failure without a GPU.  GPU part: a pool on device 0 with two workers produces, for
every segment, the words of the single-context path and of the oracle, whatever the grouping."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_plan_is_the_harness_rule(zkm):
    """The groups of a pool call follow the rule bench.py's harness uses for its calls (zkm_amd/dist.py chunk_segments): the same number
    of groups for every worker, none above max_stack, sizes as even as the count allows, segments in order."""
    from zkm_amd.dist import chunk_segments
    assert zkm.pool_plan(20, 2, 4) == [4, 4, 3, 3, 3, 3] and zkm.pool_plan(64, 8, 8) == [8] * 8 and zkm.pool_plan(0, 4, 8) == []
    assert zkm.pool_plan(5, 16, 8) == [1] * 5                 # fewer segments than workers: one each
    assert zkm.pool_plan(17, 2, 0) == zkm.pool_plan(17, 2, 8)  # 0 = the default of 8 per group
    for nseg in range(0, 70):
        for workers in (1, 2, 3, 4, 16):
            for stack in (1, 2, 4, 5, 8, 32):
                want = [len(c) for c in chunk_segments(range(nseg), workers, stack)] if stack > 1 else [1] * nseg
                assert zkm.pool_plan(nseg, workers, stack) == want, (nseg, workers, stack)


def test_pool_argument_checks_and_loud_failure_without_a_gpu(zkm):
    """A device listed twice, no device, no context: refused before anything touches the runtime.  Without a GPU the pool cannot be
    created at all -- there is no CPU fallback behind it."""
    import torch
    for devices, per, msg in (((0, 0), 1, "listed twice"), ((), 1, "at least one device"), ((0,), 0, "at least one device")):
        with pytest.raises(zkm.ZkmError, match=msg):
            zkm.Pool(devices, per)
    if not torch.cuda.is_available():
        with pytest.raises(zkm.ZkmError, match="zkm_pool_create: context 0 on device 0"):
            zkm.Pool((0,), 2)


def _segment(variant):
    from tests.test_segments_batch import _segment as seg
    return seg(variant)


@pytest.mark.gpu
def test_pool_on_one_device_equals_single_context_and_oracle(ctx, zkm, oracle):
    """Seven segments through a pool of two workers on device 0 (the multi-device code path with the one device a test box has: the
    workers are the same threads, contexts and queue a second device would get), groups of at most 3 -> 3, 2, 2 ... : every blob and
    every challenge equals zkm_prove_segment's for that segment alone; segment 0 also the oracle's.  Both workers took part."""
    from zkm_amd import tables as T
    segs = []
    for v in range(7):
        tr, lg = _segment(v)
        segs.append((tr, lg, [v, 5, 11 * v]))
    want = [ctx.prove_segment(tr, lg, public_values=pub) for tr, lg, pub in segs]
    pool = zkm.Pool((0,), 2)
    try:
        assert pool.workers() == 2 and pool.device(0) == 0 and pool.device(1) == 0
        for stack in (3, 1, 0):
            got = pool.prove_segments(segs, max_stack=stack)
            for v in range(7):
                assert list(got[v][2]) == list(want[v][2]) and (got[v][1] == want[v][1]).all()
                bad = np.nonzero(got[v][0] != want[v][0])[0]
                assert bad.size == 0, "max_stack %d, segment %d: first differing word %d" % (stack, v, bad[0])
            plan = zkm.pool_plan(7, 2, stack)
            groups = [pool.last_assignment(v)[1] for v in range(7)]
            assert groups == [g for g, k in enumerate(plan) for _ in range(k)]          # consecutive segments, the planned sizes
            if len(plan) >= 2:
                assert {pool.last_assignment(v)[0] for v in range(7)} == {0, 1}
        # column pointers (the Rust caller's form), host memory
        col_segs = []
        for tr, lg, pub in segs[:3]:
            cols = []
            for i in range(12):
                w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
                m = tr[i].reshape(w, -1)
                cols.append([np.ascontiguousarray(m[k]) for k in range(w)])
            col_segs.append((cols, lg, pub))
        got = pool.prove_segments(col_segs, max_stack=2)
        for v in range(3):
            assert (got[v][0] == want[v][0]).all() and (got[v][1] == want[v][1]).all()
        pool.set_tuning("throughput_profile", 1)
        got = pool.prove_segments(segs[:4], max_stack=2)
        for v in range(4):
            assert (got[v][0] == want[v][0]).all()
        with pytest.raises(zkm.ZkmError, match="unknown key"):
            pool.set_tuning("no_such_key", 1)
        assert pool.prove_segments([]) == []                                   # no segment: nothing to do, no worker started
        with pytest.raises(zkm.ZkmError, match="max_stack beyond 32"):
            pool.prove_segments(segs[:2], max_stack=33)
        one = pool.prove_segments(segs[5:6], max_stack=8)                       # fewer groups than workers: the calling thread proves it
        assert (one[0][0] == want[5][0]).all() and pool.last_assignment(0) == (0, 0)
    finally:
        pool.close()
    ctl_tables, ctls = T.all_cross_table_lookups()
    tr, lg, pub = segs[0]
    tables = [(T.TABLE_ENUM_ORDER[i], tr[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], lg[i], ctl_tables[i]) for i in range(12)]
    ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=pub)
    assert (want[0][0] == ref).all() and (want[0][1] == rchal).all()


@pytest.mark.gpu
def test_pool_names_worker_device_and_segments_of_a_failure(zkm):
    """A segment that cannot be proven (a non-binary CTL filter in segment 4's Logic table) fails its group: the message names the worker,
    the device, the group's positions in the CALL and, inside, the segment and the table; nothing is returned."""
    from zkm_amd import tables as T
    segs = []
    for v in range(6):
        tr, lg = _segment(v)
        segs.append(([t.copy() for t in tr], lg, [v]))
    # Logic table (Table::all()[10]): its operation flags are the filter of its CTL column set (logic.rs ctl_filter); 2 is not a filter value
    logic = 10
    w = T.WIDTH[T.TABLE_ENUM_ORDER[logic]]
    segs[4][0][logic].reshape(w, -1)[0][0] = 2
    pool = zkm.Pool((0,), 2)
    try:
        with pytest.raises(zkm.ZkmError) as e:
            pool.prove_segments(segs, max_stack=2)
        msg = str(e.value)
        assert "worker " in msg and "(device 0)" in msg and "segments 4..4" in msg, msg      # (6 segments, 2 workers, <= 2 per group: 2, 2, 1, 1)
        assert "segment 4" in msg and "table %d" % logic in msg and "Non-binary filter?" in msg, msg
        good = pool.prove_segments(segs[:4], max_stack=2)       # the pool is usable after a failure
        assert len(good) == 4
    finally:
        pool.close()
