"""CPU suite: the N>1 sharding path under torch.distributed with the gloo backend, world_size 2."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    import numpy as np
    from zkm_amd import dist as zd
    import torch.distributed as dist
    world, rank, _ = zd.init("gloo")
    assert world == 2
    NSEG = 7
    mine = zd.assign_segments(NSEG, world, rank)
    def fake_prove(s):            # stands in for Context.prove_single_table (no GPU in this suite)
        time.sleep(0.02 * (rank + 1))
        return np.full(4, 1000 + s, dtype=np.uint64)
    proofs, elapsed = zd.prove_segments(fake_prove, NSEG)
    slowest = 0.02 * 2 * len(zd.assign_segments(NSEG, world, 1))
    assert elapsed >= min(slowest, 0.02 * len(zd.assign_segments(NSEG, world, 0))) - 1e-3
    assert abs(zd.max_over_ranks(float(rank)) - 1.0) < 1e-9
    if rank == 0:
        assert sorted(proofs) == list(range(NSEG))
        assert all(int(proofs[s][0]) == 1000 + s for s in proofs)
        print("OK", mine, round(elapsed, 3))
    else:
        assert proofs is None
    dist.destroy_process_group()
""") % ROOT


def test_assign_segments_partition():
    sys.path.insert(0, ROOT)
    from zkm_amd.dist import assign_segments
    for world in (1, 2, 4, 8):
        seen = sorted(s for r in range(world) for s in assign_segments(64, world, r))
        assert seen == list(range(64))
        assert all(len(assign_segments(64, world, r)) == 64 // world for r in range(world))


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29513", str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK [0, 2, 4, 6]" in r.stdout


BENCH_WORKER = textwrap.dedent("""
    # bench.py itself under torchrun with world_size 2 on gloo: a stub Context stands in for the GPU (none in this suite), so the
    # sharding (--segments 7: seeds 100..106 round-robin), the barrier / max-over-ranks timing, the per-rank trace slots and the JSON
    # line on rank 0 are the code the first multi-GPU run executes.
    import json, sys, time, types
    sys.path.insert(0, %r)
    import numpy as np, torch
    import zkm_amd
    from zkm_amd import dist as zd
    proved = []
    calls = []
    class Buf:
        def __init__(self, seed): self.seed = seed
        def upload(self, a): return self
        def download(self): return np.zeros(8, dtype=np.uint64)
        def free(self): pass
    class StubContext:
        def __init__(self, device): self.on = False
        def poseidon_trace(self, seed, n, log_n): return Buf(seed)
        def alloc(self, words): return Buf(None)
        def prove_single_table(self, trace, log_n, aux, nh):
            proved.append(trace.seed); time.sleep(0.01); return np.full(5, trace.seed, dtype=np.uint64)
        def prove_single_tables(self, traces, log_n, aux, nh):
            calls.append([t.seed for t in traces]); time.sleep(0.01 * len(traces))
            return [self.prove_single_table(t, log_n, aux, nh) for t in traces]
        def synchronize(self): pass
        def set_tuning(self, key, value): pass
        def profile(self, on): pass
        def profile_reset(self): pass
        def profile_records(self): return {"merkle_leaves": (len(proved), 1.0 * len(proved)), "ntt_pass_strided": (3, 0.5)}
        def close(self): pass
    zkm_amd.Context = StubContext
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda: proved.append("sync")      # (the timed region is what lies between the first two)
    torch.cuda.device_count = lambda: 2
    torch.cuda.current_device = lambda: 0
    torch.cuda.mem_get_info = lambda d=None: (200 << 30, 288 << 30)
    torch.cuda.get_device_properties = lambda d: types.SimpleNamespace(multi_processor_count=256)
    real_init = zd.init
    zd.init = lambda backend=None: real_init(%r)
    NCTX = %d
    STACK = %d
    sys.argv = [__file__, "--gpus", "2", "--segments", "7", "--warmup", "1", "--log-n", "10", "--no-cpu-baseline", "--no-extras",
                "--contexts", str(NCTX), "--stack", str(STACK)]
    import runpy
    runpy.run_path(%r, run_name="__main__")
    rank = int(__import__("os").environ["RANK"])
    want = [100 + s for s in range(rank, 7, 2)]
    i0 = proved.index("sync"); i1 = proved.index("sync", i0 + 1)
    warm, timed = proved[:i0], proved[i0 + 1:i1]
    if NCTX == 1 and STACK == 1:
        assert timed == want, (rank, proved)                        # one context, one proof per call: in segment order
    else:
        assert sorted(timed) == want, (rank, proved)                # a queue: every segment of this rank exactly once inside the clock
    if STACK == 1:
        assert len(warm) == NCTX and not calls                      # the warm-ups, one per context
    else:
        # lock-step calls: this rank's 4 (rank 0) or 3 (rank 1) segments in calls of at most STACK, the same number of calls per context
        # (a call of one goes through prove_single_table); every context warms up with one call of its own first
        assert all(1 < len(c) <= STACK for c in calls) and calls, (rank, calls)
        nctx = min(NCTX, -(-len(want) // STACK))
        assert len(warm) >= nctx, (rank, proved)
    print("RANK%%d OK %%s" %% (rank, proved))
""")


import pytest  # noqa: E402


@pytest.mark.parametrize("nctx,backend,stack", [(1, "gloo", 1), (2, "gloo", 1), (2, "nccl", 1), (2, "gloo", 2), (1, "gloo", 4)])
def test_bench_sharding_path_two_rank_gloo(tmp_path, nctx, backend, stack):
    """backend "nccl": what the driver's multi-GPU run asks for.  There is no GPU here, so RCCL cannot come up -- the probe child of
    every rank fails, the ranks agree on that over the gloo control group, and the job must finish on gloo and say so."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER % (ROOT, backend, nctx, stack, os.path.join(ROOT, "bench.py")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(29517 + nctx + (10 if backend == "nccl" else 0) + 20 * stack), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RANK0 OK" in r.stdout and "RANK1 OK" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout            # exactly one JSON line, from rank 0
    import json
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["segments_total"] == 7 and j["config"]["segments_per_gpu"] == 4
    assert j["config"]["contexts_per_gpu"] == min(nctx, -(-4 // stack)) and j["single_context"]["ms_per_step"] > 0
    assert j["config"]["proofs_per_call"] == (1 if stack == 1 else (2 if stack == 2 else 4)) and sum(j["config"]["calls_per_gpu"]) == 4
    assert j["timed_segments"] == 7 and j["timed_segments_per_gpu"] == 4 and j["config"]["distinct_traces_per_gpu"] == 4
    assert j["config"]["proofs_gathered_on_rank0"] == 7            # gathered over the process group after the clock stopped
    if backend == "nccl":
        assert j["config"]["process_group"].startswith("gloo (nccl failed: child-process probe failed"), j["config"]["process_group"]
        assert "RCCL unavailable" in r.stderr
    else:
        assert j["config"]["process_group"] == "gloo"
    assert j["config"]["preflight"]["world"] == 2 and r.stderr.count("zkm preflight rank=") == 2     # one pre-flight line per rank
    assert j["value"] > 0 and abs(j["value"] * j["ms_per_step"] * 4 / 7 / 1e3 - 1) < 1e-6   # value = total / elapsed, ms_per_step = elapsed / 4


def test_bench_started_as_a_plain_process_relaunches_itself_under_torchrun(tmp_path, monkeypatch):
    """`python bench.py --gpus N` -- the driver's N = 1 command with N swapped -- used to exit with a hint.  Started without
    a launcher's environment it now replaces itself by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py ...` (zkm_amd.dist.self_launch): both ranks run, rank 0 prints the one JSON line.  A process
    that HAS a launcher's environment and the wrong world is an error, not a relaunch."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER % (ROOT, "gloo", 1, 1, os.path.join(ROOT, "bench.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["ZKM_BENCH_ENTRY"] = str(script)      # (the stub-context wrapper is the script to relaunch; runpy hides it from bench.py's sys.argv[0])
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "relaunching as 2 ranks" in r.stderr and "--nproc-per-node 2 --master-addr 127.0.0.1" in r.stderr
    assert "RANK0 OK" in r.stdout and "RANK1 OK" in r.stdout
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["config"]["segments_total"] == 7 and j["config"]["preflight"]["world"] == 2
    sys.path.insert(0, ROOT)
    from zkm_amd import dist as zd
    cmd = zd.self_launch_command(8, ["bench.py", "--gpus", "8", "--steps", "20"], port=29400)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29400",
                       "bench.py", "--gpus", "8", "--steps", "20"]
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit, match="launcher's environment says WORLD_SIZE=1"):
        zd.self_launch(2, ["bench.py", "--gpus", "2"])


def test_run_workers_queue_and_errors():
    """The per-GPU worker queue: every segment exactly once, worker indices in range, first exception re-raised on the caller."""
    sys.path.insert(0, ROOT)
    import threading
    import time
    from zkm_amd.dist import run_workers
    seen, lock = [], threading.Lock()

    def fn(s, w):
        time.sleep(0.002 * (s % 3))
        with lock:
            seen.append((s, w))
        return s * s
    out = run_workers(fn, range(23), 4)
    assert out == {s: s * s for s in range(23)}
    assert sorted(s for s, _ in seen) == list(range(23)) and {w for _, w in seen} <= {0, 1, 2, 3} and len({w for _, w in seen}) > 1

    def bad(s, w):
        if s == 5:
            raise ValueError("segment 5")
        return s
    with pytest.raises(ValueError, match="segment 5"):
        run_workers(bad, range(50), 3)

    # lock-step calls: the queue holds calls of up to `stack` segments, every worker gets the same number of calls, sizes as even as possible
    from zkm_amd.dist import chunk_segments
    assert [len(c) for c in chunk_segments(range(20), 2, 4)] == [4, 4, 3, 3, 3, 3]
    assert [len(c) for c in chunk_segments(range(8), 2, 4)] == [4, 4]
    assert [len(c) for c in chunk_segments(range(3), 2, 4)] == [2, 1] and chunk_segments(range(3), 2, 1) == [[0], [1], [2]]
    for nseg in range(0, 70):
        for w in (1, 2, 3, 4):
            for st in (2, 4, 5, 32):
                c = chunk_segments(range(nseg), w, st)
                assert [x for call in c for x in call] == list(range(nseg)) and all(0 < len(call) <= st for call in c)
                assert len(c) % w == 0 or len(c) == nseg, (nseg, w, st, c)
    shapes = []

    def many(segs, w):
        with lock:
            shapes.append(len(segs))
        return [s * s for s in segs]
    assert run_workers(many, range(23), 2, stack=4) == {s: s * s for s in range(23)} and max(shapes) <= 4 and len(shapes) == 6
    with pytest.raises(RuntimeError, match="proofs for a call"):
        run_workers(lambda segs, w: [0], range(8), 2, stack=4)


def test_one_rank_per_gpu_is_enforced(monkeypatch):
    """bench.py refuses a world that does not fit the visible GPUs (two ranks sharing a device would fake a scaling curve)."""
    sys.path.insert(0, ROOT)
    import torch
    from zkm_amd import dist as zd
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert zd.check_gpus(2, 1) == 1
    with pytest.raises(SystemExit, match="ranks on this node but 2 GPUs"):
        zd.check_gpus(8, 3)
    assert zd.check_gpus(8, 3, share_gpu=True) == 0               # the single-GPU rehearsal puts every rank on device 0
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    with pytest.raises(SystemExit, match="no GPU visible"):
        zd.check_gpus(1, 0)


def test_cpu_pinning_splits_the_allowed_cpus(monkeypatch):
    """Without a readable GPU topology every local rank gets its own even slice of the allowed CPUs (and with one, the GPU's NUMA
    node); worker threads started afterwards inherit the mask."""
    sys.path.insert(0, ROOT)
    import torch
    from zkm_amd import dist as zd
    allowed = sorted(os.sched_getaffinity(0))
    set_to = {}
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: set_to.__setitem__("cpus", list(cpus)))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    seen = []
    for r in range(2):
        info = zd.pin_to_gpu(r, 2)
        assert info["pinned"] and info["how"] == "even-split"
        seen.append(set(set_to["cpus"]))
        assert seen[-1] <= set(allowed) and seen[-1]
    if len(allowed) >= 2:
        assert not (seen[0] & seen[1])
    # a GPU on NUMA node with known CPUs: that node's CPUs (intersected with the allowed set)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(zd, "gpu_numa_cpus", lambda d: allowed[:max(1, len(allowed) // 2)])
    info = zd.pin_to_gpu(0, 1)
    assert info["pinned"] and info["how"] == "numa" and set_to["cpus"] == allowed[:max(1, len(allowed) // 2)]
    assert zd._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_forced_process_group_at_world_1():
    """ZKM_FORCE_PG=1: a single process still builds the process group (the world-1 rehearsal of the RCCL path on a one-GPU box).  Here:
    gloo requested -> gloo; nccl requested without a GPU -> probe fails -> gloo, with the reason recorded."""
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        from zkm_amd import dist as zd
        assert zd.init(sys.argv[1]) == (1, 0, 0)
        import torch.distributed as dist
        assert dist.is_initialized() and dist.get_world_size() == 1
        zd.barrier()
        assert zd.max_over_ranks(1.25) == 1.25
        assert list(zd.gather_proofs({3: np.arange(4)})) == [3]
        print("PG", zd.process_group_info()["process_group"])
        zd.shutdown()
    """) % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ZKM_FORCE_PG"] = "1"
    for want, expect in (("gloo", "PG gloo\n"), ("nccl", "PG gloo (nccl failed: child-process probe failed")):
        r = subprocess.run([sys.executable, "-c", code, want], capture_output=True, text=True, timeout=240, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert expect in r.stdout, r.stdout + r.stderr


def test_cpu_baseline_is_measured_in_the_run(tmp_path):
    """bench.py's cpu_baseline leg at toy sizes (the oracle on this host, no GPU): with --cpu-full always the value is ONE full-size
    proof timed in the run (verified by the oracle's verifier) and the bounded sample stays beside it; with --cpu-full never the value is
    the sample's extrapolation and a tracked full-size figure may only appear as `full_size_reference`, marked as not measured in the run
    (the figure in the line is measured by the run that prints it)."""
    code = textwrap.dedent("""
        import json, sys
        sys.path.insert(0, %r)
        sys.argv = ["bench.py"]
        import bench
        a, checker = bench.cpu_baseline(5, 7, full="always")
        assert a["value_source"] == "measured in this run at full size" and abs(a["value"] * a["full_size_s"] - 1) < 1e-9
        assert a["sample_log_n"] == 5 and a["sample_value_extrapolated"] > 0 and a["kind"] == "port" and a["cores"] >= 1
        b, _ = bench.cpu_baseline(5, 7, full="never")
        assert b["value"] == b["sample_value_extrapolated"] and "extrapolated from this run's bounded sample" in b["value_source"]
        assert "full_size_s" not in b and "full_size_reference" not in b          # (the tracked file is a 2^20-row figure: not quoted at 2^7)
        c, _ = bench.cpu_baseline(5, 20, full="never")
        ref = c.get("full_size_reference")
        assert ref is None or "NOT measured in this run" in ref["source"]
        assert sorted(bench.FINGERPRINT_FILES) == bench.FINGERPRINT_FILES and "zkm_amd/csrc/poseidon_mfma_dev.h" in bench.FINGERPRINT_FILES
        assert "zkm_amd/csrc/pool.hip" in bench.FINGERPRINT_FILES and "include/zkm_hip.h" in bench.FINGERPRINT_FILES
        print("OK")
    """) % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
