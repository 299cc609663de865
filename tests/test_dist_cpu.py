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
        def synchronize(self): pass
        def set_tuning(self, key, value): pass
        def profile(self, on): pass
        def profile_reset(self): pass
        def profile_records(self): return {"merkle_leaves": (len(proved), 1.0 * len(proved)), "ntt_pass_strided": (3, 0.5)}
        def close(self): pass
    zkm_amd.Context = StubContext
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda: None
    torch.cuda.device_count = lambda: 2
    torch.cuda.current_device = lambda: 0
    torch.cuda.mem_get_info = lambda d=None: (200 << 30, 288 << 30)
    torch.cuda.get_device_properties = lambda d: types.SimpleNamespace(multi_processor_count=256)
    real_init = zd.init
    zd.init = lambda backend=None: real_init(%r)
    NCTX = %d
    sys.argv = ["bench.py", "--gpus", "2", "--segments", "7", "--warmup", "1", "--log-n", "10", "--no-cpu-baseline", "--no-extras",
                "--contexts", str(NCTX)]
    import runpy
    runpy.run_path(%r, run_name="__main__")
    rank = int(__import__("os").environ["RANK"])
    want = [100 + s for s in range(rank, 7, 2)]
    timed = proved[NCTX:NCTX + len(want)]                           # the first NCTX calls are the warm-ups, one per context
    if NCTX == 1:
        assert timed == want, (rank, proved)                        # one context: in segment order
    else:
        assert sorted(timed) == want, (rank, proved)                # a queue: every segment of this rank exactly once
    print("RANK%%d OK %%s" %% (rank, proved))
""")


import pytest  # noqa: E402


@pytest.mark.parametrize("nctx,backend", [(1, "gloo"), (2, "gloo"), (2, "nccl")])
def test_bench_sharding_path_two_rank_gloo(tmp_path, nctx, backend):
    """backend "nccl": what the driver's multi-GPU run asks for.  There is no GPU here, so RCCL cannot come up -- the probe child of
    every rank fails, the ranks agree on that over the gloo control group, and the job must finish on gloo and say so."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER % (ROOT, backend, nctx, os.path.join(ROOT, "bench.py")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(29517 + nctx + (10 if backend == "nccl" else 0)), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RANK0 OK" in r.stdout and "RANK1 OK" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout            # exactly one JSON line, from rank 0
    import json
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["segments_total"] == 7 and j["config"]["segments_per_gpu"] == 4
    assert j["config"]["contexts_per_gpu"] == nctx and j["single_context"]["ms_per_step"] > 0
    assert j["timed_segments"] == 7 and j["timed_segments_per_gpu"] == 4 and j["config"]["distinct_traces_per_gpu"] == 4
    assert j["config"]["proofs_gathered_on_rank0"] == 7            # gathered over the process group after the clock stopped
    if backend == "nccl":
        assert j["config"]["process_group"].startswith("gloo (nccl failed: child-process probe failed"), j["config"]["process_group"]
        assert "RCCL unavailable" in r.stderr
    else:
        assert j["config"]["process_group"] == "gloo"
    assert j["config"]["preflight"]["world"] == 2 and r.stderr.count("zkm preflight rank=") == 2     # one pre-flight line per rank
    assert j["value"] > 0 and abs(j["value"] * j["ms_per_step"] * 4 / 7 / 1e3 - 1) < 1e-6   # value = total / elapsed, ms_per_step = elapsed / 4


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
