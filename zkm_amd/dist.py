"""Multi-GPU plumbing: independent segment proofs shard across ranks with NO data-path collective.

The reference proves segments one after another in one process (prover/examples/utils/src/utils.rs:57-68,
105-133); segments are independent proofs, so the MI355X design is one process per GPU, round-robin
assignment, and a host-side gather of the finished proofs (a few hundred KB each).  torch.distributed
carries only the barrier and the max-over-ranks time (RCCL, i.e. backend "nccl" on ROCm, once it has been seen to
work; gloo otherwise and in the CPU tests) and the proof gather (always the gloo control group, outside the clock).
"""
import os
import time

import torch
import torch.distributed as dist



def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def self_launch_command(nproc, argv, port=None):
    """The command that runs `argv` (script + its arguments) as `nproc` ranks of one node, one per GPU: the driver's launch line
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script args...).  The
    rendezvous is on 127.0.0.1 (the container hostname may not resolve) and the port is a free one unless given."""
    import socket
    import sys
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
            "--master-port", str(port)] + list(argv)


def self_launch(nproc, argv):
    """`python bench.py --gpus N` started as a PLAIN process (no RANK / WORLD_SIZE in the environment): replace this process by the
    torchrun launch of the same command line -- the multi-GPU job must be startable the way the single-GPU one is.  A
    process that already has a launcher's environment and the wrong world size is an error, not a relaunch (it would recurse)."""
    import sys
    if any(k in os.environ for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")):
        raise SystemExit("zkm_amd.dist: --gpus %d but the launcher's environment says WORLD_SIZE=%s: launch one rank per GPU "
                         "(python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 ...)" % (
                             nproc, os.environ.get("WORLD_SIZE", "unset"), nproc))
    cmd = self_launch_command(nproc, argv)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("zkm_amd.dist: relaunching as %d ranks: %s" % (nproc, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(cmd[0], cmd)


def check_gpus(world, local_rank, share_gpu=False):
    """One process per GPU means one GPU per process: refuse to start when torchrun's world does not fit the visible devices
    (two ranks silently sharing a device would report a scaling curve that is not one).  share_gpu is the single-GPU rehearsal."""
    ngpu = torch.cuda.device_count()
    if ngpu == 0:
        raise SystemExit("zkm_amd.dist: no GPU visible (torch.cuda.device_count() == 0); there is no CPU fallback for the proving path")
    if share_gpu:
        return 0
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > ngpu or local_rank >= ngpu:
        raise SystemExit("zkm_amd.dist: %d ranks on this node but %d GPUs visible (LOCAL_RANK %d): launch one rank per GPU "
                         "(--nproc-per-node <= %d) or fix HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES" % (local_world, ngpu, local_rank, ngpu))
    return local_rank


def _parse_cpulist(s):
    cpus = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(device):
    """CPUs of the NUMA node the GPU hangs off (sysfs), or None when the topology cannot be read."""
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        return _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) or None
    except Exception:
        return None


def pin_to_gpu(local_rank, local_world, device=None):
    """Restrict this process (and the worker threads it starts later) to CPUs near its GPU: the GPU's NUMA node when sysfs tells,
    otherwise an even slice of the allowed CPUs -- so that N ranks x k proving threads neither pile onto the same cores nor
    drive their GPU across the socket interconnect.  Never fatal; returns a description for the bench line."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        return {"pinned": False, "why": "sched_getaffinity unavailable"}
    how = "numa"
    me = device if device is not None else local_rank
    cpus = gpu_numa_cpus(me) if torch.cuda.is_available() else None
    if cpus:
        cpus = sorted(set(cpus) & set(allowed))
    if cpus and local_world > 1 and device is None:
        # ranks whose GPUs hang off the same NUMA node split that node's CPUs among themselves
        nodes = [gpu_numa_cpus(r) or [] for r in range(min(local_world, torch.cuda.device_count()))]
        sharers = [r for r, nc in enumerate(nodes) if set(nc) & set(cpus)]
        if local_rank in sharers and len(sharers) > 1 and len(cpus) >= 2 * len(sharers):
            k = sharers.index(local_rank)
            per = len(cpus) // len(sharers)
            cpus = cpus[k * per:(k + 1) * per]
    if not cpus:
        how = "even-split"
        per = max(1, len(allowed) // max(1, local_world))
        cpus = allowed[local_rank * per:(local_rank + 1) * per] or allowed
    try:
        os.sched_setaffinity(0, cpus)
    except Exception as e:
        return {"pinned": False, "why": str(e)}
    return {"pinned": True, "how": how, "cpus": len(cpus), "first_cpu": cpus[0], "last_cpu": cpus[-1]}


# The process group.  The CONTROL plane is always a gloo group (host memory, TCP on 127.0.0.1): rendezvous, agreement between the
# ranks, and the gather of the finished proofs.  RCCL ("nccl") is a second group that carries what the clock needs -- the barrier
# and the max-over-ranks time -- and is adopted only after every rank has seen it work: first in a throw-away child process per rank
# (a hang there is killed by a timeout instead of hanging the job), then in this process, with the ranks agreeing on the outcome over
# gloo.  Any failure leaves the job on gloo and says so in the bench line: RCCL carries one barrier and one MAX, it must not be able
# to zero a measurement.
_PG = {"group": None, "backend": None, "device": None, "info": None}


def _quiet_stdout(fn):
    """gloo announces its connections on STDOUT; rank 0's stdout must carry nothing but the JSON line: swap the descriptor while it connects."""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        return fn()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _all_agree(ok):
    """True iff `ok` on every rank (MIN over the gloo control group)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def _rccl_probe(world, rank, device, timeout_s):
    """Run zkm_amd/rccl_probe.py as a child of every rank: its own RCCL communicator over a file store, barrier + all_reduce(MAX) on a
    device tensor.  Returns (ok, detail)."""
    import json
    import shutil
    import subprocess
    import sys
    import tempfile
    box = [tempfile.mkdtemp(prefix="zkm_rccl_probe_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    store = os.path.join(box[0], "store")
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_probe.py"), store, str(rank), str(world), str(device),
           str(max(5, int(timeout_s) - 10))]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        line = ([l for l in r.stdout.splitlines() if l.startswith("{")] or ["{}"])[-1]
        detail = json.loads(line)
        ok = r.returncode == 0 and bool(detail.get("ok"))
        if not ok:
            detail.setdefault("error", (r.stderr.strip().splitlines() or ["exit code %d" % r.returncode])[-1][:300])
    except subprocess.TimeoutExpired:
        ok, detail = False, {"error": "probe timed out after %d s (killed)" % timeout_s}
    except Exception as e:  # noqa: BLE001 -- whatever went wrong, the job continues on gloo
        ok, detail = False, {"error": "%s: %s" % (type(e).__name__, e)}
    detail["wall_s"] = round(time.perf_counter() - t0, 3)
    all_ok = _all_agree(ok)
    dist.barrier()
    if rank == 0:
        shutil.rmtree(box[0], ignore_errors=True)
    return all_ok, detail


def _rccl_adopt(world, rank, device):
    """Create the RCCL group in this process and run the two collectives the clock uses.  (ok-on-every-rank, detail)."""
    import datetime
    detail, g, ok = {}, None, False
    t0 = time.perf_counter()
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # RCCL needs dmabuf IPC on this driver stack
        dev = torch.device("cuda", device)
        g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=int(os.environ.get("ZKM_RCCL_TIMEOUT_S", "300"))), device_id=dev)
        detail["new_group_s"] = round(time.perf_counter() - t0, 3)
        t1 = time.perf_counter()
        dist.barrier(group=g, device_ids=[device])
        t = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=g)
        torch.cuda.synchronize(dev)
        detail["first_barrier_allreduce_s"] = round(time.perf_counter() - t1, 3)
        ok = float(t.item()) == float(world - 1)
        if not ok:
            detail["error"] = "all_reduce(MAX) of the ranks returned %r, expected %d" % (t.item(), world - 1)
    except Exception as e:  # noqa: BLE001
        detail["error"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:300] if str(e) else "")
    return (_all_agree(ok), detail, g)


def init(backend=None):
    """Initialise the process group from torchrun's environment.  A single process does nothing unless ZKM_FORCE_PG=1 (the world-1
    rehearsal of the whole RCCL path on the one GPU a development box has).  backend: "nccl" (RCCL; default when a GPU is visible) or
    "gloo".  Returns (world, rank, local_rank); process_group_info() tells what was adopted."""
    world, rank, local_rank = env_world()
    force = os.environ.get("ZKM_FORCE_PG") == "1"
    if dist.is_initialized() or (world == 1 and not force):
        return world, rank, local_rank
    if world == 1:
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    want = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    t0 = time.perf_counter()
    _quiet_stdout(lambda: dist.init_process_group("gloo"))
    info = {"requested": want, "control": "gloo", "control_init_s": round(time.perf_counter() - t0, 3)}
    _PG.update(group=None, backend="gloo", device=None, info=info)
    if want == "nccl":
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0   # the caller has set the rank's device (bench.py)
        ok, why = True, None
        if os.environ.get("ZKM_RCCL_PROBE", "1") != "0":
            ok, info["probe"] = _rccl_probe(world, rank, device, int(os.environ.get("ZKM_RCCL_PROBE_TIMEOUT_S", "180")))
            why = None if ok else "child-process probe failed on at least one rank (this rank: %s)" % info["probe"].get("error", "ok")
        if ok:
            ok, info["adopt"], g = _rccl_adopt(world, rank, device)
            if ok:
                _PG.update(group=g, backend="nccl", device=device)
            else:
                why = "in-process group failed on at least one rank (this rank: %s)" % info["adopt"].get("error", "ok")
        info["process_group"] = "nccl" if ok else "gloo (nccl failed: %s)" % why
        if not ok and rank == 0:
            import sys
            print("zkm_amd.dist: RCCL unavailable, barrier and max-over-ranks stay on gloo: %s" % why, file=sys.stderr)
    else:
        info["process_group"] = "gloo"
    return world, rank, local_rank


def process_group_info():
    """What init() adopted: {"process_group": "nccl" | "gloo" | "gloo (nccl failed: ...)", "probe": ..., "adopt": ...}; None without a group."""
    return _PG["info"]


def backend():
    return _PG["backend"] if dist.is_initialized() else None


def preflight(device, world, rank, local_rank, need_bytes=0, free_bytes=None, pin=None, what=""):
    """Before the clock: one line per rank on stderr with what a failed multi-GPU run is usually explained by -- visible devices, free
    HBM against what this rank is about to allocate, hardware queues, CPU affinity, the process group adopted.  Returns the same as a
    dict (rank 0's goes into the bench line).  Raises SystemExit only when the memory cannot fit."""
    import sys
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if free_bytes is None and ngpu:
        free_bytes = torch.cuda.mem_get_info(device)[0]
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        ncpu = os.cpu_count()
    info = {"rank": rank, "world": world, "local_rank": local_rank, "device": device, "gpus_visible": ngpu,
            "free_hbm_gib": round((free_bytes or 0) / 2.0**30, 1), "need_hbm_gib": round(need_bytes / 2.0**30, 1), "need": what,
            "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "cpus_allowed": ncpu, "pinned": bool(pin and pin.get("pinned")),
            "process_group": (_PG["info"] or {}).get("process_group", "none (single process)")}
    print("zkm preflight " + " ".join("%s=%s" % (k, str(v).replace(" ", "_")) for k, v in info.items()), file=sys.stderr, flush=True)
    if free_bytes is not None and need_bytes > free_bytes:
        raise SystemExit("zkm_amd.dist: rank %d needs %.1f GiB of HBM on device %d (%s) but %.1f GiB are free" % (
            rank, need_bytes / 2.0**30, device, what, free_bytes / 2.0**30))
    return info


def assign_segments(num_segments, world, rank):
    """Round-robin: segment s is proven by rank s % world (config 3: 64 segments -> 8 per GPU)."""
    return list(range(rank, num_segments, world))


def barrier():
    if not dist.is_initialized():
        return
    if _PG["backend"] == "nccl":
        dist.barrier(group=_PG["group"], device_ids=[_PG["device"]])
    else:
        dist.barrier()


def max_over_ranks(seconds):
    """Whole-job time = the slowest rank."""
    if not dist.is_initialized():
        return float(seconds)
    if _PG["backend"] == "nccl":
        t = torch.tensor([seconds], dtype=torch.float64, device=torch.device("cuda", _PG["device"]))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_PG["group"])
    else:
        t = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_proofs(local):
    """local: {segment_index: proof ndarray}.  Returns the merged dict on rank 0 (None elsewhere).  Host objects travel over the gloo
    control group (host memory to host memory, no detour through the device), after the clock has stopped."""
    if not dist.is_initialized():
        return dict(local)
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local, out, dst=0)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for d in out:
        for k, v in d.items():
            if k in merged:
                raise RuntimeError("segment %d proven twice" % k)
            merged[k] = v
    return merged


def shutdown():
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 -- the measurement is printed already; a teardown error must not turn the exit code
            pass
    _PG.update(group=None, backend=None, device=None)


def chunk_segments(segments, workers, stack):
    """The calls a rank makes when `workers` contexts prove up to `stack` segments per call in lock-step: every worker gets the same
    number of calls (the fewest that keeps a call within `stack`), and the calls are as even as the count allows -- 20 segments,
    2 workers, stack 4 -> six calls of 4, 4, 3, 3, 3, 3; 8 -> two calls of 4.  stack <= 1: one segment per call."""
    segs = list(segments)
    if stack <= 1 or not segs:
        return [[s] for s in segs]
    workers = max(1, workers)
    per_worker = -(-len(segs) // (workers * stack))
    ncalls = min(len(segs), workers * per_worker)
    base, extra = divmod(len(segs), ncalls)
    out, i = [], 0
    for k in range(ncalls):
        size = base + (1 if k < extra else 0)
        out.append(segs[i:i + size])
        i += size
    return out


def run_workers(prove_fn, segments, workers, stack=1):
    """`workers` host threads pull `segments` from one queue; thread w calls prove_fn(s, w) -- w selects the worker's own context
    (own stream, allocator and transcript; a context is single-owner).  Independent segments proven side by side on
    one GPU fill each other's transcript round trips and overlap HBM-bound NTT passes with VALU-bound hashing.  stack > 1: the queue
    holds CALLS of up to `stack` segments (chunk_segments) and prove_fn(list_of_segments, w) returns their proofs in order -- the
    lock-step entry points (zkm_prove_single_tables / zkm_prove_segments).  The first exception stops the queue and is re-raised here."""
    import threading
    items = chunk_segments(segments, workers, stack) if stack > 1 else list(segments)
    it = iter(items)
    lock = threading.Lock()
    out, errs = {}, []

    def loop(w):
        while not errs:
            with lock:
                s = next(it, None)
            if s is None:
                return
            try:
                r = prove_fn(s, w)
                with lock:
                    if stack > 1:
                        if len(r) != len(s):
                            raise RuntimeError("run_workers: %d proofs for a call of %d segments" % (len(r), len(s)))
                        out.update(zip(s, r))
                    else:
                        out[s] = r
            except BaseException as e:  # noqa: B902 -- re-raised on the calling thread
                errs.append(e)
    th = [threading.Thread(target=loop, args=(w,), name="zkm-worker-%d" % w) for w in range(workers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return out


def prove_segments(prove_fn, num_segments, sync_fn=None, gather=True, workers=1, stack=1):
    """Prove `num_segments` independent segments across all ranks (this is bench.py's timed region).

    prove_fn(segment_index) -> proof; sync_fn() drains the local GPU (torch.cuda.synchronize).  The clock runs from a
    barrier + sync to a sync + barrier and the slowest rank defines the job time.  The proofs are gathered on rank 0 AFTER the
    clock has stopped (gather=False skips it).  workers = k > 1: this rank's segments go through run_workers and prove_fn is
    called as prove_fn(segment_index, worker_index); stack > 1: as prove_fn(list_of_segment_indices, worker_index) -> list of proofs
    (run_workers).  A rank without segments (num_segments < world) only takes part in the barriers.
    Returns (proofs_on_rank0_or_None, whole_job_seconds)."""
    world, rank, _ = env_world()
    mine = assign_segments(num_segments, world, rank)
    barrier()
    if sync_fn:
        sync_fn()
    t0 = time.perf_counter()
    local = {s: prove_fn(s) for s in mine} if (workers <= 1 and stack <= 1) else run_workers(prove_fn, mine, max(1, workers), stack)
    if sync_fn:
        sync_fn()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    return (gather_proofs(local) if gather else None), elapsed
