"""Multi-GPU plumbing: independent segment proofs shard across ranks with NO data-path collective.

The reference proves segments one after another in one process (prover/examples/utils/src/utils.rs:57-68,
105-133); segments are independent proofs, so the MI355X design is one process per GPU, round-robin
assignment, and a host-side gather of the finished proofs (a few hundred KB each).  torch.distributed
(backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) carries only the barrier and the max-over-ranks
time; the proof gather goes over a gloo side group (host memory to host memory, outside the clock).
"""
import os
import time

import torch
import torch.distributed as dist

_HOST_GROUP = None  # gloo side group for host-side object collectives when the main backend is nccl


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


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


def init(backend=None):
    """Initialise the process group from torchrun's environment (no-op for a single process)."""
    world, rank, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            # RCCL needs dmabuf IPC on this driver stack (the image exports it; keep it for any env built by hand)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dev = torch.cuda.current_device()   # the caller has set the rank's device (bench.py: check_gpus -> set_device)
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend, **kw)
    return world, rank, local_rank


def assign_segments(num_segments, world, rank):
    """Round-robin: segment s is proven by rank s % world (config 3: 64 segments -> 8 per GPU)."""
    return list(range(rank, num_segments, world))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds):
    """Whole-job time = the slowest rank."""
    if not dist.is_initialized():
        return float(seconds)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_proofs(local):
    """local: {segment_index: proof ndarray}.  Returns the merged dict on rank 0 (None elsewhere).  Host objects travel over the
    gloo side group when the main backend is RCCL (no detour through device memory)."""
    global _HOST_GROUP
    if not dist.is_initialized():
        return dict(local)
    if dist.get_backend() == "nccl" and _HOST_GROUP is None:
        # created on first use (a collective call: every rank gathers or none does): gloo announces its connections on STDOUT, and a run
        # that gathers nothing -- the driver's scaling runs -- must print nothing but rank 0's JSON line
        # (... and the announcement itself goes to stderr: the descriptor is swapped while the group connects)
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            _HOST_GROUP = dist.new_group(backend="gloo")
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local, out, dst=0, group=_HOST_GROUP)
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
    global _HOST_GROUP
    if dist.is_initialized():
        dist.destroy_process_group()
    _HOST_GROUP = None


def run_workers(prove_fn, segments, workers):
    """`workers` host threads pull `segments` from one queue; thread w calls prove_fn(s, w) -- w selects the worker's own context
    (own stream, allocator and transcript; a context is single-owner).  Independent segments proven side by side on
    one GPU fill each other's transcript round trips and overlap HBM-bound NTT passes with VALU-bound hashing.  The first
    exception stops the queue and is re-raised here."""
    import threading
    it = iter(list(segments))
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


def prove_segments(prove_fn, num_segments, sync_fn=None, gather=True, workers=1):
    """Prove `num_segments` independent segments across all ranks (this is bench.py's timed region).

    prove_fn(segment_index) -> proof; sync_fn() drains the local GPU (torch.cuda.synchronize).  The clock runs from a
    barrier + sync to a sync + barrier and the slowest rank defines the job time.  The proofs are gathered on rank 0 AFTER the
    clock has stopped (gather=False skips it).  workers = k > 1: this rank's segments go through run_workers and prove_fn is
    called as prove_fn(segment_index, worker_index).  A rank without segments (num_segments < world) only takes part in the barriers.
    Returns (proofs_on_rank0_or_None, whole_job_seconds)."""
    world, rank, _ = env_world()
    mine = assign_segments(num_segments, world, rank)
    barrier()
    if sync_fn:
        sync_fn()
    t0 = time.perf_counter()
    local = {s: prove_fn(s) for s in mine} if workers <= 1 else run_workers(prove_fn, mine, workers)
    if sync_fn:
        sync_fn()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    return (gather_proofs(local) if gather else None), elapsed
