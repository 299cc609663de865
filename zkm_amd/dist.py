"""Multi-GPU plumbing: independent segment proofs shard across ranks with NO data-path collective.

The reference proves segments one after another in one process (prover/examples/utils/src/utils.rs:57-68,
105-133); segments are independent proofs, so the MI355X design is one process per GPU, round-robin
assignment, and a host-side gather of the finished proofs (a few hundred KB each).  torch.distributed
(backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) carries only the barrier, the max-over-ranks
time and the proof gather.
"""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the process group from torchrun's environment (no-op for a single process)."""
    world, rank, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
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
    """local: {segment_index: proof ndarray}.  Returns the merged dict on rank 0 (None elsewhere)."""
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
        dist.destroy_process_group()


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
    barrier + sync to a sync + barrier and the slowest rank defines the job time.  gather=False skips collecting the proofs on
    rank 0 (bench.py only needs the time).  workers = k > 1: this rank's segments go through run_workers and prove_fn is
    called as prove_fn(segment_index, worker_index).
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
