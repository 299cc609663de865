"""Child process of zkm_amd.dist.init: does RCCL work between these ranks on these GPUs?

  python rccl_probe.py <file-store path> <rank> <world> <device> <timeout seconds>

Its own communicator over a file store (no port to collide with the job's rendezvous): init_process_group("nccl", device_id=...),
barrier, all_reduce(MAX) of the rank numbers on a device tensor.  Prints one JSON line; exit code 0 iff the collectives returned what
they must.  The parent kills it on a timeout, so an RCCL hang costs the job the timeout and nothing else.
"""
import datetime
import json
import os
import sys
import time


def main():
    store, rank, world, device, timeout_s = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = {"ok": False, "rank": rank, "world": world, "device": device}
    t0 = time.perf_counter()
    try:
        import torch
        import torch.distributed as dist
        out["torch"] = torch.__version__
        out["import_s"] = round(time.perf_counter() - t0, 3)
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible to the probe")
        torch.cuda.set_device(device)
        dev = torch.device("cuda", device)
        t1 = time.perf_counter()
        dist.init_process_group("nccl", init_method="file://" + store, rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=timeout_s))
        out["init_s"] = round(time.perf_counter() - t1, 3)
        t2 = time.perf_counter()
        dist.barrier(device_ids=[device])
        out["barrier_s"] = round(time.perf_counter() - t2, 3)
        t3 = time.perf_counter()
        t = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize(dev)
        out["all_reduce_s"] = round(time.perf_counter() - t3, 3)
        out["all_reduce_max"] = float(t.item())
        out["ok"] = float(t.item()) == float(world - 1)
        try:
            out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            pass
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        out["error"] = "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0][:300])
    print(json.dumps(out), flush=True)
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
