#!/usr/bin/env python3
"""2^16-cycle segments, P processes x K contexts on ONE GPU: does the aggregate rate of small segments depend on how the contexts
are spread over host processes (one HIP runtime, one launch lock per process)?

  python tools/bench_segment_procs.py P K [reps]      prints one JSON object
"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, nctx, reps, start, done, out):
    import threading
    import zkm_amd
    from tools.bench_segment import tiled_segment
    ctxs = [zkm_amd.Context(int(os.environ.get("ZKM_BENCH_DEVICE", "0"))) for _ in range(nctx)]
    for kv in filter(None, os.environ.get("ZKM_SEG_TUNING", "").split(",")):      # "key=value,...": zkm_ctx_set_tuning on every context
        for c in ctxs:
            c.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    data = [tiled_segment(c, 16) for c in ctxs]
    for c, (bufs, logs) in zip(ctxs, data):
        c.prove_segment(bufs, logs, public_values=[1, 2, 3])
        c.synchronize()
    inner = threading.Barrier(nctx + 1)

    def work(c, bufs, logs):
        inner.wait()
        for _ in range(reps):
            c.prove_segment(bufs, logs, public_values=[1, 2, 3])
        c.synchronize()
    th = [threading.Thread(target=work, args=(c, b, l)) for c, (b, l) in zip(ctxs, data)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    inner.wait()
    for t in th:
        t.join()
    out.put((rank, t0, time.perf_counter()))
    done.wait()


if __name__ == "__main__":
    P, K = int(sys.argv[1]), int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    mp.set_start_method("spawn")
    start, done, out = mp.Barrier(P), mp.Barrier(P), mp.Queue()
    ps = [mp.Process(target=worker, args=(r, K, reps, start, done, out)) for r in range(P)]
    for p in ps:
        p.start()
    res = [out.get() for _ in range(P)]
    for p in ps:
        p.join()
    wall = max(r[2] for r in res) - min(r[1] for r in res)     # perf_counter is CLOCK_MONOTONIC: comparable across processes
    print(json.dumps({"processes": P, "contexts_per_process": K, "segments": P * K * reps, "segments_per_s": P * K * reps / wall}))
