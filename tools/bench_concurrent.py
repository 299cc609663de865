#!/usr/bin/env python3
"""PoseidonStark 262 x 2^log_n proofs with k contexts (one host thread each) on ONE GPU: do the kernels of independent segments
fill each other's gaps (transcript round trips) and overlap HBM-bound NTT passes with VALU-bound hashing?
  python tools/bench_concurrent.py [log_n] [reps]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import zkm_amd  # noqa: E402


def rate(device, log_n, nctx, reps):
    n = 1 << log_n
    ctxs = [zkm_amd.Context(device) for _ in range(nctx)]
    traces = [c.poseidon_trace(100 + i, n, log_n) for i, c in enumerate(ctxs)]
    auxs = [c.alloc(4 * n).upload(np.zeros(4 * n, dtype=np.uint64)) for c in ctxs]
    for c, t, a in zip(ctxs, traces, auxs):
        c.prove_single_table(t, log_n, a, [1, 1])
        c.synchronize()
    start = threading.Barrier(nctx + 1)

    def work(c, t, a):
        start.wait()
        for _ in range(reps):
            c.prove_single_table(t, log_n, a, [1, 1])
        c.synchronize()
    th = [threading.Thread(target=work, args=x) for x in zip(ctxs, traces, auxs)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    for c, t, a in zip(ctxs, traces, auxs):
        t.free(); a.free(); c.close()
    return {"contexts": nctx, "proofs_per_s": nctx * reps / wall, "ms_per_proof_amortised": wall * 1e3 / (nctx * reps)}


if __name__ == "__main__":
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    print(json.dumps([rate(0, log_n, k, reps) for k in (1, 2, 3, 4)], indent=1))
