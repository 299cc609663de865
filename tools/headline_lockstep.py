#!/usr/bin/env python3
"""BASELINE config 2 (PoseidonStark 262 x 2^20, full prove_single_table incl. trace commitment, traces resident in HBM) with K proofs per
call in lock-step (zkm_prove_single_tables) from G contexts: proofs per second.   python tools/headline_lockstep.py "G,K[,reps]" ..."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import zkm_amd  # noqa: E402
log_n = 20
n = 1 << log_n
for spec in sys.argv[1:]:
    p = [int(x) for x in spec.split(",")]
    G, K, reps = p[0], p[1], (p[2] if len(p) > 2 else 4)
    ctxs = [zkm_amd.Context(int(os.environ.get("ZKM_BENCH_DEVICE", "0"))) for _ in range(G)]
    traces = [[c.poseidon_trace(100 + i * K + k, n, log_n) for k in range(K)] for i, c in enumerate(ctxs)]
    aux = ctxs[0].alloc(4 * n).upload(np.zeros(4 * n, dtype=np.uint64))

    def call(i):
        if K == 1:
            return [ctxs[i].prove_single_table(traces[i][0], log_n, aux, [1, 1])]
        return ctxs[i].prove_single_tables(traces[i], log_n, aux, [1, 1])
    for i in range(G):
        call(i)
        ctxs[i].synchronize()
    start = threading.Barrier(G + 1)

    def work(i):
        start.wait()
        for _ in range(reps):
            call(i)
        ctxs[i].synchronize()
    th = [threading.Thread(target=work, args=(i,)) for i in range(G)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("contexts %d x proofs per call %d: %.3f proofs/s (%.2f ms per proof)" % (G, K, G * K * reps / dt, dt * 1e3 / (G * K * reps)), flush=True)
    for c, ts in zip(ctxs, traces):
        for t in ts:
            t.free()
    aux.free()
    for c in ctxs:
        c.close()
