#!/usr/bin/env python3
"""tools/stage_probe.py -- where does a staged (cross-proof pipelined) 2^20-row proof lose time?  One context, then four:
device-resident proofs, staged proofs, and how long the NEXT trace's upload still needs when the current proof returns."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkm_amd
log_n, W = 20, 262
n = 1 << log_n
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = 6
ctxs = [zkm_amd.Context(0) for _ in range(nctx)]
dev = ctxs[0].poseidon_trace(100, n, log_n)
host = ctxs[0].pinned_array(W * n)
host[:] = dev.download()
aux = ctxs[0].alloc(4 * n).upload(np.zeros(4 * n, dtype=np.uint64))
for c in ctxs:
    c.prove_single_table(dev, log_n, aux, [1, 1])
    st = c.stage_trace(host, W, log_n)
    st2 = c.stage_trace(host, W, log_n)          # two staged blocks live at once, as in the loop below: both come from the cache afterwards
    c.prove_single_table(st, log_n, aux, [1, 1])
    c.prove_single_table(st2, log_n, aux, [1, 1])
    st.free()
    st2.free()

def run(fn):
    th = [threading.Thread(target=fn, args=(w,)) for w in range(nctx)]
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in ctxs:
        c.synchronize()
    return (time.perf_counter() - t0) / (nctx * reps) * 1e3

def resident(w):
    for _ in range(reps):
        ctxs[w].prove_single_table(dev, log_n, aux, [1, 1])
waits = []
def staged(w):
    cur = ctxs[w].stage_trace(host, W, log_n)
    cur.ready(wait=True)
    for _ in range(reps):
        t0 = time.perf_counter()
        nxt = ctxs[w].stage_trace(host, W, log_n)
        t1 = time.perf_counter()
        ctxs[w].prove_single_table(cur, log_n, aux, [1, 1])
        t2 = time.perf_counter()
        cur.free()
        nxt.ready(wait=True)
        t3 = time.perf_counter()
        waits.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 1), round((t3 - t2) * 1e3, 1)))
        cur = nxt
    cur.free()
def upload_only(w):
    for _ in range(reps):
        st = ctxs[w].stage_trace(host, W, log_n)
        st.ready(wait=True)
        st.free()
print("contexts", nctx, "piece", os.environ.get("ZKM_STAGE_PIECE_COLS", "8 (default)"))
print("  device-resident  %.1f ms/proof" % run(resident))
ms = run(upload_only)
print("  upload only      %.1f ms/trace = %.1f GB/s aggregate" % (ms, W * n * 8 / ms / 1e6))
print("  staged           %.1f ms/proof" % run(staged))
print("  (stage call ms, prove ms, wait for next upload after the proof ms):", waits[:12])
print("  device-resident  %.1f ms/proof" % run(resident))

# ---- which host memory?  pinned (zkm_host_alloc) above; here: pageable (what a fresh Rust Vec is) and pageable + zkm_host_register
import ctypes as C
c = ctxs[0]
pageable = np.empty(W * n, dtype=np.uint64)
pageable[:] = host
def stage_time(arr, label):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        st = c.stage_trace(arr, W, log_n)
        t1 = time.perf_counter()
        st.ready(wait=True)
        t2 = time.perf_counter()
        st.free()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print("  %-34s stage call returns after %.1f ms, uploaded after %.1f ms (%.1f GB/s)" % (label, ts[-1][0], ts[-1][1], W * n * 8 / ts[-1][1] / 1e6))
stage_time(host, "pinned (zkm_host_alloc)")
stage_time(pageable, "pageable")
err = C.c_char_p()
t0 = time.perf_counter()
rc = c.L.zkm_host_register(c.h, pageable.ctypes.data, pageable.nbytes, C.byref(err))
t_reg = (time.perf_counter() - t0) * 1e3
print("  zkm_host_register of 2.2 GB: rc %d, %.1f ms" % (rc, t_reg))
if rc == 0:
    stage_time(pageable, "pageable + zkm_host_register")
    t0 = time.perf_counter()
    c.L.zkm_host_unregister(c.h, pageable.ctypes.data)
    print("  zkm_host_unregister: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
