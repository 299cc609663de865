#!/usr/bin/env python3
"""Soak: k contexts (one host thread each) prove the SAME twelve-table 2^16-cycle segment r times side by side; every proof must
equal the first one word for word (the prover is deterministic: any difference is a race -- allocator reuse across streams, the
pinned transfer ring, the download flag, the commit lanes).   python tools/soak_segments.py [contexts=8] [reps=30] [stack=1]
stack > 1: every call is a zkm_prove_segments of `stack` segments in lock-step (segment j of a call has public values [1, 2, 3, j]); every
blob of every call must equal the single-segment proof of that transcript."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import zkm_amd  # noqa: E402
from tools.bench_segment import tiled_segment  # noqa: E402

nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
stack = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctxs = [zkm_amd.Context(0) for _ in range(nctx)]
for kv in filter(None, os.environ.get("ZKM_SEG_TUNING", "").split(",")):   # "key=value,...": zkm_ctx_set_tuning on every context
    for c_ in ctxs:
        c_.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
data = [tiled_segment(c, 16) for c in ctxs]
ref, _, _ = ctxs[0].prove_segment(*data[0], public_values=[1, 2, 3])
ref = np.array(ref, copy=True)
refs = [np.array(ctxs[0].prove_segment(*data[0], public_values=[1, 2, 3, j])[0], copy=True) for j in range(stack)] if stack > 1 else []
bad = []


def work(i):
    c, (bufs, logs) = ctxs[i], data[i]
    for r in range(reps):
        if stack > 1:
            got = c.prove_segments([(bufs, logs, [1, 2, 3, j]) for j in range(stack)])
            for j in range(stack):
                if got[j][0].shape != refs[j].shape or not (got[j][0] == refs[j]).all():
                    bad.append((i, r, j))
                    return
            continue
        p, _, _ = c.prove_segment(bufs, logs, public_values=[1, 2, 3])
        if p.shape != ref.shape or not (p == ref).all():
            bad.append((i, r, int(np.nonzero(p != ref)[0][0]) if p.shape == ref.shape else -1))
            return


th = [threading.Thread(target=work, args=(i,)) for i in range(nctx)]
for t in th:
    t.start()
for t in th:
    t.join()
mem = [c.memory() for c in ctxs]
print("soak: %d contexts x %d calls x %d segment(s) per call, %d words each: %s; live - resident bytes per context after the run: %s" % (
    nctx, reps, stack, ref.size, "all equal" if not bad else "MISMATCH %r" % bad[:4], sorted(set(m[0] - c.resident_bytes() for m, c in zip(mem, ctxs)))))
sys.exit(1 if bad else 0)
