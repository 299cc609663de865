"""A/B of the pipelined auxiliary commitments (zkm_ctx_set_tuning "aux_pipeline"): one 2^16-cycle segment's latency in this process,
tuning from ZKM_SEG_TUNING ("key=value,..."), optional contexts side by side.  usage: auxpipe_ab.py [reps] [nctx]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import zkm_amd
    from tools.bench_segment import tiled_segment
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctxs = [zkm_amd.Context(0) for _ in range(nctx)]
    for kv in filter(None, os.environ.get("ZKM_SEG_TUNING", "").split(",")):
        for c in ctxs:
            c.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    data = [tiled_segment(c, 16) for c in ctxs]
    ref = None
    for c, (bufs, logs) in zip(ctxs, data):
        for _ in range(2):
            p, _, _ = c.prove_segment(bufs, logs, public_values=[1, 2, 3])
        c.synchronize()
        ref = p if ref is None else ref
        assert (p == ref).all()
    times = []
    if nctx == 1:
        c, (bufs, logs) = ctxs[0], data[0]
        for _ in range(reps):
            t0 = time.perf_counter()
            c.prove_segment(bufs, logs, public_values=[1, 2, 3])
            times.append((time.perf_counter() - t0) * 1e3)
        times.sort()
        out = {"ms_min": round(times[0], 2), "ms_median": round(times[len(times) // 2], 2), "ms_max": round(times[-1], 2)}
    else:
        bar = threading.Barrier(nctx + 1)

        def work(c, bufs, logs):
            bar.wait()
            for _ in range(reps):
                c.prove_segment(bufs, logs, public_values=[1, 2, 3])
            c.synchronize()
            bar.wait()
        th = [threading.Thread(target=work, args=(c, b, l)) for c, (b, l) in zip(ctxs, data)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        wall = time.perf_counter() - t0
        for t in th:
            t.join()
        out = {"segments_per_s": round(nctx * reps / wall, 2)}
    out.update(tuning=os.environ.get("ZKM_SEG_TUNING", ""), nctx=nctx, prio=os.environ.get("ZKM_EXPERIMENT_STREAM_PRIORITY", ""),
               sha=__import__("hashlib").sha256(ref.tobytes()).hexdigest()[:16])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
