#!/usr/bin/env python3
"""Whole-segment timing: zkm_prove_segment (prove_with_traces, prover.rs:130-232, on all twelve tables with the fifteen
cross-table lookups) with wall time next to the per-kernel HIP-event profile.

The traces are the committed test segment (tests/golden/segment12.npz) tiled to the table heights of a 2^16-cycle segment
(the reference's default segment size, emulator/src/utils.rs:6) or of a 2^20-cycle one -- rows repeat, so the witness is not
valid across the seams; the prover does not care, the numbers are timings only.  No oracle, no fixture builders.

  python tools/bench_segment.py [16|20] [single]   prints one JSON object (single: without the concurrent-context runs, e.g. under rocprofv3)
  from tools.bench_segment import small_segment_rate   (bench.py's `segment_2_16` key)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

# table heights (log2 rows) in Table::all() order: Arithmetic, Cpu, Poseidon, PoseidonSponge, Keccak, KeccakSponge, ShaExtend,
# ShaExtendSponge, ShaCompress, ShaCompressSponge, Logic, Memory -- a CPU table of 2^k cycles with the precompile mix of the test segment
HEIGHTS = {16: [16, 16, 10, 10, 11, 8, 12, 12, 12, 6, 13, 17],
           20: [19, 20, 14, 14, 15, 12, 16, 16, 16, 10, 17, 21]}


def tiled_segment(ctx, log_cycles, shrink=0):
    """shrink = k: the precompile tables (everything but Arithmetic, Cpu, Memory) are 2^k times shorter (never below the test segment's own
    heights) -- segments of one program differ in how much hashing they do."""
    from zkm_amd import tables as T
    seg = np.load(os.path.join(ROOT, "tests", "golden", "segment12.npz"))
    base = [int(x) for x in seg["log_n"]]
    bufs, logs = [], []
    for i in range(12):
        w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
        L = max(HEIGHTS[log_cycles][i] - (shrink if i not in (0, 1, 11) else 0), base[i])
        t = np.ascontiguousarray(np.tile(seg["t%d" % i].reshape(w, -1), (1, 1 << (L - base[i])))).reshape(-1)
        bufs.append(ctx.alloc(t.size).upload(t))
        logs.append(L)
    return bufs, logs


def segment_rate(ctx, log_cycles, reps=3):
    bufs, logs = tiled_segment(ctx, log_cycles)
    ctx.prove_segment(bufs, logs, public_values=[1, 2, 3])       # warm-up: allocator, twiddles, power tables
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        proofs, _, offs = ctx.prove_segment(bufs, logs, public_values=[1, 2, 3])
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps                     # the latency: profiler off (its ~1000 event records per segment cost host time)
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.prove_segment(bufs, logs, public_values=[1, 2, 3])
    ctx.synchronize()
    wall_profiled = (time.perf_counter() - t0) / reps
    rec = ctx.profile_records()
    ctx.profile(False)
    for b in bufs:
        b.free()
    krec = {k: v for k, v in rec.items() if not k.startswith("stage/")}      # kernel records; "stage/..." = the reference's timed! scopes
    kernel_ms = sum(v[1] for v in krec.values()) / reps
    return {"cpu_cycles_log2": log_cycles, "table_heights_log2": logs, "segments_per_s": 1.0 / wall, "ms_per_segment": wall * 1e3,
            "ms_per_segment_profiler_on": wall_profiled * 1e3,
            "kernel_ms_per_segment": kernel_ms, "wall_over_kernel_sum": wall * 1e3 / kernel_ms,
            "launches_per_segment": sum(v[0] for v in krec.values()) / reps, "proof_words": int(offs[12]),
            "kernel_ms": {k: round(v[1] / reps, 3) for k, v in sorted(krec.items(), key=lambda kv: -kv[1][1])[:12]},
            "launches": {k: round(v[0] / reps, 1) for k, v in sorted(krec.items(), key=lambda kv: -kv[1][0])[:12]},
            "stage_ms": {k[6:]: round(v[1] / reps, 3) for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1]) if k.startswith("stage/")},
            "note": "twelve tables, fifteen lookups, one zkm_prove_segment call; traces device-resident, tiled test segment (timing only). The trace "
                    "commitments, and after the CTL challenges the CTL data + auxiliary commitments, of the twelve tables run side by side on the "
                    "context's commit lanes: kernel_ms sums launches that overlapped (each counts its own duration), and the per-table stage "
                    "scopes of those two phases overlap too -- wall_over_kernel_sum below 1 means concurrency, not a faster clock"}


def concurrent_segment_rate(device, log_cycles, nctx, reps=6, tuning=None, cu_parts=0):
    """`nctx` host threads, each with its OWN context (own stream, allocator, transcript) on the same GPU, proving independent
    segments at the same time: in the launch-bound regime of small segments the GPU interleaves their kernels, so the per-level
    Merkle / per-layer FRI latencies of one segment are filled with the work of the others.  Segments are independent proofs
    (prover/examples/utils/src/utils.rs:57-68), so this is the same sharding as across GPUs, applied within one."""
    import threading
    import zkm_amd
    ctxs = []
    for i in range(nctx):
        if cu_parts:                                   # measurement aid: context i confined to part i % cu_parts of the CUs (csrc/core.hip)
            os.environ["ZKM_CU_MASK_PART"] = "%d/%d" % (i % cu_parts, cu_parts)
        ctxs.append(zkm_amd.Context(device))
    os.environ.pop("ZKM_CU_MASK_PART", None)
    for c in ctxs:
        for k, v in (tuning or {}).items():
            c.set_tuning(k, v)
    data = [tiled_segment(c, log_cycles) for c in ctxs]
    for c, (bufs, logs) in zip(ctxs, data):
        c.prove_segment(bufs, logs, public_values=[1, 2, 3])
        c.synchronize()
    start = threading.Barrier(nctx + 1)

    def work(c, bufs, logs):
        start.wait()
        for _ in range(reps):
            c.prove_segment(bufs, logs, public_values=[1, 2, 3])
        c.synchronize()
    th = [threading.Thread(target=work, args=(c, b, l)) for c, (b, l) in zip(ctxs, data)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    for c, (bufs, _) in zip(ctxs, data):
        for b in bufs:
            b.free()
        c.close()
    return {"contexts": nctx, "segments_per_context": reps, "segments_per_s": nctx * reps / wall, "ms_per_segment_amortised": wall * 1e3 / (nctx * reps)}


def lockstep_segment_rate(device, log_cycles, nctx, stack, reps=3, tuning=None, host=False, ragged=0):
    """zkm_prove_segments: `nctx` host threads with one context each, every call proving `stack` independent segments in LOCK-STEP
    (one launch per stage for all of them; include/zkm_hip.h).  The segments of a call share the tiled traces in HBM and differ in
    their public values, so every segment has its own transcript, challenges and proof."""
    import threading
    import zkm_amd
    if os.environ.get("ZKM_SLEEPING_WAITS") == "1":     # host waits that sleep (with tuning block_after_us=0): before the first context
        import ctypes
        ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(ctypes.c_uint(4))       # hipDeviceScheduleBlockingSync
    ctxs = [zkm_amd.Context(device) for _ in range(nctx)]
    for c in ctxs:
        for k, v in (tuning or {}).items():
            c.set_tuning(k, v)
    data = [tiled_segment(c, log_cycles) for c in ctxs]
    pinned = []
    if host:
        # the deployed input: traces in (pinned) HOST memory, as the CPU witness generator leaves them (generation/mod.rs:25-76) -- every
        # call uploads its segments' ~0.3 GB each inside the clock; the contexts' uploads overlap each other's kernels
        for i, (c, (bufs, logs)) in enumerate(zip(ctxs, data)):
            hb = []
            for b_ in bufs:
                h = c.pinned_array(b_.words)
                h[:] = b_.download()
                b_.free()
                hb.append(h)
            pinned.append((c, hb))
            data[i] = (hb, logs)
    segs = [[(bufs, logs, [1, 2, 3, i, j]) for j in range(stack)] for i, (bufs, logs) in enumerate(data)]
    extra = []
    if ragged and not host:
        # RAGGED calls: segment j of a call has its precompile tables 2^(j % ragged) times shorter -- every such table then forms `ragged`
        # groups (one per height) instead of one
        for i, c in enumerate(ctxs):
            variants = [data[i]] + [tiled_segment(c, log_cycles, shrink=k) for k in range(1, ragged)]
            extra.append(variants[1:])
            segs[i] = [(variants[j % ragged][0], variants[j % ragged][1], [1, 2, 3, i, j]) for j in range(stack)]
    for c, sg in zip(ctxs, segs):
        c.prove_segments(sg)                      # warm-up: allocator, twiddles, power tables
        c.synchronize()
    start = threading.Barrier(nctx + 1)

    stage_ms = []           # (staged mode: host milliseconds of the stage calls / the prove call / the frees of every call)
    staged = host == 2      # the NEXT call's traces staged behind the current call (zkm_trace_stage: cross-call pipelining of the uploads)

    def stage_call(c, sg):      # one zkm_segment_stage per segment: (device pointers of its twelve tables, heights, public values, handle)
        out = []
        for bufs, logs, pub in sg:
            out.append((c.stage_segment(bufs, logs), logs, pub))
        return out

    def as_segments(call):      # (tables() orders the compute stream behind the upload: only when the call's turn has come)
        return [(st.tables(), logs, pub) for st, logs, pub in call]

    def free_call(call):
        for st, _, _ in call:
            st.free()
    if staged:
        for c, sg in zip(ctxs, segs):              # warm-up of this path: two calls' worth of staged blocks alive at once
            a, b = stage_call(c, sg), stage_call(c, sg)
            c.prove_segments(as_segments(a))
            c.prove_segments(as_segments(b))
            free_call(a)
            free_call(b)
            c.synchronize()
    first = [stage_call(c, sg) for c, sg in zip(ctxs, segs)] if staged else None
    if staged:
        for call in first:
            for st, _, _ in call:
                st.ready(wait=True)

    def work(c, sg, k=0):
        start.wait()
        if staged:                                  # every call of the timed region stages exactly one call (the last one's is waited for)
            cur = first[k]
            for _ in range(reps):
                t_a = time.perf_counter()
                nxt = stage_call(c, sg)
                t_b = time.perf_counter()
                c.prove_segments(as_segments(cur))
                t_c = time.perf_counter()
                free_call(cur)
                stage_ms.append(((t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (time.perf_counter() - t_c) * 1e3))
                cur = nxt
            free_call(cur)
        else:
            for _ in range(reps):
                c.prove_segments(sg)
        c.synchronize()
    th = [threading.Thread(target=work, args=(c, sg, k)) for k, (c, sg) in enumerate(zip(ctxs, segs))]
    for t in th:
        t.start()
    start.wait()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    mem = [c.memory() for c in ctxs]
    for c, hb in pinned:
        for h in hb:
            c.free_pinned(h)
    for vs in extra:
        for bufs, _ in vs:
            for b in bufs:
                b.free()
    for c, (bufs, _) in zip(ctxs, data):
        if not host:
            for b in bufs:
                b.free()
        c.close()
    total = nctx * stack * reps
    return {"contexts": nctx, "segments_per_call": stack, "calls_per_context": reps, "segments_per_s": total / wall, "traces": ("pinned host memory, next call staged behind the current one" if host == 2 else "pinned host memory") if host else "HBM", "ragged_heights": ragged,
            "ms_per_segment_amortised": wall * 1e3 / total, "ms_per_call": wall * 1e3 / reps, "cpu_seconds_per_segment": cpu_s / total,
            "host_cpus_busy": cpu_s / wall, "host_waits": "sleeping" if os.environ.get("ZKM_SLEEPING_WAITS") == "1" else "polling",
            "tuning": tuning or {},
            "memory_live_cached_GB": [round((m[0] + m[1]) / 2**30, 2) for m in mem],
            **({"staged_host_ms_per_call": {"stage": round(sum(x[0] for x in stage_ms) / len(stage_ms), 2), "prove": round(sum(x[1] for x in stage_ms) / len(stage_ms), 1),
                                            "free": round(sum(x[2] for x in stage_ms) / len(stage_ms), 2)}} if stage_ms else {})}


def multi_process_rate(procs, nctx, reps=8, tuning=""):
    """The same with the contexts in `procs` fresh host processes (tools/bench_segment_procs.py): every process has its own HIP runtime,
    so launches of different processes do not queue behind one another on the host; 1 x k is the deployment shape itself -- a prover
    process with k contexts and nothing else in it (the in-process `concurrent` figures share bench.py's process with its four
    headline contexts and torch)."""
    import subprocess
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(max(2, 16 // procs)), ZKM_SEG_TUNING=tuning)   # ~16 hardware queues on the GPU in total
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_segment_procs.py"), str(procs), str(nctx), str(reps)],
                       capture_output=True, text=True, timeout=300, env=env)
    if r.returncode != 0:
        raise RuntimeError("bench_segment_procs failed: " + r.stderr[-400:])
    return json.loads(r.stdout.strip().splitlines()[-1])


THROUGHPUT = {"throughput_profile": 1}   # zkm_ctx_set_tuning: one stream per context, latency forms of the permutation for tiny launches only


def throughput_rates(device, contexts=(8, 12, 16)):
    """Many contexts per GPU in the throughput profile (include/zkm_hip.h "throughput_profile"; profiles/r04_throughput_profile.txt), each
    measurement in a FRESH process with nothing else in it: the profile is about the number of streams a process maps onto the runtime's
    hardware queues, and the contexts of the calling process (bench.py: four headline contexts and their lanes) would count too."""
    out = []
    for k in contexts:
        r = multi_process_rate(1, k, reps=5, tuning="throughput_profile=1")
        r["profile"] = "throughput (commit_lanes 1, wide_max_hashes 256, quad_max_hashes 4096), fresh process"
        out.append(r)
    return out


def lockstep_rates(device, shapes=((4, 8), (8, 8)), reps=3, sleeping_shape=(8, 8)):
    """zkm_prove_segments (K segments per call in lock-step; include/zkm_hip.h), every shape in a FRESH process of its own (one
    tools/sweep_lockstep.py run per shape: a shape measured after another one in the same process inherits its runtime state -- 8 x 8 after
    16 x 8 read 98 segments/s instead of 108): contexts x segments per call, throughput profile.  With the VALU count of a lock-step segment
    from profiles/lockstep_valu_latest.json (rocprofv3 --pmc SQ_INSTS_VALU pass of tools/gpu_lockstep_prof.sh, quoted while its code
    fingerprint matches) every rate also says which fraction of the segment's own VALU-issue budget it is: instructions / (SIMDs x the leaf
    kernel's measured issue rate).  (16 x 8 reaches 111.7 segments/s = 0.95 of the budget in a process of its own and 100.7 as a child of
    bench.py -- sixteen polling threads are the box's whole 16-CPU quota -- so it is not one of the default shapes.)"""
    import subprocess
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", ZKM_BENCH_DEVICE=str(device))
    specs = ["%d,%d,throughput_profile=1,reps=%d" % (g, k, reps) for g, k in shapes]
    # ... and one shape once more with SLEEPING host waits (blocking-sync device flag + block_after_us 0): the rate next to the host
    # CPUs it keeps busy (`host_cpus_busy`; polling: one per context)
    specs.append("%d,%d,throughput_profile=1,reps=%d,sleeping=1" % (sleeping_shape[0], sleeping_shape[1], reps))
    out = []
    for spec in specs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_lockstep.py"), spec], capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0:
            raise RuntimeError("sweep_lockstep %s failed: %s" % (spec, r.stderr[-400:]))
        out += [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{")]
    try:
        budget = json.load(open(os.path.join(ROOT, "profiles", "lockstep_valu_latest.json")))
        from bench import code_fingerprint
        if budget.get("code_fingerprint") == code_fingerprint():
            for o in out:
                if "ms_per_segment_amortised" in o:
                    o["valu_budget_ms_per_segment"] = budget["budget_ms_per_segment"]
                    o["valu_budget_frac"] = budget["budget_ms_per_segment"] / o["ms_per_segment_amortised"]
            out.append({"valu_budget": budget})
        else:
            out.append({"valu_budget": "profiles/lockstep_valu_latest.json is stale for this code (fingerprint mismatch) -- rerun tools/gpu_lockstep_prof.sh"})
    except Exception as e:  # noqa: BLE001
        out.append({"valu_budget_error": str(e)})
    return out


def small_segment_rate(ctx, device=0):
    out = segment_rate(ctx, 16)
    try:
        out["lockstep"] = lockstep_rates(device)
        out["concurrent"] = [concurrent_segment_rate(device, 16, k) for k in (4, 8)]
        out["concurrent_throughput_profile"] = throughput_rates(device, contexts=(16,))
    except Exception as e:  # the single-context figure stands on its own
        out["concurrent_error"] = str(e)
    return out


if __name__ == "__main__":
    import zkm_amd
    c = zkm_amd.Context(0)
    lc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    out = segment_rate(c, lc)
    out["memory_live_cached"] = c.memory()
    if lc == 16 and "single" not in sys.argv[2:]:
        out["lockstep"] = lockstep_rates(0)
        out["concurrent"] = [concurrent_segment_rate(0, 16, k) for k in (2, 4, 8)]
        out["concurrent_throughput_profile"] = throughput_rates(0)
        out["concurrent_processes"] = [multi_process_rate(p, k) for p, k in ((1, 8), (2, 4), (4, 2))]
    print(json.dumps(out, indent=1))
