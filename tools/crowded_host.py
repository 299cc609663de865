#!/usr/bin/env python3
"""What a rank gets on the 8-GPU node: 1/8 of the host's cores (zkm_amd/dist.py pin_to_gpu), 8 contexts x 4 commit lanes waiting for
transcript round trips.  Measures 2^16-cycle segments/s and the CPU seconds burnt per segment for three waiting policies
(zkm_ctx_set_tuning "block_after_us"): spin-then-block-when-crowded (default 50), spin only (10^9), always block (0) -- with the
process confined to 256/8 = 32, 8 and 4 CPUs, and unconfined.

  python tools/crowded_host.py            -> JSON lines + a table (profiles/r04_crowded_host.txt)
  python tools/crowded_host.py one <ncpus> <block_after_us> <nctx>     (child: one configuration in a fresh process)
"""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def one(ncpus, block_after_us, nctx):
    allowed = sorted(os.sched_getaffinity(0))
    if ncpus:
        os.sched_setaffinity(0, allowed[:ncpus])            # before the library caches the count and before any thread starts
    from tools.bench_segment import concurrent_segment_rate
    r0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    out = concurrent_segment_rate(0, 16, nctx, reps=6, tuning={"block_after_us": block_after_us})
    r1, wall = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter() - t0
    segs = out["contexts"] * out["segments_per_context"]
    out.update({"cpus_allowed": len(os.sched_getaffinity(0)), "block_after_us": block_after_us,
                "cpu_s_per_segment_incl_setup": ((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)) / (segs + out["contexts"]),
                "wall_s_incl_setup": wall})
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    host = len(os.sched_getaffinity(0))
    rows = []
    for ncpus in (0, max(4, host // 8), 8, 4):
        for policy in (50, 10**9, 0):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(ncpus), str(policy), "8"], capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                rows.append({"cpus_allowed": ncpus or host, "block_after_us": policy, "error": r.stderr[-300:]})
                continue
            rows.append(json.loads(line[-1]))
            print(line[-1], flush=True)
    print("\n%-14s %-22s %14s %22s" % ("cpus allowed", "policy", "segments/s", "CPU s per segment"))
    name = {50: "spin 50 us, then block*", 10**9: "spin only", 0: "always block"}
    for r in rows:
        if "error" in r:
            print("%-14s %-22s ERROR %s" % (r["cpus_allowed"], name[r["block_after_us"]], r["error"]))
        else:
            print("%-14d %-22s %14.1f %22.3f" % (r["cpus_allowed"], name[r["block_after_us"]], r["segments_per_s"], r["cpu_s_per_segment_incl_setup"]))
    print("* blocks only while the process is crowded: more than half as many threads waiting as CPUs allowed (csrc/core.hip wait_flag)")
