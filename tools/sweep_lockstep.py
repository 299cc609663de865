#!/usr/bin/env python3
"""Sweep of zkm_prove_segments on 2^16-cycle twelve-table segments: contexts x segments per call x tunings -> segments/s.
   python tools/sweep_lockstep.py "G,K[,key=value...]" ...     one JSON line per configuration
   keys: any zkm_ctx_set_tuning key, reps=N, host=1 (traces in pinned host memory, uploaded inside the call) / host=2 (the next call staged behind the current one), ragged=k, sleeping=1 (blocking-sync waits; last spec only)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from tools.bench_segment import lockstep_segment_rate  # noqa: E402

for spec in sys.argv[1:]:
    parts = spec.split(",")
    g, k = int(parts[0]), int(parts[1])
    tuning = {}
    reps = 3
    host = False
    ragged = 0
    for kv in parts[2:]:
        key, v = kv.split("=")
        if key == "reps":
            reps = int(v)
        elif key == "host":
            host = int(v)            # 1: uploads inside the call; 2: the next call's traces staged behind the current call
        elif key == "ragged":
            ragged = int(v)
        elif key == "sleeping":        # host waits that sleep: the device flag is process-wide, so put such a spec LAST
            if int(v):
                os.environ["ZKM_SLEEPING_WAITS"] = "1"
                tuning["block_after_us"] = 0
        else:
            tuning[key] = int(v)
    try:
        r = lockstep_segment_rate(int(os.environ.get("ZKM_BENCH_DEVICE", "0")), 16, g, k, reps=reps, tuning=tuning, host=host, ragged=ragged)
    except Exception as e:  # keep sweeping
        r = {"contexts": g, "segments_per_call": k, "tuning": tuning, "error": str(e)[:300]}
    print(json.dumps(r), flush=True)
