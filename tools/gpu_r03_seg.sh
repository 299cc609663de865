#!/bin/bash
# development round trip: twelve-table segments (parity + timing)
TAG=${1:-r03_s}
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python -m pytest tests/test_segment.py tests/test_cpu_table.py tests/test_gpu_tables.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/bench_segment.py 16 > $O/seg16.json 2> $O/seg16.err
python tools/bench_segment.py 20 > $O/seg20.json 2> $O/seg20.err
python - <<P
import json
for f in ("seg16", "seg20"):
    s = json.load(open("$O/%s.json" % f))
    print(f, {k: round(s[k], 2) for k in ("ms_per_segment", "kernel_ms_per_segment", "launches_per_segment")}, s["stage_ms"], s["kernel_ms"], s.get("launches"), [(c["contexts"], round(c["segments_per_s"], 1)) for c in s.get("concurrent", [])], [(c["processes"], c["contexts_per_process"], round(c["segments_per_s"], 1)) for c in s.get("concurrent_processes", [])])
P
