#!/bin/bash
# k_ntt_small: threads per workgroup (a column of 2^9..2^13 points per workgroup) -- kernel time per segment from the profile scope
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export GPU_MAX_HW_QUEUES=16
O=gpurun_out/r04_smallntt.txt; : > $O
for r in ${SMALL_NTT_VALUES:-1 512 256 128 64 1 256}; do
python - $r >> $O 2>&1 <<'PY'
import sys, json
sys.path.insert(0, ".")
import zkm_amd
from tools.bench_segment import segment_rate
c = zkm_amd.Context(0)
c.set_tuning("small_ntt", int(sys.argv[1]))
o = segment_rate(c, 16, reps=10)
print(json.dumps({"small_ntt": int(sys.argv[1]), "ms_per_segment": round(o["ms_per_segment"], 2), "ntt_small_ms": o["kernel_ms"].get("ntt_small"),
                  "trace_commit_stage_ms": o["stage_ms"].get("compute all trace commitments")}))
PY
done
cat $O
