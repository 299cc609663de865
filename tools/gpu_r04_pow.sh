#!/bin/bash
# Proof-of-work search: candidates per round 2^16 / 2^17 / 2^18 (zkm_ctx_set_tuning "pow_round_log"), per-launch time from the profile scope
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export GPU_MAX_HW_QUEUES=16
O=gpurun_out/r04_pow.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_segment.py -m gpu -x -q 2>&1 | tail -3 >> $O
for r in 16 17 18 16 17 18; do
python - $r >> $O 2>&1 <<'PY'
import sys, json
sys.path.insert(0, ".")
import zkm_amd
from tools.bench_segment import segment_rate
c = zkm_amd.Context(0)
c.set_tuning("pow_round_log", int(sys.argv[1]))
o = segment_rate(c, 16, reps=10)
print(json.dumps({"pow_round_log": int(sys.argv[1]), "ms_per_segment": round(o["ms_per_segment"], 2), "pow_ms": o["kernel_ms"].get("fri_pow_search"),
                  "launches": o["launches_per_segment"]}))
PY
done
cat $O
