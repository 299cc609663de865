#!/bin/bash
# GPU suite + bench line + configs 4/5 on the current tree (development round trip)
TAG=${1:-r03_b}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
python bench.py --steps 8 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<P
import json
try:
    d = json.load(open("$O/bench.json"))
    print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
    print("kernel", d["kernel_ms_per_proof"])
    print("stage", d.get("stage_ms"))
    print("host", d.get("host_resident"))
    s = d.get("segment_2_16", {})
    print("seg16", {k: s.get(k) for k in ("ms_per_segment", "kernel_ms_per_segment", "wall_over_kernel_sum", "launches_per_segment")}, s.get("kernel_ms"), [(c["contexts"], round(c["segments_per_s"], 1)) for c in s.get("concurrent", [])])
    f = d.get("fri_2_22", {})
    print("fri", f.get("ms"), f.get("kernel_ms"), f.get("launches"))
    k = d.get("keccak_sponge_2_20", {})
    print("sponge", {x: k.get(x) for x in ("witness_ms", "witness_kernel_ms", "witness_permutations_per_s", "commit_ms")}, k.get("keccakf_batch"))
    print({x: d[x] for x in d if x.endswith("_error")})
except Exception as e:
    print("bench parse error", e)
P
