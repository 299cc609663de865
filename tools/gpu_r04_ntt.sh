#!/bin/bash
# tools/gpu_r04_ntt.sh <tag> -- NTT / LDE experiment pass: the parity tests that cover every transform path, then the headline bench
# (no extras) for per-kernel times.
TAG=${1:-r04_b}
O=gpurun_out/$TAG
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 1200 python -m pytest tests/test_gpu_large_parity.py tests/test_gpu_primitives.py tests/test_gpu_prove.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<P
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
print("kernels", d["kernel_ms_per_proof"])
print("ntt", {k: d["roofline_ntt"][k] for k in ("frac", "ms_per_proof")})
P
