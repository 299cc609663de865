#!/bin/bash
mkdir -p gpurun_out/b12
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_large_parity.py tests/test_gpu_prove.py tests/test_golden_pipeline.py -m gpu -q -x > gpurun_out/b12/pytest.log 2>&1; tail -5 gpurun_out/b12/pytest.log
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/b12/bench.json 2>gpurun_out/b12/err
python - <<'P'
import json
d=json.load(open('gpurun_out/b12/bench.json')); k=d['kernel_ms_per_proof']
print(round(d['ms_per_step'],2), d['value'], k)
P
