#!/bin/bash
mkdir -p gpurun_out/b2
python -m pytest tests/test_segment.py tests/test_gpu_large_parity.py tests/test_reference_fixtures.py tests/test_gpu_primitives.py tests/test_gpu_prove.py -m gpu -q -x > gpurun_out/b2/pytest.log 2>&1
tail -5 gpurun_out/b2/pytest.log
python bench.py --steps 5 --warmup 1 > gpurun_out/b2/bench.json 2> gpurun_out/b2/bench.err
tail -3 gpurun_out/b2/bench.err
python tools/bench_segment.py 16 > gpurun_out/b2/seg16.json 2>&1
python tools/bench_segment.py 20 > gpurun_out/b2/seg20.json 2>&1
python - <<'P'
import json
d=json.load(open('gpurun_out/b2/bench.json'))
print(d['value'], d['ms_per_step'], d.get('host_resident_ms_per_step'), d.get('host_resident_error'), d.get('segment_2_16_error'))
print(d['kernel_ms_per_proof'])
print(d['roofline']['valu_issue'])
print({k:v for k,v in d.get('segment_2_16',{}).items() if k!='kernel_ms'})
P
