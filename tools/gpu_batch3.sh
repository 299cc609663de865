#!/bin/bash
mkdir -p gpurun_out/b3
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_large_parity.py tests/test_gpu_prove.py tests/test_golden_pipeline.py -m gpu -q -x > gpurun_out/b3/pytest.log 2>&1
tail -15 gpurun_out/b3/pytest.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b3/bench.json 2> gpurun_out/b3/bench.err
ZKM_NTT_LDE_3PASS=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/b3/bench_3pass.json 2> gpurun_out/b3/bench_3pass.err
timeout 300 python tools/bench_segment.py 16 > gpurun_out/b3/seg16.json 2>&1
python - <<'P'
import json
for f in ('bench','bench_3pass'):
    try:
        d=json.load(open('gpurun_out/b3/%s.json'%f))
        print(f, d['value'], d['ms_per_step'], d.get('host_resident_ms_per_step'))
        print(d['kernel_ms_per_proof'])
        if 'segment_2_16' in d: print({k:v for k,v in d['segment_2_16'].items() if k not in('note','table_heights_log2')})
    except Exception as e: print(f, 'ERR', e)
P
tail -30 gpurun_out/b3/seg16.json
