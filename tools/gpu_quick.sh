#!/bin/bash
# development aid: GPU parity subset + a single-context and a four-context bench line
mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_large_parity.py tests/test_gpu_prove.py tests/test_gpu_ctl.py -m gpu -q -x > gpurun_out/quick/pytest.log 2>&1; tail -4 gpurun_out/quick/pytest.log
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --contexts 1 > gpurun_out/quick/c1.json 2> gpurun_out/quick/err1
python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/quick/c4.json 2> gpurun_out/quick/err4
python - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/quick/*.json')):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], round(d['value'], 3), round(d['ms_per_step'], 2), d['kernel_ms_per_proof'])
    except Exception as e:
        print(f, 'ERR', e)
P
