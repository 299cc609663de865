#!/bin/bash
mkdir -p gpurun_out/b23
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_large_parity.py tests/test_gpu_prove.py tests/test_keccak_sponge.py -m gpu -q -x > gpurun_out/b23/pytest.log 2>&1; tail -5 gpurun_out/b23/pytest.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --contexts 1"
$B > gpurun_out/b23/new.json 2>gpurun_out/b23/err0
python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/b23/new_c4.json 2>gpurun_out/b23/err2
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b23/*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],2), {a:b for a,b in k.items() if 'merkle' in a or 'ntt' in a})
    except Exception as e: print(f,'ERR',e)
P
