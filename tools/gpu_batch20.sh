#!/bin/bash
mkdir -p gpurun_out/b20
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_large_parity.py tests/test_gpu_prove.py -m gpu -q -x > gpurun_out/b20/pytest.log 2>&1; tail -5 gpurun_out/b20/pytest.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --contexts 1"
$B > gpurun_out/b20/new.json 2>gpurun_out/b20/err0
ZKM_NTT_LDE_DIF=1 $B > gpurun_out/b20/old.json 2>gpurun_out/b20/err1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b20/*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],2), {a:b for a,b in k.items() if 'ntt' in a})
    except Exception as e: print(f,'ERR',e)
P
