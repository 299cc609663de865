#!/bin/bash
mkdir -p gpurun_out/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --contexts 1"
$B > gpurun_out/tmp/base.json 2>gpurun_out/tmp/e0
ZKM_NTT_TILE=4096 $B > gpurun_out/tmp/tile4096.json 2>gpurun_out/tmp/e1
ZKM_NTT_TILE=1024 $B > gpurun_out/tmp/tile1024.json 2>gpurun_out/tmp/e2
python - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/tmp/*.json')):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'], 2), {a: b for a, b in d['kernel_ms_per_proof'].items() if 'ntt' in a})
    except Exception as e: print(f, 'ERR', e)
P
