#!/bin/bash
mkdir -p gpurun_out/b19
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --contexts 1"
$B > gpurun_out/b19/base.json 2>gpurun_out/b19/err0
for v in FT4; do
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_exp_$v.so $B > gpurun_out/b19/$v.json 2>gpurun_out/b19/err_$v
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b19/*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],2), {a:b for a,b in k.items() if 'ntt' in a})
    except Exception as e: print(f,'ERR',e)
P
