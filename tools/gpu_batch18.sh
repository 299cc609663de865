#!/bin/bash
mkdir -p gpurun_out/b18
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --contexts 1"
$B > gpurun_out/b18/base.json 2>gpurun_out/b18/err0
for v in S1W6 S1W4 S2W5; do
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_exp_$v.so $B > gpurun_out/b18/$v.json 2>gpurun_out/b18/err_$v
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b18/*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],2), {a:b for a,b in k.items() if 'ntt' in a})
    except Exception as e: print(f,'ERR',e)
P
