#!/bin/bash
# timing experiments on the NTT passes: variant libraries with barriers / arithmetic / global memory removed (results are wrong; timings only)
mkdir -p gpurun_out/x1
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
$B > gpurun_out/x1/base.json 2>gpurun_out/x1/err0
for v in NOBAR NOCOMPUTE NOMEM; do
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_exp_$v.so $B > gpurun_out/x1/$v.json 2>gpurun_out/x1/err_$v
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/x1/*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['ms_per_step'],2), {a:b for a,b in k.items() if 'ntt' in a})
    except Exception as e: print(f,'ERR',e)
P
