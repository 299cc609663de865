#!/bin/bash
mkdir -p gpurun_out/b14
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
$B > gpurun_out/b14/base.json 2>gpurun_out/b14/err1
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_nttbf.so $B > gpurun_out/b14/nttbf.json 2>gpurun_out/b14/err2
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b14/*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['ms_per_step'],2), k)
    except Exception as e: print(f,'ERR',e)
P
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/b14/pytest.log 2>&1; tail -3 gpurun_out/b14/pytest.log
