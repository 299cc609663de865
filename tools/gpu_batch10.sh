#!/bin/bash
mkdir -p gpurun_out/b10
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
$B > gpurun_out/b10/base.json 2>gpurun_out/b10/err1
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_sg4.so $B > gpurun_out/b10/sg4.json 2>gpurun_out/b10/err2
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_sg4.so ZKM_NTT_S2=13 $B > gpurun_out/b10/sg4_s13.json 2>gpurun_out/b10/err3
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_sg4.so timeout 300 python -m pytest tests/test_gpu_large_parity.py -m gpu -q -x -k "commit" > gpurun_out/b10/pytest.log 2>&1; tail -2 gpurun_out/b10/pytest.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b10/*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['ms_per_step'],2), {x:k[x] for x in k if x.startswith('ntt')})
    except Exception as e: print(f,'ERR',e)
P
