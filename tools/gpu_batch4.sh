#!/bin/bash
mkdir -p gpurun_out/b4
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
timeout 600 python -m pytest tests/test_gpu_large_parity.py tests/test_gpu_primitives.py -m gpu -q -x -k "commit or ntt" > gpurun_out/b4/pytest.log 2>&1; tail -3 gpurun_out/b4/pytest.log
$B > gpurun_out/b4/occ4_s13.json 2>gpurun_out/b4/err1
ZKM_NTT_S2=12 $B > gpurun_out/b4/occ4_s12.json 2>gpurun_out/b4/err2
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_occ2.so $B > gpurun_out/b4/occ2_s13.json 2>gpurun_out/b4/err3
ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_occ2.so ZKM_NTT_S2=12 $B > gpurun_out/b4/occ2_s12.json 2>gpurun_out/b4/err4
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b4/occ*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms_per_proof']
        print(f.split('/')[-1], round(d['ms_per_step'],2), {x:k[x] for x in k if x.startswith('ntt')})
    except Exception as e: print(f,'ERR',e)
P
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/b4/trace -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /root/repo/gpurun_out/b4/trace.log 2>&1
cd /root/repo; find gpurun_out/b4/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/b4/kernel_stats.csv \;
head -25 gpurun_out/b4/kernel_stats.csv | cut -c1-150
