#!/bin/bash
mkdir -p gpurun_out/b22
for st in 5 8; do for c in 3 4 5 6 8; do
python bench.py --steps $st --warmup 1 --no-cpu-baseline --no-extras --contexts $c > gpurun_out/b22/s${st}_c$c.json 2>gpurun_out/b22/err_s${st}_c$c
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b22/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],2))
    except Exception as e: print(f,'ERR',e)
P
python - <<'P'
import zkm_amd, numpy as np
c=zkm_amd.Context(0); n=1<<20
t=c.poseidon_trace(1,n,20); a=c.alloc(4*n).upload(np.zeros(4*n,dtype=np.uint64))
c.prove_single_table(t,20,a,[1,1]); c.synchronize(); print("memory live/cached bytes", c.memory())
P
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
