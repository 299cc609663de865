#!/bin/bash
mkdir -p gpurun_out/b15
for c in 1 2 4; do
python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --contexts $c > gpurun_out/b15/c$c.json 2>gpurun_out/b15/err$c
done
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/b15/default5.json 2>gpurun_out/b15/err_d
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/b15/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],2), d['single_context']['ms_per_step'], d['roofline']['frac'] if d.get('roofline') else None)
    except Exception as e: print(f,'ERR',e)
P
tail -3 gpurun_out/b15/err*
