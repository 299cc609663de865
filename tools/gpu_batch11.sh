#!/bin/bash
# SQ stall counters of the NTT kernels (own PMC passes, no trace domains)
R=$PWD
OUT=$R/gpurun_out/b11
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
cd $R
python - <<'P'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/b11/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'ntt' in k or 'merkle_leaves' == k.strip():
            agg[k][r['Counter_Name']]=max(agg[k][r['Counter_Name']], float(r['Counter_Value']))
for k,v in agg.items():
    if v.get('SQ_WAVE_CYCLES',0) < 1e8: continue
    wc=v['SQ_WAVE_CYCLES']
    print(k[:52])
    print('   ', {c: round(x/wc,3) for c,x in v.items() if c.startswith(('SQ_WAIT','SQ_ACTIVE'))})
    print('   ', {c: int(x) for c,x in v.items() if not c.startswith(('SQ_WAIT','SQ_ACTIVE'))})
P
