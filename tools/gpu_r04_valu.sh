#!/bin/bash
# VALU issue budget of a 2^16-cycle segment in the throughput profile and with the defaults (tools/segment_valu_budget.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r04_valu; export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
for T in "throughput_profile=1" ""; do
  N=$( [ -n "$T" ] && echo tp || echo default )
  ZKM_SEG_TUNING="$T" rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/r04_valu/$N -- python $R/tools/auxpipe_ab.py 3 1 > $R/gpurun_out/r04_valu/$N.log 2>&1
  F=$(find $R/gpurun_out/r04_valu/$N -name "*counter_collection.csv" | head -1)
  echo "== $N" >> $R/gpurun_out/r04_valu/budget.txt
  python $R/tools/segment_valu_budget.py $F 5 >> $R/gpurun_out/r04_valu/budget.txt 2>&1
  rm -rf $R/gpurun_out/r04_valu/$N
done
cat $R/gpurun_out/r04_valu/budget.txt
