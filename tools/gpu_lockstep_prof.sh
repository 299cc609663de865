#!/bin/bash
# rocprofv3 passes over one context proving K 2^16-cycle segments in lock-step: kernel stats, then (own pass) VALU / wave counters.
# usage: tools/gpu_lockstep_prof.sh TAG [K]     -> gpurun_out/TAG_lockstep_kernel_stats.csv, TAG_lockstep_valu.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r05}; K=${2:-16}; cd $R; mkdir -p gpurun_out; export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ls_stats /tmp/ls_pmc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ls_stats -- python $R/tools/lockstep_run.py $K 2 1 > $R/gpurun_out/${TAG}_lockstep_stats.log 2>&1
F=$(find /tmp/ls_stats -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $R/gpurun_out/${TAG}_lockstep_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/ls_pmc -- python $R/tools/lockstep_run.py $K 1 0 > $R/gpurun_out/${TAG}_lockstep_pmc.log 2>&1
F=$(find /tmp/ls_pmc -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/tools/segment_valu_budget.py $F $K $R/gpurun_out/lockstep_valu_latest.json > $R/gpurun_out/${TAG}_lockstep_valu.txt 2>&1
head -40 $R/gpurun_out/${TAG}_lockstep_kernel_stats.csv | cut -c1-200
cat $R/gpurun_out/${TAG}_lockstep_valu.txt
tail -n 2 $R/gpurun_out/${TAG}_lockstep_stats.log; tail -n 2 $R/gpurun_out/${TAG}_lockstep_pmc.log
