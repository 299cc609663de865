#!/bin/bash
# SQ counters of the NTT kernels of one 262 x 2^20 commitment (own --pmc passes, no trace domains): where do the waves wait?
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r05_ntt}; cd $R; mkdir -p gpurun_out/$TAG; export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/nttpmc_$i
  rocprofv3 --pmc $C --output-format csv -d /tmp/nttpmc_$i -- python $R/tools/commit_once.py 1 > $R/gpurun_out/$TAG/pass_$i.log 2>&1
  F=$(find /tmp/nttpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/$TAG/pmc_$i.csv
done
rm -rf /tmp/ntt_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ntt_stats -- python $R/tools/commit_once.py 2 > $R/gpurun_out/$TAG/stats.log 2>&1
F=$(find /tmp/ntt_stats -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/$TAG/kernel_stats.csv
python - <<P
import csv, collections, glob
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in sorted(glob.glob("$R/gpurun_out/$TAG/pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in per.items():
    if "ntt" in k or "lde" in k or "merkle_leaves" in k:
        print(k); print("   " + "  ".join("%s=%.3g" % (a, b) for a, b in sorted(c.items())))
P
head -12 $R/gpurun_out/$TAG/kernel_stats.csv | cut -c1-150
