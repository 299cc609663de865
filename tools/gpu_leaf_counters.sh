#!/bin/bash
# tools/gpu_leaf_counters.sh -- what keeps k_merkle_leaves off its issue floor: scheduler counters of ONE trace commitment
# (tools/leaf_only.py), one rocprofv3 --pmc pass per group, no trace domains.  Output: gpurun_out/leafpmc/*.csv + summary.txt
set -u
R=$PWD
OUT=$R/gpurun_out/leafpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/sq_counters.txt
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_BRANCH" \
         "SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_SMEM SQ_WAVE_DEP_WAIT" \
         "SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32 SQ_BUSY_CU_CYCLES SQ_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python $R/tools/leaf_only.py 20 2 > $OUT/p$i.log 2>&1
  echo "pass $i ($C): rc=$?" >> $OUT/summary.txt
done
cd $R
python - <<PY >> $OUT/summary.txt 2>&1
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if k.startswith("k_merkle_leaves(") or k == "k_merkle_leaves" or k.startswith("void k_ntt_blk12") or k.startswith("k_merkle_fused("):
            k = k.split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]): print("   %-28s %18.0f  over %d rows" % (c, tot[k][c], n[k][c]))
PY
cat $OUT/summary.txt
