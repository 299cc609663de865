#!/bin/bash
# rocprofv3 kernel trace of one twelve-table 2^16-cycle segment workload (warm-up + 3 timed segments): real launch counts per kernel
R=$PWD
O=$R/gpurun_out/segprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/bench_segment.py 16 single > $O/seg16_under_rocprof.json 2> $O/err.log
cd $R
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
python - <<P
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
calls = sum(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print("kernel launches in 4 segments (1 warm-up + 3 timed) + set-up:", calls, " -> per segment ~", calls / 4, " total kernel ms", round(tot, 1))
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:16]:
    print("%-64s calls %5s avg %8.1f us total %7.2f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
