#!/bin/bash
# tools/gpu_r04_timeline.sh <tag> -- rocprofv3 kernel trace of a few 2^16-cycle segments (one context) and its idle-time digest
TAG=${1:-r04_c}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/bench_segment.py 16 single > $O/seg16_profiled.json 2> $O/seg16.err
cd $R
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/segment_timeline.py $F 4 > $O/timeline.txt 2>&1; head -70 $O/timeline.txt
python tools/segment_phases.py $F --lanes > $O/phases.txt 2>&1; cat $O/phases.txt
rm -rf $O/trace
