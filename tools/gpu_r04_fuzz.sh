#!/bin/bash
# differential fuzz + soak of the round's final code (FRI LDS tiles, factored NTT rounds, throughput profile)
O=gpurun_out/r04_fuzz
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
{
echo "== fuzz_openings 200 11";  timeout 600 python tools/fuzz_openings.py 200 11 2>&1 | tail -2
echo "== fuzz_openings 200 12";  timeout 600 python tools/fuzz_openings.py 200 12 2>&1 | tail -2
echo "== fuzz_segments 10 8";   timeout 900 python tools/fuzz_segments.py 10 8 2>&1 | tail -2
echo "== fuzz_segments 8 9 (throughput profile)";   ZKM_SEG_TUNING=throughput_profile=1 timeout 900 python tools/fuzz_segments.py 8 9 2>&1 | tail -2
echo "== soak 8 x 30";  timeout 900 python tools/soak_segments.py 8 30 2>&1 | tail -2
echo "== soak 16 x 20 (throughput profile)";  ZKM_SEG_TUNING=throughput_profile=1 timeout 900 python tools/soak_segments.py 16 20 2>&1 | tail -2
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
