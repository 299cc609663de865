#!/bin/bash
# A/B: wave priority in the latency forms of the permutation (hash.hip ZKM_LATENCY_PRIO) on 2^16-cycle segments
TAG=${1:-r04_d}
O=gpurun_out/$TAG
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_segment.py tests/test_gpu_tables.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for V in prio noprio prio noprio; do
  if [ $V = noprio ]; then export ZKM_HIP_LIB=$PWD/zkm_amd/csrc/libzkmhip_noprio.so; else unset ZKM_HIP_LIB; fi
  python tools/bench_segment.py 16 > $O/seg16_$V.json 2> $O/seg16_$V.err
  python - <<P
import json
s = json.load(open("$O/seg16_$V.json"))
print("$V", round(s["ms_per_segment"], 2), "ms alone;", [(c["contexts"], round(c["segments_per_s"], 1)) for c in s.get("concurrent", [])], [(c["processes"], c["contexts_per_process"], round(c["segments_per_s"], 1)) for c in s.get("concurrent_processes", [])])
P
done
unset ZKM_HIP_LIB
python bench.py --no-extras --no-cpu-baseline --steps 8 > $O/bench.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench.json')); print('headline', round(d['value'],3), round(d['single_context']['ms_per_step'],2), d['kernel_ms_per_proof']['merkle_compress'])"
