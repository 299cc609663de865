#!/bin/bash
# Where does an NTT pass spend its time?  Builds three variants of libzkmhip.so with the ZKM_EXP_* hooks of ntt.hip -- workgroup barriers
# removed (NOBAR), butterflies replaced by an add and an xor (NOCOMPUTE), global loads / stores after the first removed (NOMEM) -- and times
# the bench workload with each (results are WRONG by construction; only the per-kernel times mean anything).
#   gpurun -- 'bash tools/ntt_timing_experiments.sh'      (r02: profiles/r02_ntt_split_ab.txt)
set -u
R=$PWD
C=$R/zkm_amd/csrc
mkdir -p gpurun_out/ntt_exp
make -C $C -s -j8
for v in NOBAR NOCOMPUTE NOMEM; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DZKM_EXP_$v -c $C/ntt.hip -o /tmp/ntt_$v.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libzkmhip_exp_$v.so $C/core.o $C/hash.o $C/witness.o /tmp/ntt_$v.o $C/stark.o $C/ctl.o $C/segment.o
done
B="python bench.py --steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --no-extras"
$B > gpurun_out/ntt_exp/base.json 2> gpurun_out/ntt_exp/err_base
for v in NOBAR NOCOMPUTE NOMEM; do
  ZKM_HIP_LIB=/tmp/libzkmhip_exp_$v.so $B > gpurun_out/ntt_exp/$v.json 2> gpurun_out/ntt_exp/err_$v
done
python - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/ntt_exp/*.json')):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], round(d['ms_per_step'], 2), {a: b for a, b in d['kernel_ms_per_proof'].items() if 'ntt' in a})
    except Exception as e:
        print(f, 'ERR', e)
P
