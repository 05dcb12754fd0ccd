#!/bin/bash
# phase profile of simon_table.hip (profile build) at several batch sizes on one box: bash profiles/gpu_tprof.sh <tag>
set -u
TAG=${1:-tprof}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
L=$PWD/open-simulator_amd/csrc/libsimon_hip_tprof.so
for C in 64; do
  SIMON_HIP_LIB=$L SIMON_TABLE_PROF=1 timeout 300 python bench.py --steps 1 --warmup 0 --counts $C --no-cpu-baseline --no-sub --pmc off 2> "$OUT/prof_c$C.err" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('counts', $C, 'kernel_ms', d['roofline']['kernel_ms'])"
  grep SIMON_TABLE_PROF "$OUT/prof_c$C.err"
done
