#!/bin/bash
# generation 7: phase profile of the Service workload (-DSIMON_TABLE_PROFILE build); usage: bash profiles/gpu_r3t.sh <tag>
set -u
TAG=${1:-r3t}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
for CNT in 64 1024; do
  SIMON_HIP_LIB=$C/libsimon_hip_tprof.so SIMON_TABLE_PROF=1 timeout 600 python bench.py --workload service --counts $CNT --steps 1 --warmup 0 --no-cpu-baseline --no-sub --pmc off 2>&1 >/dev/null | grep SIMON_TABLE_PROF | tail -2
done | tee "$OUT/service_phase_profile.txt"
