#!/bin/bash
# N-way same-box comparison of library builds on config 5: bash profiles/gpu_ab5.sh <tag> <reps> name1 name2 ...  (see gpu_ab3.sh)
set -u
TAG=$1; REPS=$2; shift 2
for rep in $(seq 1 $REPS); do
for V in "$@"; do
  L=$PWD/open-simulator_amd/csrc/libsimon_hip_$V.so; [ $V = default ] && L=$PWD/open-simulator_amd/csrc/libsimon_hip.so
  SIMON_HIP_LIB=$L timeout 300 python bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', 'kernel_ms', d['roofline']['kernel_ms'])"
done; done
