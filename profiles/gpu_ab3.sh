#!/bin/bash
# N-way same-box comparison of library builds on the default bench (config 3): bash profiles/gpu_ab3.sh <tag> <reps> name1 name2 ...
# (name "default" = libsimon_hip.so, other names = libsimon_hip_<name>.so)
set -u
TAG=$1; REPS=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for rep in $(seq 1 $REPS); do
for V in "$@"; do
  L=$PWD/open-simulator_amd/csrc/libsimon_hip_$V.so; [ $V = default ] && L=$PWD/open-simulator_amd/csrc/libsimon_hip.so
  SIMON_HIP_LIB=$L timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', 'kernel_ms', d['roofline']['kernel_ms'], 'plan', d['config']['plan'])"
done; done
