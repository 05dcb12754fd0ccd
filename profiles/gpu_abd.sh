#!/bin/bash
# default-bench (config 3, cache kernel) comparison of library builds on ONE box: bash profiles/gpu_abd.sh <reps> name1 name2 ...
REPS=$1; shift
for rep in $(seq 1 $REPS); do for V in "$@"; do
  L=$PWD/open-simulator_amd/csrc/libsimon_hip_$V.so; [ $V = default ] && L=$PWD/open-simulator_amd/csrc/libsimon_hip.so
  SIMON_HIP_LIB=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V', 'kernel_ms', d['roofline']['kernel_ms'], 'value', round(d['value']))"
done; done
