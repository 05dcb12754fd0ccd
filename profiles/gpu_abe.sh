#!/bin/bash
# general-path (profiles/e2e_sweep.py) comparison of library builds on ONE box: bash profiles/gpu_abe.sh <reps> name1 name2 ...
REPS=$1; shift
for rep in $(seq 1 $REPS); do for V in "$@"; do
  L=$PWD/open-simulator_amd/csrc/libsimon_hip_$V.so; [ $V = default ] && L=$PWD/open-simulator_amd/csrc/libsimon_hip.so
  SIMON_HIP_LIB=$L timeout 600 python profiles/e2e_sweep.py ${E2E_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V', 'kernel_ms', d['kernel_ms'], 'wg', d['workgroup'])"
done; done
