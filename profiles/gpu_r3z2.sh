#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3z2; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
for L in "" pfp "" pfp; do
  LIB=$C/libsimon_hip${L:+_$L}.so
  SIMON_HIP_LIB=$LIB timeout 600 python profiles/e2e_sweep.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${L:-product}', 'kernel_ms', d['kernel_ms'], 'unscheduled', d['unscheduled_first_last'])"
done | tee "$OUT/e2e_prefetch_ab.txt"
