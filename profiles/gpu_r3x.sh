#!/bin/bash
# all-feature kernel, general path: product build vs timing-only builds; usage: bash profiles/gpu_r3x.sh <tag> <lib suffixes...>
set -u
TAG=${1:-r3x}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
for L in "" "$@"; do
  LIB=$C/libsimon_hip${L:+_$L}.so
  SIMON_HIP_LIB=$LIB timeout 600 python profiles/e2e_sweep.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${L:-product}', 'kernel_ms', d['kernel_ms'], 'host_s', d['host_s'], 'unscheduled', d['unscheduled_first_last'])"
done | tee "$OUT/e2e_ab.txt"
