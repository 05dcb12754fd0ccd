#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3am; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
for L in "" ipa3 "" ipa3; do
  LIB=$C/libsimon_hip${L:+_$L}.so
  for CNT in 64 1024; do
  SIMON_HIP_LIB=$LIB timeout 600 python bench.py --workload service --pref 60 --counts $CNT --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service + preferred (60) ${L:-cap4}', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done | tee "$OUT/ipa_waves_ab.txt"
