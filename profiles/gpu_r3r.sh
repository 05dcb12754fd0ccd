#!/bin/bash
# generation 7: where the time of a cycle goes -- timing-only builds that cut pass 1 / pass 2 short (WRONG results, never shipped; the -DSIMON_SPREAD_ABLATE_* switches were removed from the source afterwards); usage: bash profiles/gpu_r3r.sh <tag>
set -u
TAG=${1:-r3r}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
{
for LIB in $C/libsimon_hip.so $C/libsimon_hip_abl1.so $C/libsimon_hip_abl2.so $C/libsimon_hip_abl12.so; do
  for CNT in 64 1024; do
    SIMON_HIP_LIB=$LIB timeout 600 python bench.py --workload service --counts $CNT --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $LIB)', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done
} 2>&1 | tee "$OUT/service_ablation.txt"
