#!/bin/bash
# preferred pod (anti-)affinity in self-referential form on generation 7: parity tests, fuzzers, regression checks; usage: bash profiles/gpu_r3ag.sh <tag>
set -u
TAG=${1:-r3ag}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x -k "hard or preferred or spread or generation_7 or fold or anti_affinity or service" 2>&1 | tail -12 ) > "$OUT/pytest.log"; tail -5 "$OUT/pytest.log"
( timeout 900 python tests/fuzz_spread.py 500 140000 2>&1 | tail -5 ) > "$OUT/fuzz_spread.log"; tail -3 "$OUT/fuzz_spread.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
for LIB in $PWD/profiles/ab/libsimon_r3ab0.so $PWD/open-simulator_amd/csrc/libsimon_hip.so; do
  SIMON_HIP_LIB=$LIB timeout 300 python bench.py --workload service --steps 3 --warmup 1 --no-sub --pmc off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service $(basename $LIB)', 'kernel_ms', d['roofline']['kernel_ms'], 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
done
} | tee "$OUT/service_regression.txt"
echo "total $(( $(date +%s) - t0 )) s"
