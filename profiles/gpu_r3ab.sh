#!/bin/bash
# the fold (hostname-level anti-affinity / ports as monotone infeasibility of the score table): parity, fuzzers, config-3 regression check; usage: bash profiles/gpu_r3ab.sh <tag>
set -u
TAG=${1:-r3ab}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x -k "fold or anti_affinity or rest_features or spread" 2>&1 | tail -12 ) > "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
( timeout 900 python tests/fuzz_table.py 300 90000 2>&1 | tail -5 ) > "$OUT/fuzz_table.log"; tail -2 "$OUT/fuzz_table.log"
( timeout 900 python tests/fuzz_spread.py 300 91000 2>&1 | tail -5 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
for LIB in $PWD/profiles/ab/libsimon_r3ab0.so $PWD/open-simulator_amd/csrc/libsimon_hip.so $PWD/profiles/ab/libsimon_r3ab0.so $PWD/open-simulator_amd/csrc/libsimon_hip.so; do
  SIMON_HIP_LIB=$LIB timeout 300 python bench.py --steps 10 --warmup 2 --no-sub --pmc off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config3 $(basename $LIB)', 'kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'])"
done
} | tee "$OUT/config3_regression.txt"
echo "total $(( $(date +%s) - t0 )) s"
