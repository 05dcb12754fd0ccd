#!/bin/bash
# the fold only in COARSE && !REST && HAS_PIN instantiations: parity of the fold tests + the 200-signature record against the library before the fold; usage: bash profiles/gpu_r3af.sh <tag>
set -u
TAG=${1:-r3af}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -q -x -k "fold or anti_affinity or spread or generation_7 or signatures" 2>&1 | tail -4 ) | tee "$OUT/pytest.log"
( timeout 900 python tests/fuzz_table.py 200 95000 2>&1 | tail -2 ) | tee "$OUT/fuzz_table.log"
{
for LIB in $PWD/profiles/ab/libsimon_r3ab0.so $PWD/open-simulator_amd/csrc/libsimon_hip.so $PWD/profiles/ab/libsimon_r3ab0.so $PWD/open-simulator_amd/csrc/libsimon_hip.so; do
  SIMON_HIP_LIB=$LIB timeout 300 python bench.py --workload config3sig --sigs 200 --steps 5 --warmup 1 --no-sub --pmc off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config3 200 signatures $(basename $LIB)', 'kernel_ms', d['roofline']['kernel_ms'])"
done
} | tee "$OUT/sigs200_regression.txt"
