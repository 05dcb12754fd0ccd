#!/bin/bash
# generation 7 with batched loads in spread_select: parity, then the Service workload per batch size (same box); usage: bash profiles/gpu_r3i.sh <tag>
set -u
TAG=${1:-r3i}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "spread or service" 2>&1 | tail -8 ) > "$OUT/pytest_spread.log"; tail -3 "$OUT/pytest_spread.log"
( timeout 900 python tests/fuzz_spread.py 400 70000 2>&1 | tail -6 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
for LIB in $PWD/profiles/ab/libsimon_r3v.so $C/libsimon_hip.so; do
  [ -f $LIB ] || continue
  for CNT in 64 1024; do
    SIMON_HIP_LIB=$LIB timeout 600 python bench.py --workload service --counts $CNT --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $LIB)', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'gen', d['config']['kernel_generation'], 'parity', d.get('parity_sample',{}).get('mismatches'))"
  done
done
} 2>&1 | tee "$OUT/service_batch_ab.txt"
echo "total $(( $(date +%s) - t0 )) s"
