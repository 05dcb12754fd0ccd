#!/bin/bash
# generation 7 (soft spread constraints on the score-table kernel): parity at scale + the Service workload on both engines; usage: bash profiles/gpu_r3h.sh <tag>
set -u
TAG=${1:-r3h}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "spread or service" 2>&1 | tail -8 ) > "$OUT/pytest_spread.log"; tail -3 "$OUT/pytest_spread.log"
( timeout 900 python tests/fuzz_spread.py 300 1000 2>&1 | tail -6 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
for C in 64 256 1024; do
  for ENG in table wide; do
    E="X=1"; [ $ENG = wide ] && E="SIMON_NO_SPREAD=1"
    env $E timeout 600 python bench.py --workload service --counts $C --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service', '$ENG', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'kernel', d['config']['kernel'], 'gen', d['config']['kernel_generation'], 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
  done
done
} 2>&1 | tee "$OUT/service_ab.txt"
echo "total $(( $(date +%s) - t0 )) s"
