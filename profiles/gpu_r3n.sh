#!/bin/bash
# mid-round check: every GPU test, the spread and table fuzzers, the Service workload row; usage: bash profiles/gpu_r3n.sh <tag>
set -u
TAG=${1:-r3n}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
( timeout 900 python tests/fuzz_spread.py 400 3000 2>&1 | tail -6 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log"
( timeout 900 python tests/fuzz_table.py 150 4000 2>&1 | tail -6 ) > "$OUT/fuzz_table.log"; tail -2 "$OUT/fuzz_table.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
for CNT in 16 64 256 1024; do
  timeout 600 python bench.py --workload service --counts $CNT --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'gen', d['config']['kernel_generation'], 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
done
} 2>&1 | tee "$OUT/service_row.txt"
echo "total $(( $(date +%s) - t0 )) s"
