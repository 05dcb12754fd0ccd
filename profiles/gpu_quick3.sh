#!/bin/bash
# quick visit: the cpu+memory GPU tests, default bench (no CPU legs), batch-size row.  bash profiles/gpu_quick3.sh <tag> [pytest -k expr]
set -u
TAG=${1:-q3}; K=${2:-"config3 or config2 or config4 or cpu_mem or golden or known_answer or many_sig or large_pool or group or threads or size_independent or narrow or explain or pin or preset"}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
( timeout 900 python -m pytest tests -m gpu -q -x -k "$K" > "$OUT/pytest.log" 2>&1 ); grep -n "tests/test_|passed|failed|Fatal|Memory access" "$OUT/pytest.log" | head -12
for rep in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --pmc off 2> "$OUT/bench.$rep.err" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'], 'plan', d['config']['plan'])"
done
for C in 64 256 512 2048; do
  timeout 300 python bench.py --steps 3 --warmup 1 --counts $C --orders-per-gpu 4 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
timeout 300 python bench.py --steps 3 --warmup 1 --orders-per-gpu 8 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])"
timeout 300 python bench.py --workload config3sig --sigs 100 --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2> "$OUT/sig100.err" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('100 sigs kernel_ms', d['roofline']['kernel_ms'], d['config']['kernel'])"
