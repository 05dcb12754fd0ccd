#!/bin/bash
# summary rows in LDS vs in the HBM workspace (SIMON_TABLE_SUMG=0/1) across batch sizes and signature counts, one box
set -u
TAG=${1:-sumg}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
( SIMON_TABLE_SUMG=1 timeout 600 python -m pytest tests -m gpu -q -x -k "config3 or config2 or cpu_mem or many_sig or large_pool or pin or preset" 2>&1 | tail -4 ) | tee "$OUT/pytest_sumg1.log"
run() { env $1 timeout 300 python bench.py $2 --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])"; }
for G in 0 1; do
  run SIMON_TABLE_SUMG=$G ""
  run SIMON_TABLE_SUMG=$G "--orders-per-gpu 8"
  run SIMON_TABLE_SUMG=$G "--workload config3sig --sigs 100"
  run SIMON_TABLE_SUMG=$G "--counts 256"
done
run X=1 "--workload config3sig --sigs 100"
run X=1 "--workload config3sig --sigs 126"
run X=1 "--orders-per-gpu 8"
