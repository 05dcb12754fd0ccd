#!/bin/bash
# one-level vs two-level summary of the score-table kernel (SIMON_TABLE_COARSE=0/1/unset) on one box: parity fuzz, then
# config 3 at 4 096 / 8 192 scenarios, 100 signatures, 256 scenarios.  bash profiles/gpu_coarse.sh <tag> [fuzz cases]
set -u
TAG=${1:-coarse}; NF=${2:-400}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for G in 1 0; do
  echo "fuzz SIMON_TABLE_COARSE=$G"; SIMON_TABLE_COARSE=$G timeout 600 python tests/fuzz_table.py $NF 0 2>&1 | tail -2
done
run() { env $1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'kernel_ms', r['kernel_ms'], 'gen', d['config'].get('kernel_generation'), 'lds', r.get('lds_bytes_per_workgroup'), 'parity', d.get('parity_sample',{}).get('mismatches'))"; }
for G in SIMON_TABLE_COARSE=0 SIMON_TABLE_COARSE=1 X=1; do
  run $G ""
  run $G "--orders-per-gpu 8"
  run $G "--workload config3sig --sigs 100"
  run $G "--counts 256"
done
