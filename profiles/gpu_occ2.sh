#!/bin/bash
# kernel time against batch size, ONE launch band (SIMON_CACHE_BANDS=1), one box: bash profiles/gpu_occ2.sh <tag>
set -u
TAG=${1:-occ2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for B in 1 4; do
for C in 16 64 128 256 512 1024; do
  SIMON_CACHE_BANDS=$B timeout 300 python bench.py --steps 3 --warmup 1 --counts $C --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bands', $B, 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'nodes', d['config']['workload'][33:60])"
done; done
